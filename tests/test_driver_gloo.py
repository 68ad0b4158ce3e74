"""N>1 host path on CPU: view sharding and the result gather, world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tsar_mvs_amd.driver import alloc_gather_buffers, gather_results, owner_of_view, shard_views


def test_shard_views_partitions_all_views():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            mine = shard_views(44, world, r)
            assert all(owner_of_view(v, world) == r for v in mine)
            seen += mine
        assert sorted(seen) == list(range(44))
        sizes = [len(shard_views(44, world, r)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_views(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    views = shard_views(5, world, rank)
    depth = torch.full((6, 8), float(rank + 1))
    normal = torch.full((6, 8, 3), float(10 * (rank + 1)))
    cost = torch.arange(48, dtype=torch.float32).reshape(6, 8) + rank
    got = gather_results(dist, [depth, normal, cost], dst=0)
    if rank == 0:
        ok = len(got) == 3 and all(len(b) == world for b in got)
        ok &= all(float(got[0][r][0, 0]) == r + 1 for r in range(world))
        ok &= all(float(got[1][r][0, 0, 0]) == 10 * (r + 1) for r in range(world))
        ok &= all(float(got[2][r][0, 1]) == 1 + r for r in range(world))
        q.put((ok, views))
    else:
        assert got is None
        q.put((True, views))
    dist.destroy_process_group()


def test_gather_results_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    assert sorted(sum((r[1] for r in res), [])) == [0, 1, 2, 3, 4]


def _worker_async(rank, world, port, q):
    """bench.py's double-buffered loop: the gather of step k is in flight while step k+1 fills the other buffer set"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sets = [[torch.zeros((6, 8)), torch.zeros((6, 8, 3))] for _ in range(2)]
    recv = [alloc_gather_buffers(dist, st, dst=0) for st in sets]
    pending = [None, None]
    seen = []
    ok = True
    for step in range(5):
        k = step % 2
        if pending[k] is not None:
            for wk in pending[k]:
                wk.wait()
            if rank == 0:                                  # the set now holds step - 2 of every rank
                seen.append([float(recv[k][0][r][0, 0]) for r in range(world)])
        sets[k][0].fill_(100.0 * step + rank)              # "compute" step into set k
        sets[k][1].fill_(-(100.0 * step + rank))
        pending[k] = gather_results(dist, sets[k], dst=0, out=recv[k], async_op=True)
    for k in ((5 % 2), (6 % 2)):                            # drain in issue order: step 3 then step 4
        for wk in pending[k]:
            wk.wait()
        if rank == 0:
            seen.append([float(recv[k][0][r][0, 0]) for r in range(world)])
    if rank == 0:
        ok = seen == [[100.0 * st + r for r in range(world)] for st in range(5)]
        ok &= all(float(recv[0][1][r][0, 0, 0]) == -(400.0 + r) for r in range(world))
    if rank != 0:
        ok &= recv[0] is None
    q.put((bool(ok), seen))
    dist.destroy_process_group()


def test_async_double_buffered_gather_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res), res


def test_bench_self_launch_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` typed without a launcher: bench.self_launch starts torch.distributed.run as a CHILD with the same
    arguments (+ --verify-gather), rendezvous on 127.0.0.1.  The parent makes NO torch.cuda call — asserted by replacing torch.cuda
    with an object that raises on any attribute — because the transport (RCCL, or gloo when ranks outnumber devices) is chosen
    inside the ranks (bench.choose_backend).  (The launched ranks themselves are covered by tests/test_gpu_multirank.py.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    class NoCuda:
        def __getattr__(self, name):
            raise AssertionError("the launching parent touched torch.cuda." + name)

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("TSAR_BENCH_BACKEND", raising=False)
    monkeypatch.setattr(bench.torch, "cuda", NoCuda())

    class A:
        gpus = 4
    assert bench.self_launch(A()) == 7                       # the child's exit code is handed on
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(root, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--verify-gather"]
    assert "TSAR_BENCH_BACKEND" not in seen["env"]           # not the parent's decision
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_backend_is_chosen_inside_the_ranks():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    assert bench.choose_backend(8, 8, env={}) == "nccl"      # one device per rank: RCCL
    assert bench.choose_backend(2, 1, env={}) == "gloo"      # rehearsal: ranks share the device
    assert bench.choose_backend(1, 1, env={}) == "nccl"
    assert bench.choose_backend(2, 8, env={"TSAR_BENCH_BACKEND": "gloo"}) == "gloo"
    # device keys: a PCI address or UUID tells devices apart; without either, the visible-devices string and index
    a = {"index": 0, "name": "x", "pci": "0000:05:00"}
    b = {"index": 0, "name": "x", "pci": "0000:15:00"}
    assert bench.device_key(a) != bench.device_key(b)
    assert bench.device_key({"index": 1, "name": "x", "visible": None}) != bench.device_key({"index": 0, "name": "x", "visible": None})
    # the traffic figure names the profile directory it was read from
    t = bench.traffic_from_profiles()
    assert t is None or (t["source"].startswith("profiles/r") and t["bytes"] > 0)
