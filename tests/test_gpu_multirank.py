"""Multi-rank control flow of bench.py on ONE GPU (the 8-GPU legs are the driver's to run).

bench.py --gpus 2 is launched both ways the driver may type it — under python -m torch.distributed.run (one rank per process,
fresh processes), and as plain `python bench.py --gpus 2`, which starts the ranks itself as a child process before touching the
GPU — with the gloo transport (TSAR_BENCH_BACKEND=gloo, or chosen by bench.py when ranks outnumber devices) so that both ranks may share the single device of the test box: each
rank matches its own reference view, the results are gathered to rank 0 through the double-buffered asynchronous
gather, and --verify-gather checks on rank 0 that every gathered buffer equals the sending rank's own output bit for
bit.  What this cannot show is RCCL itself moving the bytes over xGMI — that row stays "unmeasured on hardware".
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(world, extra, env_extra=None, timeout=600, backend="gloo", as_typed=False):
    """as_typed: plain `python bench.py --gpus N ...` with no launcher and no TSAR_BENCH_* in the environment — bench.py starts the
    ranks itself (and, on a box with fewer devices than ranks, picks the gloo rehearsal transport itself)"""
    env = dict(os.environ)
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "2"})
    if as_typed:
        for k in ("TSAR_BENCH_BACKEND", "TSAR_BENCH_FORCE_DIST", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    else:
        env.update({"TSAR_BENCH_BACKEND": backend, "MASTER_ADDR": "127.0.0.1"})
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    env.update(env_extra or {})
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("as_typed", [False, True])
def test_bench_two_ranks_gather_is_bit_exact(as_typed):
    """as_typed: `python bench.py --gpus 2 --steps 3 --warmup 1 ...` exactly as the driver types its N = 1 run — bench.py launches
    the two ranks under torch.distributed.run itself and verifies the gather without being asked"""
    line = _run_bench(2, ["--steps", "3", "--warmup", "1", "--width", "640", "--height", "480", "--views", "4", "--iters", "2",
                          "--no-cpu-baseline", "--no-kernel-timing"] + ([] if as_typed else ["--verify-gather"]), as_typed=as_typed)
    assert line["gather_check"]["verified"] is True, line["gather_check"]
    assert line["gather_check"]["ranks_differ"] is True, line["gather_check"]     # each rank matched its own view
    # proof of ranks: the one-device rehearsal says what it is — 2 ranks, 1 device, gloo — in the line itself
    assert line["ranks"] == 2 and line["backend"] == "gloo" and line["distinct_devices"] == 1 and line["n_gpus"] == 1
    assert len(line["devices"]) == 2 and all(d["name"] for d in line["devices"]) and line["devices"][0].get("pci", 0) == line["devices"][1].get("pci", 0)
    assert len(line["per_rank_ms_per_step"]) == 2 and all(0 < t <= line["ms_per_step"] * 1.001 for t in line["per_rank_ms_per_step"])
    assert line["scaling"] == "weak" and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["mode"] == "fast"
    # whole-job value = the pixels of both ranks over the max-over-ranks time
    assert abs(line["value"] - 2 * 640 * 480 * 3 / (line["ms_per_step"] * 3e-3) / 1e6) < 1e-6 * line["value"]
    assert line["config"]["frac_depth_within_1pct_of_gt"] > 0.5


@pytest.mark.parametrize("as_typed", [False, True])
def test_bench_three_ranks_odd_step_count(as_typed):
    """an odd number of steps ends on the other buffer set; three ranks on the one device"""
    line = _run_bench(3, ["--steps", "2", "--warmup", "0", "--width", "320", "--height", "240", "--views", "3", "--iters", "1",
                          "--verify-gather", "--no-cpu-baseline", "--no-kernel-timing"], as_typed=as_typed)
    assert line["gather_check"]["verified"] is True and line["gather_check"]["ranks_differ"] is True
    assert len(line["gather_check"]["per_rank_depth_checksum"]) == 3


def test_bench_one_rank_through_rccl():
    """the collective path on the RCCL backend itself, with the one rank a one-GPU box allows (TSAR_BENCH_FORCE_DIST=1):
    communicator creation, the asynchronous double-buffered gather on the collective's stream beside the next view's kernels,
    barrier and all-reduce — everything of the N > 1 path but bytes crossing xGMI"""
    line = _run_bench(1, ["--steps", "3", "--warmup", "1", "--width", "1280", "--height", "960", "--views", "4", "--iters", "2",
                          "--verify-gather", "--no-cpu-baseline", "--no-kernel-timing", "--no-host-boundary", "--no-strict-record"],
                      env_extra={"TSAR_BENCH_FORCE_DIST": "1"}, backend="nccl")
    assert line["gather_check"]["verified"] is True, line["gather_check"]
    assert line["n_gpus"] == 1 and line["steps"] == 3
    assert line["ranks"] == 1 and line["backend"] == "nccl" and line["distinct_devices"] == 1 and len(line["per_rank_ms_per_step"]) == 1
    assert "MI355" in line["devices"][0]["name"] or "gfx950" in str(line["devices"][0].get("arch"))
    assert line["config"]["frac_depth_within_1pct_of_gt"] > 0.5
