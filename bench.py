#!/usr/bin/env python3
"""bench.py — depthmap Mpixels/s of the PatchMatch matcher (BASELINE.json metric).

One "step" = the whole hot path for one reference view: random init + ITERS red/black iterations
(propagation + refinement) + plane->depth conversion, at BASELINE.json configs[1]:
ETH3D full resolution 6048x4032, 10 source views, 8 iterations, --blocksize 11, --n_best 1
(reference scripts/courtyard.sh:10-15).  Inputs (images, cameras) are resident in HBM before the timed
region.  N GPUs = N reference views, one per rank (weak scaling), results gathered to rank 0 over RCCL
inside the timed region (the "gather before fusion" of the north star).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 6048 --height 4032 --views 10 --iters 8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def alg_bytes_per_pixel_iteration(n_src: int) -> int:
    """compulsory HBM bytes for one pixel through propagation + refinement once."""
    return 136 + 16 * (1 + n_src)


def alg_flops_per_pixel_iteration(n_src: int, box: int, f: float, depth_min: float) -> float:
    """FLOPs of one pixel-iteration as the reference writes it (no hoisting):
    Hyp * N * (150 + 56 * S), Hyp = 8 propagation arms + R refinement steps, R = number of deltaZ values
    max_disp/2, /10, ... >= 0.01 (gipuma.cu:1066-1090), S = taps of the dilated window."""
    taps = len(range(-(box // 2), box // 2 + 1, 2)) ** 2
    dz, r = f / depth_min / 2.0, 0
    while dz >= 0.01:
        r += 1
        dz /= 10.0
    return (8 + r) * n_src * (150.0 + 56.0 * taps)


def traffic_from_profiles(args=None):
    """HBM bytes per pm_sweep launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE runs of this same command, profiles/r*/pmc_{fetch,write}_size_sweep.csv; bench.py cannot
    read hardware counters itself).  FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled as
    MI355X_MICROARCH.md §HBM prescribes for gfx950 (it tallies 128-B requests at 64 B) — uncalibrated for
    this kernel's 4-byte gathers, so read it as an upper bound.  Converged launches only (the first two of a run
    start from random planes).  Returns {"bytes", "source"} — source = the latest profiles/rNN that holds both passes, named in
    the line as roofline.traffic_source — or None when no profile is committed."""
    import csv
    import glob
    if args is not None and (args.width, args.height, args.views, args.iters, args.box, args.n_best) != (6048, 4032, 10, 8, 11, 1):
        return None          # the committed counters were collected on the default workload only
    prof = sorted(d for d in glob.glob(os.path.join(ROOT, "profiles", "r*")) if os.path.isdir(d))
    if not prof:
        return None
    prof = [d for d in prof if all(os.path.exists(os.path.join(d, f"pmc_{n}_size_sweep.csv")) for n in ("fetch", "write"))]
    if not prof:
        return None
    vals = {"source": os.path.relpath(prof[-1], ROOT)}
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = os.path.join(prof[-1], f"pmc_{name}_size_sweep.csv")
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "pm_sweep" in r["Kernel_Name"]]
        rows = rows[2:] if len(rows) > 2 else rows
        if not rows:
            return None
        vals[name] = sum(float(r["Counter_Value"]) for r in rows) / len(rows) * 1024.0
    return {"bytes": 2.0 * vals["fetch"] + vals["write"], "source": vals["source"]}


def available_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a container
    16 of its 256 hardware threads through cpu.max; 128 OpenMP threads on that quota only thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args):
    """The CPU oracle (kind "port": the reference has no CPU path) on a bounded sample of the same
    workload: same view count / window / iterations, smaller image."""
    import ctypes
    import oracle_lib as ol
    from tsar_mvs_amd import synth
    w, h = args.cpu_width, args.cpu_height
    sc = synth.make_scene(w, h, args.views, seed=1234, step=args.cam_step)
    orc = ol.Oracle([im.numpy() for im in sc.images], sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, box=args.box, n_best=args.n_best)
    cores = available_cpus()
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(cores)                 # same runtime instance the oracle library uses
        cores = int(omp.omp_get_max_threads())
    except OSError:
        pass
    t0 = time.perf_counter()
    orc.pm_init()
    orc.pm_iterate(args.iters)
    orc.compute_disp()
    dt = time.perf_counter() - t0
    return {"value": w * h / dt / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"{w}x{h} synthetic view, {args.views} src views, {args.iters} iters, box {args.box} (CPU oracle, OpenMP), {dt:.1f}s"}


def host_boundary(args, sc):
    """One step through the C ABI the way a host caller without device pointers uses it: images handed over as host
    buffers (H2D + quad-texture build inside tsar_set_views), results copied back to host (D2H inside tsar_get_result).
    Reported next to `value`, never as `value` (which is measured with inputs resident in HBM).  Two legs: the caller's
    buffers page-locked (tsar_host_alloc — what host/tsar_gipuma.cpp does) and plain pageable memory."""
    from tsar_mvs_amd import api
    w, h = args.width, args.height
    pageable = [im.cpu().numpy() for im in sc.images]
    pinned = []
    for im in pageable:
        a = api.pinned_empty(im.shape, np.float32)
        a[...] = im
        pinned.append(a)
    shapes = {"depth": (h, w), "normal": (h, w, 3), "cost": (h, w)}
    bufs = {"pinned": {k: api.pinned_empty(v, np.float32) for k, v in shapes.items()},      # allocated (and touched) once, outside
            "pageable": {k: np.zeros(v, np.float32) for k, v in shapes.items()}}            # the timed step, like a host loop over views
    out = {}
    # one context for both legs, like a host loop over the views of a scene: its device buffers (images, quad textures, planes, the
    # staging arena of tsar_get_result) exist after the first view, so two untimed rounds come first.  (A context created after
    # another one was destroyed pays ~5 ms per hipMalloc for memory the runtime had returned: an artefact of creating contexts in
    # a loop, not part of a view's cost.)
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=args.box, box_vsize=args.box, n_best=args.n_best, depth_min=sc.depth_min, depth_max=sc.depth_max, seed=2024))
    for _ in range(2):
        m.set_views(pinned, sc.K, sc.R, sc.t)
        m.pm_init()
        m.compute_disp()
        m.get_result(out=bufs["pinned"])
    # third leg (round 5): the 8-bit decode itself handed over as bytes, tsar_set_views_u8 — what tsar_gipuma does: a quarter of
    # the bytes cross PCIe and the caller holds no float copies (same results bit for bit: tests/test_gpu_parity.py)
    pinned_u8 = []
    for im in pageable:
        a = api.pinned_empty(im.shape, np.uint8)
        a[...] = im.astype(np.uint8)
        pinned_u8.append(a)
    bufs["u8_pinned"] = bufs["pinned"]
    for leg, imgs in (("pinned", pinned), ("pageable", pageable), ("u8_pinned", pinned_u8)):
        t0 = time.perf_counter()
        m.set_views(imgs, sc.K, sc.R, sc.t, u8=leg.startswith("u8"))
        t1 = time.perf_counter()
        m.pm_init()
        m.pm_iterate(args.iters)
        m.compute_disp()
        t2 = time.perf_counter()
        m.get_result(out=bufs[leg])
        t3 = time.perf_counter()
        out[leg] = {"value": w * h / (t3 - t0) / 1e6, "unit": "Mpix/s", "set_views_h2d_ms": (t1 - t0) * 1e3, "compute_ms": (t2 - t1) * 1e3,
                    "get_result_d2h_ms": (t3 - t2) * 1e3}
    m.close()
    return {"value": out["pinned"]["value"], "unit": "Mpix/s", "pinned": out["pinned"], "pageable": out["pageable"], "u8_pinned": out["u8_pinned"],
            "note": "host buffers in (H2D + quad build), host buffers out (D2H); one view; value = page-locked float32 caller buffers (tsar_host_alloc); u8_pinned = the 8-bit decode handed over as bytes (tsar_set_views_u8)"}


def tolerance_fast_vs_strict(fast, strict):
    """The float tolerance of the headline (fast-arithmetic) result, measured in THIS run against the strict result of the same
    scene, seed and step (strict = the reference's arithmetic, bit-identical to the CPU oracle): fractions over the pixels valid
    in both maps.  depth: |d_fast - d_strict| / d_strict; normal: angle between the world normals (from the chord: acos loses
    everything below ~0.02 degrees in fp32); cost: mean of the stored multi-view cost.  A few elementwise torch kernels on the
    device maps, outside every timed region.  The two runs draw the same random numbers, so every difference is rounding that
    flipped a near-tie accept (and the random walk after it); see DESIGN.md section 3 for the control (strict against itself
    reseeded agrees far less)."""
    (d_f, n_f, c_f), (d_s, n_s, c_s) = fast, strict
    valid = (d_f > 0) & (d_s > 0)
    nv = valid.float().sum().clamp_min(1.0)
    rel = (d_f - d_s).abs() / d_s.clamp_min(1e-12)
    ang = torch.rad2deg(2.0 * torch.asin(((n_f - n_s).norm(dim=-1) / 2.0).clamp(max=1.0)))
    frac = lambda mask: round(float((mask & valid).float().sum() / nv), 6)
    return {"reference": "strict (oracle-exact) run of the same scene, seed and steps, this process",
            "valid_in_both": round(float(valid.float().mean()), 6), "depth_identical": frac(d_f == d_s),
            "depth_within_1e-4": frac(rel < 1e-4), "depth_within_1e-3": frac(rel < 1e-3), "depth_within_1e-2": frac(rel < 1e-2),
            "normal_within_0.1deg": frac(ang < 0.1), "normal_within_1deg": frac(ang < 1.0),
            "mean_cost_fast": float(c_f.double().mean()), "mean_cost_strict": float(c_s.double().mean())}


def strict_mode_record(args, sc, local_rank, fast_maps=None):
    """The same step in the oracle-exact arithmetic (TSAR_FLAG_STRICT_DIV): what the bit-exact parity tests run.  Returns the record
    and, given the fast run's (depth, normal, cost) device maps of its last step, the tolerance of fast against strict."""
    from tsar_mvs_amd import api
    m = api.matcher_from_scene(sc, box=args.box, n_best=args.n_best, seed=2024, device=local_rank, flags=api.FLAG_STRICT_DIV)
    m.enable_kernel_timing(True)
    steps = args.steps                  # the same number of timed steps as the headline

    def one():
        m.pm_init()
        m.pm_iterate(args.iters)
        m.compute_disp()
    one()
    m.reset_kernel_timing()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing = m.kernel_timing()
    tol = None
    if fast_maps is not None:
        dev = fast_maps[0].device
        h, w = args.height, args.width
        maps = (torch.empty((h, w), dtype=torch.float32, device=dev), torch.empty((h, w, 3), dtype=torch.float32, device=dev),
                torch.empty((h, w), dtype=torch.float32, device=dev))
        m.get_result_device(depth=maps[0], normal=maps[1], cost=maps[2])
        tol = tolerance_fast_vs_strict(fast_maps, maps)
    m.close()
    rec = {"value": args.width * args.height * steps / dt / 1e6, "unit": "Mpix/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
           "note": "TSAR_FLAG_STRICT_DIV: correctly rounded divides and the oracle's operation order, bit-identical to the CPU oracle (the mode every bit-exact parity test runs)"}
    if "pm_sweep" in timing and timing["pm_sweep"][0] > 0:
        rec["pm_sweep_avg_launch_ms"] = timing["pm_sweep"][1] / timing["pm_sweep"][0]
    return rec, tol


def self_launch(args) -> int:
    """`python bench.py --gpus N` typed without a launcher (N > 1): start the N ranks under torch.distributed.run as a CHILD process
    and hand on its exit code; rank 0's JSON line reaches stdout through the inherited descriptor.  The parent makes NO torch.cuda /
    HIP call at all (tests/test_driver_gloo.py replaces torch.cuda with an object that raises on any attribute): the ranks
    themselves decide between RCCL and the gloo rehearsal transport (choose_backend), so the child ranks are the only processes
    that ever initialise the runtime.  Replaces the reference's one-process-per-view shell loop (scripts/courtyard.sh:29-48)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = list(sys.argv[1:])
    if "--verify-gather" not in argv:
        argv.append("--verify-gather")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def choose_backend(world: int, n_devices: int, env=os.environ) -> str:
    """Decided inside every rank (all ranks of a node see the same device count, so they agree): RCCL ("nccl") with one device per
    rank; with fewer devices than ranks — a rehearsal on a smaller box — the ranks share devices over gloo and the line says so
    (backend, devices, n_gpus = distinct devices).  TSAR_BENCH_BACKEND overrides."""
    forced = env.get("TSAR_BENCH_BACKEND")
    if forced:
        return forced
    return "nccl" if n_devices >= world else "gloo"


def device_identity(index: int) -> dict:
    """what this rank's device IS, for the proof-of-ranks fields of the line: marketing name, gfx arch, and the PCI address / UUID
    that tell two devices apart (whatever of them this torch build exposes)"""
    ident = {"index": index, "name": None, "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))}
    try:                                           # (identity is evidence, never a reason for a timed run to fail)
        p = torch.cuda.get_device_properties(index)
        ident["name"], ident["arch"] = p.name, getattr(p, "gcnArchName", None)
        if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            ident["pci"] = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        if hasattr(p, "uuid"):
            ident["uuid"] = str(p.uuid)
    except Exception as e:                         # noqa: BLE001
        ident["error"] = repr(e)
    return ident


def device_key(ident: dict):
    """two ranks ran on the same physical device iff their keys are equal"""
    return ident.get("pci") or ident.get("uuid") or (ident.get("visible"), ident["index"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=6048)
    ap.add_argument("--height", type=int, default=4032)
    ap.add_argument("--views", type=int, default=10, help="source views per reference view")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--box", type=int, default=11)
    ap.add_argument("--n_best", type=int, default=1)
    ap.add_argument("--texture-filter-8bit", action="store_true", dest="tex8",
                    help="bilinear weights with 8 fractional bits like the CUDA texture unit (TSAR_FLAG_TEX_FILTER_8BIT); not the headline configuration")
    ap.add_argument("--cam-step", type=float, default=0.03, dest="cam_step")
    ap.add_argument("--cpu-width", type=int, default=960, dest="cpu_width")
    ap.add_argument("--cpu-height", type=int, default=640, dest="cpu_height")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-host-boundary", action="store_true", dest="no_host_boundary")
    ap.add_argument("--no-strict-record", action="store_true", dest="no_strict_record")
    ap.add_argument("--verify-gather", action="store_true", dest="verify_gather",
                    help="after the timed steps, check that rank 0's gathered buffers hold every rank's own results (bit for bit)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))          # before any GPU call in this process
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: start one rank per GPU (python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ..., "
                         "or plain `python bench.py --gpus N`, which launches the ranks itself)")
    dist = None
    # TSAR_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks share devices, results are staged through host memory for the gather); never used for a reported number.
    backend = choose_backend(world, torch.cuda.device_count())
    if backend != "nccl":
        if world > 1 and rank == 0 and "TSAR_BENCH_BACKEND" not in os.environ:
            print(f"bench.py: {torch.cuda.device_count()} device(s) for {world} ranks: rehearsal over gloo, ranks share devices", file=sys.stderr)
        local_rank %= max(torch.cuda.device_count(), 1)
    # TSAR_BENCH_FORCE_DIST=1: also take the collective path with ONE rank (torch.distributed.run --nproc-per-node 1): the RCCL
    # communicator, the asynchronous gather, the barrier and the all-reduce then run on real hardware even on a one-GPU box
    if world > 1 or (os.environ.get("TSAR_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except TypeError:                      # a torch without the device_id keyword
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    from tsar_mvs_amd import api, synth
    from tsar_mvs_amd.driver import alloc_gather_buffers, gather_results

    # each rank owns one reference view of the scene (its own camera arc position), SURVEY §8(e)
    sc = synth.make_scene(args.width, args.height, args.views, device=dev, seed=1234, cam_seed=42 + rank, step=args.cam_step)
    m = api.matcher_from_scene(sc, box=args.box, n_best=args.n_best, seed=2024 + rank, device=local_rank, flags=api.FLAG_TEX_FILTER_8BIT if args.tex8 else 0)
    w, h = args.width, args.height
    # Two sets of result buffers: the gather of step k (RCCL, asynchronous, on the collective's own stream) runs while the
    # kernels of step k+1 fill the other set; a set is reused only after its gather has completed.
    n_sets = 2 if dist is not None else 1
    sets = [[torch.empty((h, w), dtype=torch.float32, device=dev), torch.empty((h, w, 3), dtype=torch.float32, device=dev),
             torch.empty((h, w), dtype=torch.float32, device=dev)] for _ in range(n_sets)]
    use_host_staging = dist is not None and backend != "nccl"
    staged = [[t.cpu() for t in st] for st in sets] if use_host_staging else None
    gathered = [alloc_gather_buffers(dist, staged[k] if use_host_staging else sets[k], dst=0) for k in range(n_sets)] if dist is not None else None
    pending = [None] * n_sets
    state = {"k": 0, "last": 0}

    def wait_set(k):
        if pending[k] is not None:
            for wk in pending[k]:
                wk.wait()
            if backend == "nccl":
                # RCCL work.wait() only makes torch's CURRENT STREAM wait for the collective; the matcher's kernels run on the
                # context's own stream and the host would run ahead.  Drain the current stream so that "the gather has finished
                # reading this buffer set" is a fact on the host before anything is launched that may overwrite it (the gather was
                # issued a whole view earlier, so this never waits in practice).
                torch.cuda.current_stream().synchronize()
            pending[k] = None

    def step():
        k = state["k"] % n_sets
        wait_set(k)                                   # the gather that last read this set
        out_depth, out_normal, out_cost = sets[k]
        m.pm_init()
        m.pm_iterate(args.iters)
        m.compute_disp()
        m.get_result_device(depth=out_depth, normal=out_normal, cost=out_cost)     # complete on return (include/tsar.h)
        if dist is not None:
            src = sets[k]
            if use_host_staging:
                for s_, t in zip(staged[k], sets[k]):
                    s_.copy_(t)
                src = staged[k]
            pending[k] = gather_results(dist, src, dst=0, out=gathered[k], async_op=True)
        state["last"] = k
        state["k"] += 1

    def drain():
        for k in range(n_sets):
            wait_set(k)

    for _ in range(args.warmup):
        step()
    drain()
    if not args.no_kernel_timing:
        m.enable_kernel_timing(True)
        m.reset_kernel_timing()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                           # every gather has landed on rank 0
    torch.cuda.synchronize()
    dt_before_barrier = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt_own = dt_before_barrier                        # this rank's own K steps + its gathers, before it waited for the others
    ranks_info = [{"rank": rank, "local_rank": local_rank, "backend": backend if dist is not None else "none", "device": device_identity(local_rank),
                   "ms_per_step": dt_own / args.steps * 1e3}]
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        everyone_info = [None] * world
        dist.all_gather_object(everyone_info, ranks_info[0])
        ranks_info = everyone_info

    timing = {} if args.no_kernel_timing else m.kernel_timing()
    # sanity on the product of the timed region (not part of the metric): converged depth vs the analytic scene
    gt = sc.gt_depth
    out_depth = sets[state["last"]][0]
    frac_ok = float(((out_depth - gt).abs() / gt < 0.01).float().mean().item())

    gather_check = None
    if dist is not None and args.verify_gather:
        def checksums(ts):
            return [int(t.contiguous().view(torch.int32).to(torch.int64).sum().item()) for t in ts]
        mine = checksums(sets[state["last"]])
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if rank == 0:
            got = [checksums([gathered[state["last"]][k][r] for k in range(3)]) for r in range(world)]
            gather_check = {"verified": got == everyone, "ranks_differ": len({tuple(c) for c in everyone}) == world, "per_rank_depth_checksum": [c[0] for c in everyone]}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * w * h * args.steps / dt / 1e6
        line = {
            "metric": "depthmap Mpixels/sec (ETH3D full-res, 8 iters)", "value": value, "unit": "Mpix/s", "n_gpus": len({device_key(r["device"]) for r in ranks_info}),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ETH3D-size view {w}x{h}, 1 ref + {args.views} src views, {args.iters} PatchMatch iters, box {args.box}, n_best {args.n_best}; one ref view per GPU",
                       "mode": "fast" if not args.tex8 else "fast + 8-bit texture filter",      # default arithmetic of the library and CLI; tolerance vs the oracle: tests/test_gpu_fast_mode.py; "strict" below = oracle-exact
                       "width": w, "height": h, "src_views": args.views, "iters": args.iters, "frac_depth_within_1pct_of_gt": round(frac_ok, 4)},
        }
        if "pm_sweep" in timing and timing["pm_sweep"][0] > 0:
            launches, total_ms = timing["pm_sweep"]
            avg_ms = total_ms / launches
            bytes_per_launch = alg_bytes_per_pixel_iteration(args.views) * (w * h / 2.0)   # one launch = one colour = W*H/2 pixel-iterations
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = traffic_from_profiles(args)
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic["bytes"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
                                "kernel": "pm_sweep_kernel", "avg_launch_ms": avg_ms, "launches": launches,
                                "kernel_note": "three instantiations of the one kernel: <.., 250, 256, false> (gathers as global loads from the byte texture) for the first sweep launch of a view (random planes), <.., 2228474, 256, false> (structured buffer loads from the half-float difference texture, v_fma_mix_f32 blend) for the next five, <.., 2228474, 256, true> (the same tap loop, the wave's surviving (pixel, arm) pairs packed 64 per trip: DESIGN.md section 4, propagation memo) for the last ten; avg_launch_ms is the mean over all 16 per view = (A + 5 x B + 10 x C) / 16 of a rocprofv3 --stats summary",
                                "algorithmic_bytes_per_launch": bytes_per_launch,
                                "note": "the kernel is FP32-VALU bound (SURVEY 8d: ~970 flop/B; issue-slot accounting in profiles/r01/README.md), so the HBM fraction is small by construction and is reported because the metric asks for it; 'valu' prices the same launch against the 157.3 TFLOP/s FP32 vector peak with the reference's as-written flop count. traffic = committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, mean over a view's launches but the first two), null if absent"}
            flops_per_launch = alg_flops_per_pixel_iteration(args.views, args.box, float(sc.K[0][0][0]), sc.depth_min) * (w * h / 2.0)
            tf = flops_per_launch / (avg_ms * 1e-3) / 1e12
            line["roofline"]["valu"] = {"achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "algorithmic_flops_per_launch": flops_per_launch,
                                        "note": "ALGORITHMIC flops = what the reference writes for a launch (every arm of every pixel scored). Since the propagation memo (DESIGN.md section 4) the kernel does not "
                                                "execute all of them: arms whose candidate the pixel already scored and rejected are dropped (0 % in the first iteration, ~68 % in the eighth, bit-identical states), so "
                                                "this fraction is throughput in the reference's units, not issue-slot utilisation; a launch that scores everything (33.5 ms: the third to sixth of a view) reaches 0.65"}
            if "pm_sweep_packed" in timing and timing["pm_sweep_packed"][0] > 0:
                n_p, ms_p = timing["pm_sweep_packed"]
                line["roofline"]["launch_forms"] = {"rolled": {"launches": launches - n_p, "avg_ms": (total_ms - ms_p) / max(launches - n_p, 1)},
                                                    "packed": {"launches": n_p, "avg_ms": ms_p / n_p}}
            line["kernel_ms"] = {k: round(v[1] / max(v[0], 1), 4) for k, v in timing.items()}
        # proof of ranks: what the transport saw, gathered from every rank (all_gather_object) — a SCALE line must show N ranks on
        # N distinct devices over "nccl" (= RCCL); a rehearsal shows its shared device and "gloo".  per_rank_ms_per_step is each rank's
        # own time for the K steps before the closing barrier: a straggler is visible here, behind the all-reduce MAX it is not.
        keys = [device_key(r["device"]) for r in ranks_info]
        line["ranks"] = world
        line["backend"] = ranks_info[0]["backend"]
        line["devices"] = [r["device"] for r in ranks_info]
        line["distinct_devices"] = len(set(keys))
        line["per_rank_ms_per_step"] = [round(r["ms_per_step"], 3) for r in ranks_info]
        if dist is not None and backend == "nccl" and len(set(keys)) != world:
            # RCCL refuses a communicator with two ranks on one device ("Duplicate GPU detected"), so ranks that got this far over
            # nccl ARE on distinct devices: equal keys then mean this torch build reports the same PCI address / UUID for every
            # device (seen with pass-through in containers).  Say so rather than lose the measurement; (visible, index) tells them apart.
            line["device_identity_ambiguous"] = True
            line["distinct_devices"] = len({(k, r["local_rank"]) for k, r in zip(keys, ranks_info)})
            line["n_gpus"] = line["distinct_devices"]
        if gather_check is not None:
            line["gather_check"] = gather_check
        line["config"]["tolerance"] = None
        if world == 1 and not args.no_strict_record:
            # the headline arithmetic's distance from the reference's, measured in this very run: fast maps of the last timed step
            # against the strict maps of the same scene, seed and step
            line["strict"], line["config"]["tolerance"] = strict_mode_record(args, sc, local_rank, fast_maps=tuple(sets[state["last"]]))
        if world == 1 and not args.no_host_boundary:
            line["host_boundary"] = host_boundary(args, sc)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    m.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
