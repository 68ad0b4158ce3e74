#!/usr/bin/env python3
"""tools/full_size_oracle_check.py — the bench workload at its FULL size (BASELINE configs[1]: 6048x4032, 1 + 10 views) through
the CPU oracle and through the HIP library in strict mode, compared bit for bit after the random initialisation and after every
red/black iteration.  The -m gpu tests make this comparison at sizes the oracle finishes in seconds; this script is the same
comparison at the size the metric is quoted on (about a minute of 16 host cores per iteration).  Test infrastructure: the oracle
is the checker here, never the thing measured.

    python tools/full_size_oracle_check.py [--width 6048 --height 4032 --views 10 --iters 8] > report.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle_lib as ol  # noqa: E402
from tsar_mvs_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=6048)
    ap.add_argument("--height", type=int, default=4032)
    ap.add_argument("--views", type=int, default=10)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--box", type=int, default=11)
    ap.add_argument("--n_best", type=int, default=1)
    ap.add_argument("--mode", choices=["strict", "fast"], default="strict",
                    help="strict: the reference's arithmetic; fast: the default arithmetic against the oracle's restatement of it (S7, device rcp table)")
    ap.add_argument("--one-call", action="store_true", dest="one_call",
                    help="the library runs all iterations in ONE tsar_pm_iterate call — what bench.py and the CLI do, with the propagation memo and, "
                         "from its seventh launch, the packed sweep — and is compared with the oracle after the last one (the oracle scores every arm of every launch)")
    args = ap.parse_args()
    sc = synth.make_scene(args.width, args.height, args.views, device="cuda", seed=1234)
    images = [im.cpu().numpy() for im in sc.images]
    fast = args.mode == "fast"
    orc = ol.Oracle(images, sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, seed=2024, box=args.box, n_best=args.n_best, flags=ol.FLAGS_FAST_8BIT_IMAGERY if fast else 0)
    m = api.matcher_from_scene(sc, box=args.box, n_best=args.n_best, seed=2024, flags=0 if fast else api.FLAG_STRICT_DIV)
    if fast:
        orc.set_rcp_table(ol.rcp_table_from_device(m))
    report = {"workload": f"{args.width}x{args.height}, 1 ref + {args.views} src views, box {args.box}, n_best {args.n_best}, {args.mode} mode vs the CPU oracle in the same arithmetic",
              "steps": []}

    def compare(tag, t_cpu, t_gpu):
        planes, cost, bv, _ = m.get_plane()
        same_planes = bool(np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32)))
        same_cost = bool(np.array_equal(cost.view(np.uint32), orc.c.view(np.uint32)))
        row = {"after": tag, "planes_bit_identical": same_planes, "costs_bit_identical": same_cost,
               "pixels": int(cost.size), "mean_cost": float(cost.mean()), "oracle_seconds": round(t_cpu, 1), "gpu_seconds": round(t_gpu, 3)}
        if not (same_planes and same_cost):
            row["differing_pixels"] = int((planes.view(np.uint32) != orc.norm4.view(np.uint32)).any(-1).sum() + 0)
        report["steps"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)      # progress: one line per step
        return same_planes and same_cost

    t0 = time.perf_counter(); orc.pm_init(); t_cpu = time.perf_counter() - t0
    t0 = time.perf_counter(); m.pm_init(); t_gpu = time.perf_counter() - t0
    ok = compare("pm_init", t_cpu, t_gpu)
    if args.one_call:
        m.enable_kernel_timing(True)
        t0 = time.perf_counter(); m.pm_iterate(args.iters); t_gpu = time.perf_counter() - t0
        t_cpu = 0.0
        for it in range(args.iters):
            t0 = time.perf_counter(); orc.pm_iterate(1); dt = time.perf_counter() - t0
            t_cpu += dt
            print(json.dumps({"oracle iteration": it + 1, "seconds": round(dt, 1)}), file=sys.stderr, flush=True)
        ok = compare(f"iteration {args.iters}, the library's {args.iters} iterations in one call", t_cpu, t_gpu) and ok
        kt = m.kernel_timing()
        report["sweep_launches"] = {"all": kt["pm_sweep"][0], "packed": kt.get("pm_sweep_packed", (0, 0.0))[0]}
    for it in range(0 if args.one_call else args.iters):
        t0 = time.perf_counter(); orc.pm_iterate(1); t_cpu = time.perf_counter() - t0
        t0 = time.perf_counter(); m.pm_iterate(1); t_gpu = time.perf_counter() - t0
        ok = compare(f"iteration {it + 1}", t_cpu, t_gpu) and ok
    d_ref = orc.compute_disp()
    m.compute_disp()
    res = m.get_result(("depth", "normal"))
    out_same = bool(np.array_equal(res["depth"], d_ref[..., 3]) and np.array_equal(res["normal"], d_ref[..., :3]))
    report["output_maps_bit_identical"] = out_same
    report["all_bit_identical"] = bool(ok and out_same)
    if fast:
        report["rcp_operands_outside_table"] = bool(orc.rcp_out_of_range)
    gt = sc.gt_depth.cpu().numpy()
    report["frac_depth_within_1pct_of_gt"] = float((np.abs(res["depth"] - gt) / gt < 0.01).mean())
    m.close()
    print(json.dumps(report, indent=1))
    sys.exit(0 if report["all_bit_identical"] else 1)


if __name__ == "__main__":
    main()
