#!/usr/bin/env python3
"""tools/ab_sweep.py — interleaved A/B timing of pm_sweep code variants in ONE process on ONE device (the guide's rule:
perf deltas come from interleaved rounds in one process, never from separate invocations).

    python tools/ab_sweep.py --variants 58,122 [--rounds 4] [--strict] [--env TSAR_STRIP=24]

Each variant gets its own context (TSAR_VARIANT is read by tsar_create); per round every variant runs init + 8 iterations
on the bench workload and reports the mean pm_sweep launch time (HIP events on the context's stream)."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tsar_mvs_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="58")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--width", type=int, default=6048)
ap.add_argument("--height", type=int, default=4032)
ap.add_argument("--views", type=int, default=10)
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--strict", action="store_true")
args = ap.parse_args()

variants = [v for v in args.variants.split(",") if v]
sc = synth.make_scene(args.width, args.height, args.views, device="cuda", seed=1234)
ms = {}
for v in variants:
    env = {}
    for kv in v.split("+")[1:]:
        k, val = kv.split("=")
        env[k] = val
    os.environ["TSAR_VARIANT"] = v.split("+")[0]
    for k, val in env.items():
        os.environ[k] = val
    m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024, flags=api.FLAG_STRICT_DIV if args.strict else 0)
    for k in env:
        os.environ.pop(k)
    m.enable_kernel_timing(True)
    ms[v] = (m, [], [], env)
os.environ.pop("TSAR_VARIANT", None)
for r in range(args.rounds + 1):
    for v in variants:
        m, sweep, init, env = ms[v]
        os.environ.update(env)                      # (harmless: every knob is read once, when the context is created)
        m.reset_kernel_timing()
        m.pm_init()
        m.pm_iterate(args.iters)
        t = m.kernel_timing()
        for k in env:
            os.environ.pop(k)
        if r > 0:                                   # round 0 warms up
            sweep.append(t["pm_sweep"][1] / t["pm_sweep"][0])
            init.append(t["pm_init"][1] / t["pm_init"][0])
out = {}
for v in variants:
    m, sweep, init, env = ms[v]
    m.compute_disp()
    d = torch.empty((args.height, args.width), dtype=torch.float32, device="cuda")
    m.get_result_device(depth=d)
    ok = float(((d - sc.gt_depth).abs() / sc.gt_depth < 0.01).float().mean())
    out[v] = {"sweep_ms_median": statistics.median(sweep), "sweep_ms_min": min(sweep), "init_ms_median": statistics.median(init), "gt_1pct": round(ok, 5),
              "rounds": [round(x, 3) for x in sweep]}
    m.close()
print(json.dumps(out, indent=1))
