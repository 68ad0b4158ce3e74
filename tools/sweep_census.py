#!/usr/bin/env python3
"""tools/sweep_census.py — how much of the sweep's wave-uniform hypothesis loop is spent on arms that most lanes skip
(VERDICT r2 item 2).  Runs the bench workload (or --width/--height), and before every half-sweep asks the library for the census of
the propagation arms (tsar_selftest_sweep_census); prints one JSON line per half-sweep with the evaluations per wave the rolled loop
runs now (arms with any surviving lane + R refinement steps) against lane-local queues (max over lanes of survivors + R)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tsar_mvs_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=6048); ap.add_argument("--height", type=int, default=4032)
ap.add_argument("--views", type=int, default=10); ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--strict", action="store_true")
a = ap.parse_args()
sc = synth.make_scene(a.width, a.height, a.views, device=torch.device("cuda", 0), seed=1234, cam_seed=42, step=0.03)
m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024, flags=api.FLAG_STRICT_DIV if a.strict else 0)
m.pm_init()
R = None
for it in range(a.iters):
    for colour in (0, 1):
        cs = m.selftest_sweep_census(colour)
        m.pm_sweep(colour)
        w = cs["waves"]
        rec = {"iter": it, "colour": colour, "per_wave": {k: round(cs[k] / w, 3) for k in ("arms_any_lane", "max_lane_survivors", "max_lane_survivors_nodup")},
               "per_pixel": {k: round(cs[k] / cs["pixels"], 3) for k in ("arms_present", "lane_survivors", "lane_survivors_nodup")}}
        print(json.dumps(rec), flush=True)
m.close()
