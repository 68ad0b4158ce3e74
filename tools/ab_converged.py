#!/usr/bin/env python3
"""tools/ab_converged.py — time tap-loop variants of an EXPERIMENTS build on ONE converged state (bench workload): the production
kernels run init + 3 iterations, the state is saved, and every variant then sweeps both colours from that same state
(TSAR_VARIANT_NOW is read per launch by pm_sweep_experiments.hip).  For variants whose results are wrong by construction
(instruction-mix upper bounds), which would never converge on their own.

    python tools/ab_converged.py --variants 131322,2228474 [--rounds 5]"""
import argparse, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tsar_mvs_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="131322")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--width", type=int, default=6048); ap.add_argument("--height", type=int, default=4032); ap.add_argument("--views", type=int, default=10)
a = ap.parse_args()
sc = synth.make_scene(a.width, a.height, a.views, device="cuda", seed=1234)
m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024)
m.pm_init(); m.pm_iterate(3)
planes, cost, _, _ = m.get_plane()
m.enable_kernel_timing(True)
res = {v: [] for v in a.variants.split(",")}
for r in range(a.rounds + 1):
    for v in res:
        m.set_plane(planes, cost)
        m.set_sweep_counter(6)
        os.environ["TSAR_VARIANT_NOW"] = v
        m.reset_kernel_timing()
        m.pm_sweep(0); m.pm_sweep(1)
        t = m.kernel_timing()["pm_sweep"]
        os.environ.pop("TSAR_VARIANT_NOW")
        if r:
            res[v].append(t[1] / t[0])
m.close()
print(json.dumps({v: {"launch_ms_median": statistics.median(x), "rounds": [round(y, 3) for y in x]} for v, x in res.items()}, indent=1))
