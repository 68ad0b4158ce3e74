#!/usr/bin/env python3
"""tools/fast_vs_strict.py — the two arithmetic modes of the matcher on the bench workload (6048x4032, 10 source
views, 8 iterations, same seed): wall time and how far the fast-mode depth map is from the strict (oracle-exact) one."""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tsar_mvs_amd import api, synth
w, h = 6048, 4032
sc = synth.make_scene(w, h, 10, device="cuda", seed=1234)
out = {}
res = {}
for name, flags in (("strict", api.FLAG_STRICT_DIV), ("fast", 0)):
    m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024, flags=flags)
    d = torch.empty((h, w), dtype=torch.float32, device="cuda"); c = torch.empty_like(d)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.pm_init(); m.pm_iterate(8); m.compute_disp()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m.get_result_device(depth=d, cost=c)
    res[name] = (d, c, dt)
    m.close()
gt = sc.gt_depth
ds, cs, ts = res["strict"]; df, cf, tf = res["fast"]
rel = (df - ds).abs() / ds.clamp_min(1e-6)
out = {"size": [w, h], "strict_s": ts, "fast_s": tf,
       "identical_depth_frac": float((df == ds).float().mean()),
       "within_1e-4_frac": float((rel < 1e-4).float().mean()), "within_1e-3_frac": float((rel < 1e-3).float().mean()),
       "mean_cost_strict": float(cs.mean()), "mean_cost_fast": float(cf.mean()),
       "gt1pct_strict": float(((ds - gt).abs() / gt < 0.01).float().mean()), "gt1pct_fast": float(((df - gt).abs() / gt < 0.01).float().mean())}
print(json.dumps(out))
