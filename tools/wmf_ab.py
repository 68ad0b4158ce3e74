#!/usr/bin/env python3
"""tools/wmf_ab.py — wall time of the four weighted-median detection launches (tsar_wmf(4, 0)) at 6048x4032 on a matched state,
with a checksum of the reliability map they leave.  A/B between two builds: run it once per library,
    TSAR_LIB=.../libtsar_hip_prev.so python tools/wmf_ab.py ; python tools/wmf_ab.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsar_mvs_amd import api, synth
w, h = 6048, 4032
sc = synth.make_scene(w, h, 4, device="cuda", seed=1234, textureless=True, flat_cell=3.0)
m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024)
m.enable_kernel_timing(True)
m.pm_init(); m.pm_iterate(2); m.getview(); m.compute_disp()
depth = torch.empty((h, w), dtype=torch.float32, device="cuda")
m.get_result_device(depth=depth)
scale = ((depth - sc.gt_depth).abs() / sc.gt_depth < 0.01).float().cpu().numpy()
res = []
for r in range(3):
    m.set_reliable_mask(scale)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.wmf(4, False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    mask = m.get_reliable_mask()
    res.append((round(dt * 1e3, 1), round(float(mask.mean()), 6), int(mask.view("uint32").sum() % 1000003)))
t = m.kernel_timing()
m.close()
print(json.dumps({"lib": os.environ.get("TSAR_LIB", "libtsar_hip.so"), "wmf_detect_x4_ms / reliable fraction / checksum": res, "wmf_detect_ms_per_launch": round(t["wmf_detect"][1] / t["wmf_detect"][0], 2)}))
