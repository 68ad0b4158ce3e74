#!/usr/bin/env python3
"""tools/wmf_only.py — the weighted median filter alone at bench size, for rocprofv3 passes (kernel trace / PMC)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tsar_mvs_amd import api, synth
w, h = int(os.environ.get("W", 6048)), int(os.environ.get("H", 4032))
sc = synth.make_scene(w, h, 2, device="cuda", seed=1234, textureless=True, flat_cell=3.0)
m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024)
m.enable_kernel_timing(True)
gt_d = sc.gt_depth.cpu().numpy()
n_world = np.ascontiguousarray((sc.gt_normal.cpu().numpy() @ sc.R[0]).astype(np.float32))
m.load_planes(gt_d, n_world)
m.getview()
rng = np.random.default_rng(0)
m.set_reliable_mask((rng.uniform(size=(h, w)) < 0.7).astype(np.float32))
m.wmf(4, False)
print(json.dumps(m.kernel_timing()))
m.close()
