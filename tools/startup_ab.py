#!/usr/bin/env python3
"""tools/startup_ab.py — one-view-per-process wall time, A/B over environment knobs, alternated in one session (medians of N runs).
    python tools/startup_ab.py KNOB=VALUE [KNOB2=VALUE ...]      # each argument is one variant against the default
"""
import os
import re
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsar_mvs_amd import io as tio, synth  # noqa: E402
import torch  # noqa: E402

cli = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")
variants = [("default", {})] + [(a, dict([a.split("=", 1)])) for a in sys.argv[1:]]
w, h, N = 6048, 4032, int(os.environ.get("AB_RUNS", "7"))
sc = synth.make_scene(w, h, 10, device="cuda" if torch.cuda.is_available() else "cpu", seed=1234)
sc.images = [im.cpu() for im in sc.images]
with tempfile.TemporaryDirectory(dir="/tmp") as root:
    root += "/"
    tio.export_scene(sc, root)
    names = [f"{k:08d}.pgm" for k in range(11)]
    res = {v[0]: [] for v in variants}
    for r in range(N):
        for tag, env_extra in variants:
            env = dict(os.environ, **env_extra)
            t0 = time.perf_counter()
            out = subprocess.run([cli, *names, "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=8", "--blocksize=11", "--n_best=1", "--timing"],
                                 capture_output=True, text=True, env=env)
            wall = (time.perf_counter() - t0) * 1e3
            m = re.search(r"main entered at (\d+), leaving at (\d+)", out.stdout)
            steps = re.search(r"steps \(ms\): (.*)", out.stdout)
            res[tag].append((wall, float(m.group(2)) if m else -1.0, steps.group(1) if steps else ""))
    for tag, rows in res.items():
        walls = sorted(r[0] for r in rows)
        print(f"{tag}: wall median {statistics.median(walls):.0f} ms (min {walls[0]:.0f}, max {walls[-1]:.0f}); inside median {statistics.median(r[1] for r in rows):.0f} ms; "
              f"after main median {statistics.median(r[0] - r[1] for r in rows):.0f} ms")
        print("   median run steps: " + sorted(rows)[len(rows) // 2][2])
