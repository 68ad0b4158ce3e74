#!/usr/bin/env python3
"""tools/startup_teardown_probe.py — how much of a one-view tsar_gipuma process is spent AFTER its main() has left (the driver tearing
down the process's GPU state), as a function of what the process held: image size (device memory) and page-locked result buffers.
wall = what the shell loop waits for; inside = exec() to the end of main() as the process reports it (--timing)."""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsar_mvs_amd import io as tio, synth  # noqa: E402
import torch  # noqa: E402

cli = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")
for (w, h, iters) in ((640, 480, 8), (3024, 2016, 8), (6048, 4032, 8), (6048, 4032, 1)):
    sc = synth.make_scene(w, h, 10, device="cuda" if torch.cuda.is_available() else "cpu", seed=1234)
    sc.images = [im.cpu() for im in sc.images]
    with tempfile.TemporaryDirectory(dir="/tmp") as root:
        root += "/"
        tio.export_scene(sc, root)
        names = [f"{k:08d}.pgm" for k in range(11)]
        for tag, env_extra in (("page-locked result buffers", {}), ("pageable result buffers", {"TSAR_GIPUMA_NO_PIN": "1"})):
            rows = []
            for r in range(3):
                env = dict(os.environ, **env_extra)
                t0 = time.perf_counter()
                out = subprocess.run([cli, *names, "-mslp_folder", root, "-images_folder", root + "images/", f"--iterations={iters}", "--blocksize=11", "--n_best=1", "--timing"],
                                     capture_output=True, text=True, env=env)
                wall = (time.perf_counter() - t0) * 1e3
                m = re.search(r"main entered at (\d+), leaving at (\d+)", out.stdout)
                rows.append((wall, float(m.group(2)) if m else -1.0))
            print(f"{w}x{h}, 10 sources, {iters} iterations, {tag}: " + "; ".join(f"wall {a:.0f} ms, inside {b:.0f} ms, after main {a - b:.0f} ms" for a, b in rows), flush=True)
