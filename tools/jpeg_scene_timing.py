#!/usr/bin/env python3
"""tools/jpeg_scene_timing.py — what reading a scene's JPEGs (host/tsar_jpeg.h) costs the host tool at ETH3D size, beside the same
scene as PGM: one process per view (the reference's shell loop) and --all.  Writes the synthetic scene both ways first (not timed).

    python tools/jpeg_scene_timing.py [--width 6048 --height 4032 --views 8]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsar_mvs_amd import io as tio, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=6048)
    ap.add_argument("--height", type=int, default=4032)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    import torch
    from PIL import Image
    sc = synth.make_scene(a.width, a.height, a.views - 1, device="cuda" if torch.cuda.is_available() else "cpu", seed=1234)
    sc.images = [im.cpu() for im in sc.images]
    cli = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")
    with tempfile.TemporaryDirectory(dir="/tmp") as top:
        pg, jp = top + "/pgm/", top + "/jpg/"
        tio.export_scene(sc, pg)
        tio.export_scene(sc, jp)
        mb = 0.0
        for k in range(a.views):
            g = np.clip(sc.images[k].numpy(), 0, 255).astype(np.uint8)
            os.remove(jp + f"images/{k:08d}.pgm")
            Image.fromarray(np.stack([g, g, g], -1)).save(jp + f"images/{k:08d}.jpg", quality=92, subsampling=1)
            mb += os.path.getsize(jp + f"images/{k:08d}.jpg") / 1e6
        print(f"{a.views} views {a.width}x{a.height}; JPEG quality 92 4:2:2, {mb / a.views:.1f} MB per file", flush=True)
        for root, ext in ((pg, "pgm"), (jp, "jpg")):
            t0 = time.perf_counter()
            out = subprocess.run([cli, f"--decode-image={root}images/00000000.{ext}"], capture_output=True, text=True)
            print(f"{ext}: decode one image in a process of its own: {(time.perf_counter() - t0) * 1e3:.0f} ms  ({out.stdout.strip()})", flush=True)
            names = [f"{k:08d}.{ext}" for k in range(a.views)]
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                out = subprocess.run([cli, *names, "-mslp_folder", root, "-images_folder", root + "images/", f"--iterations={a.iters}", "--blocksize=11", "--n_best=1", "--timing"],
                                     capture_output=True, text=True)
                ts.append(time.perf_counter() - t0)
                assert out.returncode == 0, out.stdout + out.stderr
            steps = [l for l in out.stdout.splitlines() if " steps (ms)" in l]
            print(f"{ext}: one process per view: {min(ts):.2f} / {sorted(ts)[1]:.2f} s (min / median of 3)   {steps[-1][:260] if steps else ''}", flush=True)
            t0 = time.perf_counter()
            out = subprocess.run([cli, "--all", "--force", "--gpus=1", "-mslp_folder", root, "-images_folder", root + "images/", f"--iterations={a.iters}", "--blocksize=11", "--n_best=1"],
                                 capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert out.returncode == 0, out.stdout + out.stderr
            print(f"{ext}: --all, {a.views} views: {dt:.2f} s = {a.width * a.height * a.views / dt / 1e6:.1f} Mpix/s files-to-files", flush=True)


if __name__ == "__main__":
    main()
