#!/usr/bin/env python3
"""tools/bench_cli_tsar.py — wall time of the reference's LIVE path through the C++ host tool (`tsar_gipuma --mode=tsar`,
runGipuma main.cpp:1458-1860): external depth / normal maps + weak.png in, weak-texture regions -> region RANSAC -> plane fill,
TSAR_disp.dmb / TSAR_normals.dmb out.  BASELINE configs[3] at ETH3D size on one GPU.  The scene is written first (not timed).

    python tools/bench_cli_tsar.py [--width 6048 --height 4032 --views 3 --repeat 3]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from tsar_mvs_amd import io as tio, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=6048)
    ap.add_argument("--height", type=int, default=4032)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--timing", action="store_true", help="pass --timing to the tool: wall time of each host-side step")
    ap.add_argument("--workers", type=int, default=1, help="host workers per GPU with --all")
    ap.add_argument("--all", action="store_true", help="refine every view of the scene in ONE process (tsar_gipuma --all --mode=tsar)")
    args = ap.parse_args()
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    sc = synth.make_scene(args.width, args.height, args.views - 1, device=dev, seed=5, textureless=True, flat_cell=3.0, all_gt=args.all)
    sc.images = [im.cpu() for im in sc.images]
    cli = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")
    with tempfile.TemporaryDirectory(dir="/tmp") as root:
        root += "/"
        tio.export_scene(sc, root)
        rng = np.random.default_rng(3)
        textured = sc.textured.cpu().numpy()
        gt = sc.gt_depth.cpu().numpy()
        depth = (gt * (1 + rng.normal(0, 0.002, gt.shape))).astype(np.float32)
        good = textured | (rng.uniform(size=gt.shape) < 0.1)
        junk = rng.uniform(sc.depth_min, sc.depth_max, gt.shape).astype(np.float32)
        depth[~good] = junk[~good]
        normal_world = np.ascontiguousarray((sc.gt_normal.cpu().numpy() @ sc.R[0]).astype(np.float32))
        for v in range(args.views if args.all else 1):
            if v > 0:       # the other views' external maps: their own ground truth, the reference view's failure pattern
                gt_v, n_v = sc.meta["gt_all"][v]
                gt_v = gt_v.cpu().numpy()
                depth = (gt_v * (1 + rng.normal(0, 0.002, gt_v.shape))).astype(np.float32)
                depth[~good] = junk[~good]
                normal_world = np.ascontiguousarray((n_v.cpu().numpy() @ sc.R[v]).astype(np.float32))
            apd = root + f"APD/{v:08d}/"
            os.makedirs(apd, exist_ok=True)
            tio.write_dmb(apd + "depths_geom.dmb", depth)
            tio.write_dmb(apd + "normals.dmb", normal_world)
            tio.write_reliable_mask(apd + "weak.png", good)
        names = [f"{k:08d}.pgm" for k in range(args.views)]
        common = ["-mslp_folder", root, "-images_folder", root + "images/", "--blocksize=11", "--n_best=1", "--mode=tsar", *(["--timing"] if args.timing else [])]
        cmd = [cli, "--all", "--force", "--gpus=1", f"--workers={args.workers}", *common] if args.all else [cli, *names, *common]
        n_done = args.views if args.all else 1
        for r in range(args.repeat):
            t0 = time.perf_counter()
            out = subprocess.run(cmd, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            ok = out.returncode == 0
            print(f"run {r}: {dt:.2f} s for {n_done} {args.width}x{args.height} view(s) = {n_done * args.width * args.height / dt / 1e6:.1f} Mpix/s files-to-files, {'ok' if ok else 'FAILED'}", flush=True)
            print("   " + " | ".join(l for l in out.stdout.strip().splitlines()[-12:]), flush=True)
            trace = [ln for ln in out.stderr.splitlines() if ln.startswith("[")]       # TSAR_TRACE_HOST=1: host-side steps of the operators
            if trace:
                print("   " + " | ".join(trace[-40:]), flush=True)
            if not ok:
                print(out.stderr[-2000:])
                sys.exit(1)


if __name__ == "__main__":
    main()
