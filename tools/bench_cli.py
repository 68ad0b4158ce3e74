#!/usr/bin/env python3
"""tools/bench_cli.py — wall time of the C++ host tool over a whole scene (`tsar_gipuma --all`), files in, .dmb out:
what a user of the reference's shell loop (scripts/courtyard.sh:29-48) would see.  Writes a synthetic scene in the
reference's on-disk layout first (not timed).

    python tools/bench_cli.py [--width 6048 --height 4032 --views 8 --workers 1 2]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tsar_mvs_amd import io as tio, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=6048)
    ap.add_argument("--height", type=int, default=4032)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--timing", action="store_true", help="pass --timing to the tool and print the last view's steps")
    ap.add_argument("--fuse", action="store_true", help="also fuse the views in the same process (--all --fuse): matched maps -> APD/APD_TSAR.ply")
    ap.add_argument("--fusion-cli", action="store_true", dest="fusion_cli", help="then run the separate tsar_fusion tool on the written .dmb files")
    args = ap.parse_args()
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    sc = synth.make_scene(args.width, args.height, args.views - 1, device=dev, seed=1234)
    sc.images = [im.cpu() for im in sc.images]
    cli = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")
    with tempfile.TemporaryDirectory(dir="/tmp") as root:
        root += "/"
        tio.export_scene(sc, root)
        for wk in args.workers:
            t0 = time.perf_counter()
            out = subprocess.run([cli, "--all", "--force", "--gpus=1", f"--workers={wk}", "-mslp_folder", root, "-images_folder", root + "images/",
                                  f"--iterations={args.iters}", "--blocksize=11", "--n_best=1", *(["--fuse"] if args.fuse else []), *(["--timing"] if args.timing else [])], capture_output=True, text=True)
            dt = time.perf_counter() - t0
            ok = out.returncode == 0 and all(os.path.exists(root + f"APD/{k:08d}/TSAR_disp.dmb") for k in range(args.views))
            mp = args.width * args.height * args.views / dt / 1e6
            print(f"workers/GPU {wk}: {dt:.2f} s for {args.views} views of {args.width}x{args.height} ({args.views - 1} sources each, {args.iters} iterations) "
                  f"= {mp:.1f} Mpix/s files-to-files, {'ok' if ok else 'FAILED'}", flush=True)
            if not ok:
                print(out.stdout[-2000:], out.stderr[-2000:])
            else:
                print("   " + " | ".join(l for l in out.stdout.splitlines() if l.startswith("view"))[:600])
                if args.timing:
                    print("   " + " || ".join([l for l in out.stdout.splitlines() if " steps (ms)" in l or "kernels (launches" in l][-2:])[:1500])
                for l in out.stdout.splitlines():
                    if l.startswith("fused"):
                        print("   " + l)
        if args.fusion_cli:
            t0 = time.perf_counter()
            out = subprocess.run([os.path.join(ROOT, "tsar-mvs_amd", "tsar_fusion"), root, "--num_consistent=2", "--reproj_error=2", "--depth_diff=0.01", "--angle=15"],
                                 capture_output=True, text=True)
            dt = time.perf_counter() - t0
            print(f"tsar_fusion over the {args.views} views' .dmb files: {dt:.2f} s, rc {out.returncode}", flush=True)
            print("   " + " | ".join(out.stdout.strip().splitlines()[-4:])[:600])
            if out.returncode != 0:
                print(out.stderr[-1500:])


if __name__ == "__main__":
    main()
