#!/usr/bin/env python3
"""tools/bench_cli_single.py — wall time of ONE invocation of the C++ host tool for one reference view, the way the reference's
shell loop runs it (scripts/courtyard.sh:29-48: one process per reference view, image list on the command line): process start,
HIP initialisation, image decode, matching, .dmb output.  The scene is written first (not timed).

    python tools/bench_cli_single.py [--width 6048 --height 4032 --views 11 --repeat 3 --timing]
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
    ap.add_argument("--views", type=int, default=11)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--timing", action="store_true")
    args = ap.parse_args()
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    sc = synth.make_scene(args.width, args.height, args.views - 1, device=dev, seed=1234)
    sc.images = [im.cpu() for im in sc.images]
    cli = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")
    with tempfile.TemporaryDirectory(dir="/tmp") as root:
        root += "/"
        tio.export_scene(sc, root)
        names = [f"{k:08d}.pgm" for k in range(args.views)]
        cmd = [cli, *names, "-mslp_folder", root, "-images_folder", root + "images/", f"--iterations={args.iters}", "--blocksize=11", "--n_best=1",
               *(["--timing"] if args.timing else [])]
        for r in range(args.repeat):
            t0 = time.perf_counter()
            out = subprocess.run(cmd, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            ok = out.returncode == 0
            print(f"run {r}: {dt:.2f} s for one {args.width}x{args.height} view with {args.views - 1} sources = {args.width * args.height / dt / 1e6:.1f} Mpix/s "
                  f"process-to-files, {'ok' if ok else 'FAILED'}", flush=True)
            print("   " + "\n   ".join(l for l in out.stdout.strip().splitlines()[-6:]), flush=True)
            if not ok:
                print(out.stderr[-2000:])
                sys.exit(1)


if __name__ == "__main__":
    main()
