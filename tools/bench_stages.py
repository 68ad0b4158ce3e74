#!/usr/bin/env python3
"""tools/bench_stages.py — per-stage timing of the textureless-aware refinement path (BASELINE configs[3]:
ETH3D-size view + gSLICr on the quarter-resolution image + per-region RANSAC + plane fill) and of the
fusion stage, on one GPU.  bench.py measures the headline metric; this script is how the other rows of
SURVEY §8 (A10-A15, N2, N3) are measured.  Wall time per API call (they are synchronous) and the library's own
HIP-event kernel timers.

    python tools/bench_stages.py [--width 6048 --height 4032 --views 10 --iters 2]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tsar_mvs_amd import api, synth  # noqa: E402


def timed(rows, name, fn, bytes_moved=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    row = {"stage": name, "ms": round(dt * 1e3, 3)}
    if bytes_moved:
        row["GB/s"] = round(bytes_moved / dt / 1e9, 1)
    rows.append(row)
    print(json.dumps(row), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=6048)
    ap.add_argument("--height", type=int, default=4032)
    ap.add_argument("--views", type=int, default=10)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--fuse-views", type=int, default=4, dest="fuse_views")
    args = ap.parse_args()
    w, h = args.width, args.height
    dev = torch.device("cuda", 0)
    rows = []
    sc = synth.make_scene(w, h, args.views, device=dev, seed=1234, textureless=True, flat_cell=3.0)
    m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024)
    m.enable_kernel_timing(True)
    np_ = w * h
    timed(rows, "pm_init", m.pm_init)
    timed(rows, f"pm_iterate({args.iters})", lambda: m.pm_iterate(args.iters))
    timed(rows, "lrdiff (rlCost per pixel)", m.lrdiff)
    timed(rows, "getview (confidence)", m.getview, bytes_moved=np_ * 40)
    timed(rows, "compute_disp", m.compute_disp, bytes_moved=np_ * 36)
    # reliable mask as the reference derives it from weak.png: here, pixels whose depth converged
    depth = torch.empty((h, w), dtype=torch.float32, device=dev)
    m.get_result_device(depth=depth)
    gt = sc.gt_depth
    scale = ((depth - gt).abs() / gt < 0.01).float().cpu().numpy()
    timed(rows, "set_reliable_mask (H2D)", lambda: m.set_reliable_mask(scale), bytes_moved=np_ * 4)
    # two rows: what the CLI pays (labels_out = NULL: the labels stay on the device for RANSAC / fill, host/tsar_gipuma.cpp) and the
    # same call when a caller also wants the [h][w] int32 label map on the host (98 MB into pageable memory at 24 MP: a copy, not a kernel)
    timed(rows, "detect_weak_texture, labels_out = NULL (what tsar_gipuma calls: pyrDown x2, Roberts, CCL, stats)", lambda: m.detect_weak_texture(want_labels=False))
    labels, text, size = timed(rows, f"detect_weak_texture + {np_ * 4 / 1e6:.0f} MB label D2H to pageable host memory", m.detect_weak_texture)
    print(json.dumps({"regions": int(len(text)), "weak_regions": int((text == -1).sum()), "largest_weak_px": float(size[text == -1].max()) if (text == -1).any() else 0}))
    planes, ratio = timed(rows, "ransac_regions (first call: loads the rocPRIM code objects)", m.ransac_regions)
    t_first = dict(m.kernel_timing())
    planes, ratio = timed(rows, "ransac_regions", m.ransac_regions)
    t_both = m.kernel_timing()
    warm = {k: round(t_both[k][1] - t_first[k][1], 4) for k in t_both if k.startswith("ransac_")}
    print(json.dumps({"ransac kernels, second call (ms)": warm}), flush=True)
    timed(rows, "fake_depth (D2H)", m.fake_depth)
    timed(rows, "fill_textureless (update_scale + compute_disp)", m.fill_textureless)
    timed(rows, "wmf detect x4", lambda: m.wmf(4, False))
    timed(rows, "wmf fill x3", lambda: m.wmf(3, True))
    # gSLICr on the quarter-resolution colour image (reference main.cpp:1506-1517: size 20, 5 iterations, CIELAB)
    qw, qh = w // 4, h // 4
    g = torch.nn.functional.avg_pool2d(sc.images[0][None, None], 4)[0, 0].clamp(0, 255).to(torch.uint8).cpu().numpy()
    bgra = np.stack([g, g, g, np.full_like(g, 255)], -1)
    st = api.SlicSettings(20, 5, 5.0, 1, 0)
    timed(rows, f"slic {qw}x{qh} (first call: may include the load of its code object)", lambda: m.slic(bgra, st))
    timed(rows, f"slic {qw}x{qh} (H2D image, D2H labels)", lambda: m.slic(bgra, st))
    timing = m.kernel_timing()
    print(json.dumps({"kernel_ms": {k: [v[0], round(v[1] / max(v[0], 1), 4)] for k, v in timing.items()}}))
    m.close()
    # fusion of a few views at this size (depth maps = analytic depth with noise), N3
    n = args.fuse_views
    w, h = w // 2, h // 2            # the fused cloud comes back to the host: keep it below a GB
    sc2 = synth.make_scene(w, h, n - 1, device=dev, seed=1234, all_gt=True)
    depths = [d for d, _ in sc2.meta["gt_all"]]
    normals = [(nc.reshape(-1, 3) @ torch.from_numpy(sc2.R[v]).to(dev)).reshape(h, w, 3).contiguous() for v, (_, nc) in enumerate(sc2.meta["gt_all"])]
    pairs = {v: [s for s in range(n) if s != v] for v in range(n)}
    pts = timed(rows, f"fuse {n} views {w}x{h} (D2H cloud)", lambda: api.fuse(depths, normals, sc2.images, sc2.K, sc2.R, sc2.t, pairs, api.FusionParams(2, 2.0, 0.01, 15.0, 1)))
    print(json.dumps({"fused_points": int(len(pts))}))


if __name__ == "__main__":
    main()
