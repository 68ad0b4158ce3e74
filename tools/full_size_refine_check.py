#!/usr/bin/env python3
"""tools/full_size_refine_check.py — the reference's LIVE path (external planes -> weak-texture regions -> region RANSAC -> plane
fill, runGipuma main.cpp:1458-1860) at ETH3D size through the CPU oracle and through the HIP library, compared bit for bit operator
by operator (BASELINE configs[3]; the -m gpu tests do this at sizes the oracle finishes in seconds).  Test infrastructure: the
oracle is the checker.

    python tools/full_size_refine_check.py [--width 6048 --height 4032] > report.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle_lib as ol  # noqa: E402
from tsar_mvs_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=6048)
    ap.add_argument("--height", type=int, default=4032)
    args = ap.parse_args()
    w, h = args.width, args.height
    sc = synth.make_scene(w, h, 1, device="cuda", seed=5, textureless=True, flat_cell=3.0)
    images = [im.cpu().numpy() for im in sc.images]
    rng = np.random.default_rng(3)
    textured = sc.textured.cpu().numpy()
    gt = sc.gt_depth.cpu().numpy()
    depth = (gt * (1 + rng.normal(0, 0.002, gt.shape))).astype(np.float32)
    good = textured | (rng.uniform(size=gt.shape) < 0.1)
    depth[~good] = rng.uniform(sc.depth_min, sc.depth_max, gt.shape).astype(np.float32)[~good]
    normal_world = np.ascontiguousarray((sc.gt_normal.cpu().numpy() @ sc.R[0]).astype(np.float32))
    report = {"workload": f"{w}x{h} reference view, external depth / normal maps with 10 % reliable pixels inside the textureless patches", "steps": []}

    def step(name, same, t_cpu, t_gpu, **extra):
        row = {"operator": name, "bit_identical": bool(same), "oracle_seconds": round(t_cpu, 2), "gpu_seconds": round(t_gpu, 4), **extra}
        report["steps"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)

    orc = ol.Oracle(images, sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max)
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max, seed=2024))      # the oracle's default seed
    m.set_views(sc.images[:1], sc.K[:1], sc.R[:1], sc.t[:1])          # the reference view alone: no operator below reads a source image
    t0 = time.perf_counter(); orc.load_planes(depth, normal_world); t_cpu = time.perf_counter() - t0
    t0 = time.perf_counter(); m.load_planes(depth, normal_world); t_gpu = time.perf_counter() - t0
    planes, cost, _, _ = m.get_plane()
    step("load_planes (get_disp)", np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32)) and np.array_equal(cost, orc.c), t_cpu, t_gpu)
    orc.scale[:] = good.astype(np.float32)
    m.set_reliable_mask(good.astype(np.float32))
    t0 = time.perf_counter(); ref = ol.weak_texture(images[0].astype(np.uint8), connect="true", close_lines=True); t_cpu = time.perf_counter() - t0
    t0 = time.perf_counter(); labels, text, size = m.detect_weak_texture(); t_gpu = time.perf_counter() - t0
    step("detect_weak_texture", np.array_equal(labels, ref["labels"]) and np.array_equal(text, ref["text"]) and np.array_equal(size, ref["size"]), t_cpu, t_gpu,
         regions=int(len(text)), weak_regions=int((text == -1).sum()))
    orc.set_regions(ref["labels"], ref["text"], ref["size"])
    t0 = time.perf_counter(); orc.getview(); t_cpu = time.perf_counter() - t0
    t0 = time.perf_counter(); m.getview(); t_gpu = time.perf_counter() - t0
    step("getview", True, t_cpu, t_gpu)
    t0 = time.perf_counter(); planes_ref, ratio_ref = orc.ransac_regions(); t_cpu = time.perf_counter() - t0
    t0 = time.perf_counter(); planes_g, ratio_g = m.ransac_regions(); t_gpu = time.perf_counter() - t0
    weak = np.nonzero(text == -1)[0]
    step("ransac_regions", np.array_equal(planes_g[weak].view(np.uint32), planes_ref[weak].view(np.uint32)) and np.array_equal(ratio_g, ratio_ref), t_cpu, t_gpu,
         inlier_ratio=[round(float(r), 4) for r in ratio_g[weak]])
    t0 = time.perf_counter(); orc.fake_depth(); orc.update_scale(); d_ref = orc.compute_disp(); t_cpu = time.perf_counter() - t0
    t0 = time.perf_counter(); fd = m.fake_depth(); m.fill_textureless(); res = m.get_result(("depth", "normal")); t_gpu = time.perf_counter() - t0
    step("fake_depth + fill_textureless + output maps", np.array_equal(fd, orc.fakedepth) and np.array_equal(res["depth"], d_ref[..., 3]) and np.array_equal(res["normal"], d_ref[..., :3]),
         t_cpu, t_gpu)
    weak_px = np.isin(labels, weak) & ~good
    err_before = np.abs(depth[weak_px] - gt[weak_px]) / gt[weak_px]
    err_after = np.abs(res["depth"][weak_px] - gt[weak_px]) / gt[weak_px]
    report["unreliable_pixels_inside_weak_regions"] = int(weak_px.sum())
    report["median_relative_depth_error_there"] = {"before": float(np.median(err_before)), "after": float(np.median(err_after))}
    report["all_bit_identical"] = all(r["bit_identical"] for r in report["steps"])
    m.close()
    print(json.dumps(report, indent=1))
    sys.exit(0 if report["all_bit_identical"] else 1)


if __name__ == "__main__":
    main()
