"""Run-level tolerance tests of the FAST arithmetic mode — the mode bench.py times and the CLI defaults to.

Strict mode (TSAR_FLAG_STRICT_DIV) is bit-exact against the CPU oracle (test_gpu_parity.py, test_gpu_baseline_configs.py)
and proves the logic.  Fast mode takes four rounding-level liberties in the tap loop; this file bounds
what they do to results, at the level of whole runs:

  (a) cfg1 (640x480, 4 source views, 8 iterations), fast GPU run vs the oracle: fraction of pixels whose depth agrees to
      1e-3 relative and whose normal agrees to 0.1 degree (SURVEY §8c's full-run agreement figure);
  (b) cfg2 (6048x4032, 10 source views, 8 iterations — the bench workload), fast vs strict on the GPU (strict is
      oracle-exact), same figures;
      Both carry a CONTROL: the oracle (or strict) against itself under a different RNG seed — what the reference does
      to itself on every launch (curand_init(clock64()), gipuma.cu:700,1077).  PatchMatch's late refinement steps
      perturb a plane by less than fp32 rounding moves its cost (~1e-5), so any two fp32 evaluation orders flip such
      accepts like a coin, and every flip sends the pixel's later random walk elsewhere: SURVEY §8c's expectation of
      > 99 % pixel-wise agreement after whole runs does not hold between ANY two arithmetic variants, strict ones
      included.  What is asserted instead: fast differs from the oracle by LESS than the oracle differs from itself
      reseeded, the two runs are statistically the same solution (fraction within 1 % of the analytic depth, mean cost,
      median normal error against the analytic normals), and the measured pixel-wise figures stay above stated floors;
  (c) every pixel where a fast half-sweep ends on a different plane than the oracle is re-scored with the ORACLE: the
      plane the GPU kept must be an improvement over the start state and its stored cost must be the oracle's cost of
      that plane within the tolerance — i.e. a valid PatchMatch step, whichever side of a near-tie it took;
  (d) the cost of a given plane, fast vs oracle: p50 / p99 / max instead of one max bound.

Measured values are written to gpurun_out/fast_mode_metrics.json (copied to profiles/ per round) and quoted in DESIGN.md §3.
"""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
from tsar_mvs_amd import api, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- stated tolerances (fp32 throughout; cost lives in [0, 2]) ----------------------------------------
DEPTH_REL = 1e-3          # |depth_fast - depth_ref| / depth_ref
ANGLE_DEG = 0.1           # angle between the normals
COST_P50, COST_P99, COST_MAX = 1e-5, 1e-4, 1e-3     # |cost_fast(plane) - cost_oracle(plane)|; measured 4e-6 / 5e-5 / 3e-4
# Pixel-wise agreement of whole runs has no floor fitted to a measurement (rounds 2-3 had 0.88 / 0.99 / 0.98, set from what was
# observed).  The bar is relative and the same for every configuration: at each level fast may disagree with the reference
# arithmetic at most HALF as often as the reference arithmetic disagrees with ITSELF under another RNG seed — which is what the
# reference does to itself on every launch (curand_init(clock64()), gipuma.cu:700,1077).  Measured ratios: 0.04-0.24 at 1e-3,
# 0.00-0.07 at 1e-2 (gpurun_out/fast_mode_metrics.json).
CONTROL_SHARE = 0.5


def _record(key, value):
    path = os.path.join(ROOT, "gpurun_out", "fast_mode_metrics.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except ValueError:
            data = {}
    data[key] = value
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(key, json.dumps(value))


def _oracle(scene, **kw):
    return ol.Oracle([im.cpu().numpy() for im in scene.images], scene.K, scene.R, scene.t, scene.depth_min, scene.depth_max, **kw)


def _agreement(depth, normal, depth_ref, normal_ref):
    """torch tensors (any device) -> dict of agreement fractions over pixels valid in both maps"""
    valid = (depth > 0) & (depth_ref > 0)
    rel = (depth - depth_ref).abs() / depth_ref.clamp_min(1e-12)
    cosang = (normal * normal_ref).sum(-1).clamp(-1.0, 1.0)
    # small angles from the chord: acos loses everything below ~0.02 degrees in fp32
    chord = (normal - normal_ref).norm(dim=-1)
    ang = torch.rad2deg(2.0 * torch.asin((chord / 2.0).clamp(max=1.0)))
    v = valid.float().sum().clamp_min(1.0)
    out = {
        "valid_both": float(valid.float().mean()),
        "valid_mismatch": float(((depth > 0) != (depth_ref > 0)).float().mean()),
        "identical_depth": float(((depth == depth_ref) & valid).float().sum() / v),
        "depth_within_1e-4": float(((rel < 1e-4) & valid).float().sum() / v),
        "depth_within_1e-3": float(((rel < DEPTH_REL) & valid).float().sum() / v),
        "depth_within_1e-2": float(((rel < 1e-2) & valid).float().sum() / v),
        "angle_within_0.01deg": float(((ang < 0.01) & valid).float().sum() / v),
        "angle_within_0.1deg": float(((ang < ANGLE_DEG) & valid).float().sum() / v),
        "angle_within_1deg": float(((ang < 1.0) & valid).float().sum() / v),
        "both_within": float(((rel < DEPTH_REL) & (ang < ANGLE_DEG) & valid).float().sum() / v),
    }
    del cosang
    return out


def _normal_error_deg(normal, gt_normal_world):
    chord = (normal - gt_normal_world).norm(dim=-1)
    return torch.rad2deg(2.0 * torch.asin((chord / 2.0).clamp(max=1.0)))


def _better_than_control(agree, control, floors):
    """the control-relative bar, plus conservative ABSOLUTE floors per configuration (round-4 advice: a control that happens to be
    loose for a scene or seed must not let a several-fold regression through): `floors` = (depth_within_1e-3, depth_within_1e-2),
    the floors of rounds 2-3, each a stated margin below what is measured"""
    assert agree["depth_within_1e-3"] >= floors[0] and agree["depth_within_1e-2"] >= floors[1], (floors, agree["depth_within_1e-3"], agree["depth_within_1e-2"])
    for k in ("identical_depth", "depth_within_1e-4", "depth_within_1e-3", "depth_within_1e-2", "angle_within_0.1deg", "angle_within_1deg"):
        assert agree[k] >= control[k] - 0.002, (k, agree[k], control[k])
    for k in ("depth_within_1e-3", "depth_within_1e-2"):      # disagreement <= half the reseeded control's
        assert 1.0 - agree[k] <= CONTROL_SHARE * (1.0 - control[k]) + 1e-5, (k, agree[k], control[k])


def test_cfg1_fast_run_vs_oracle():
    """(a) BASELINE configs[0] in full, production arithmetic against the CPU oracle (same seed, same RNG streams)"""
    sc = synth.make_scene(640, 480, 4, seed=21)
    runs = {}
    for tag, seed in (("oracle", 13), ("oracle_reseeded", 14)):
        orc = _oracle(sc, seed=seed)
        orc.pm_init()
        orc.pm_iterate(8)
        out = orc.compute_disp()
        runs[tag] = (torch.from_numpy(out[..., 3].copy()), torch.from_numpy(out[..., :3].copy()), float(orc.c.mean()))
    m = api.matcher_from_scene(sc, seed=13)                 # flags = 0: fast mode
    m.pm_init()
    m.pm_iterate(8)
    m.compute_disp()
    res = m.get_result(("depth", "normal", "cost"))
    m.close()
    runs["fast"] = (torch.from_numpy(res["depth"]), torch.from_numpy(res["normal"]), float(res["cost"].mean()))
    agree = _agreement(runs["fast"][0], runs["fast"][1], runs["oracle"][0], runs["oracle"][1])
    control = _agreement(runs["oracle_reseeded"][0], runs["oracle_reseeded"][1], runs["oracle"][0], runs["oracle"][1])
    gt = sc.gt_depth
    gt_nw = (sc.gt_normal @ torch.from_numpy(sc.R[0].copy()))                 # n_world = R^T n_cam, row-vector form
    for tag, (d, nrm, mc) in runs.items():
        agree[f"gt_1pct_{tag}"] = float(((d - gt).abs() / gt < 0.01).float().mean())
        agree[f"median_normal_error_deg_{tag}"] = float(_normal_error_deg(nrm, gt_nw).median())
        agree[f"mean_cost_{tag}"] = mc
    agree["control_oracle_vs_oracle_reseeded"] = control
    _record("cfg1_fast_vs_oracle_640x480_4views_8iters", agree)
    _better_than_control(agree, control, floors=(0.88, 0.995))            # measured 0.900 / 0.9977
    assert agree["valid_mismatch"] < 1e-3, agree
    # the same solution, statistically
    assert abs(agree["gt_1pct_fast"] - agree["gt_1pct_oracle"]) < 1e-3, agree
    assert abs(agree["mean_cost_fast"] - agree["mean_cost_oracle"]) < 5e-5, agree
    assert abs(agree["median_normal_error_deg_fast"] - agree["median_normal_error_deg_oracle"]) < 0.05 * max(1.0, agree["median_normal_error_deg_oracle"]), agree


def test_cfg2_fast_run_vs_strict_full_size():
    """(b) the bench workload: fast vs strict (= oracle-exact) on the GPU, same seed; 24.4 Mpixel, compared on the device"""
    w, h, n = 6048, 4032, 10
    sc = synth.make_scene(w, h, n, device="cuda", seed=1234)
    dev = torch.device("cuda")
    maps = {}
    for name, flags, seed in (("strict", api.FLAG_STRICT_DIV, 2024), ("fast", 0, 2024), ("strict_reseeded", api.FLAG_STRICT_DIV, 2025)):
        m = api.matcher_from_scene(sc, box=11, n_best=1, seed=seed, flags=flags)
        depth = torch.empty((h, w), dtype=torch.float32, device=dev)
        normal = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
        cost = torch.empty((h, w), dtype=torch.float32, device=dev)
        m.pm_init()
        m.pm_iterate(8)
        m.compute_disp()
        m.get_result_device(depth=depth, normal=normal, cost=cost)
        m.close()
        maps[name] = (depth, normal, float(cost.mean()))
        del cost
    agree = _agreement(maps["fast"][0], maps["fast"][1], maps["strict"][0], maps["strict"][1])
    control = _agreement(maps["strict_reseeded"][0], maps["strict_reseeded"][1], maps["strict"][0], maps["strict"][1])
    gt = sc.gt_depth
    gt_nw = sc.gt_normal @ torch.from_numpy(sc.R[0].copy()).to(dev)
    for tag, (d, nrm, mc) in maps.items():
        agree[f"gt_1pct_{tag}"] = float(((d - gt).abs() / gt < 0.01).float().mean())
        agree[f"median_normal_error_deg_{tag}"] = float(_normal_error_deg(nrm, gt_nw).flatten()[::16].median())
        agree[f"mean_cost_{tag}"] = mc
    agree["control_strict_vs_strict_reseeded"] = control
    _record("cfg2_fast_vs_strict_6048x4032_10views_8iters", agree)
    _better_than_control(agree, control, floors=(0.99, 0.999))            # measured 0.9981 / 0.99997
    assert agree["valid_mismatch"] < 1e-3, agree
    assert abs(agree["gt_1pct_fast"] - agree["gt_1pct_strict"]) < 5e-4, agree
    assert abs(agree["mean_cost_fast"] - agree["mean_cost_strict"]) < 2e-5, agree
    assert abs(agree["median_normal_error_deg_fast"] - agree["median_normal_error_deg_strict"]) < 0.05 * max(1.0, agree["median_normal_error_deg_strict"]), agree


@pytest.mark.parametrize("box,n_best", [(19, 2), (7, 1), (15, 3)])
def test_other_windows_fast_run_vs_strict(box, n_best):
    """the general-window tap loop (pm_core_lut.h; box 19 / n_best 2 are the reference binary's defaults) at run level: fast vs
    strict (= oracle-exact) on the GPU, same seed, with the reseeded strict run as the control"""
    w, h, n = 2016, 1344, 6
    sc = synth.make_scene(w, h, n, device="cuda", seed=77)
    dev = torch.device("cuda")
    maps = {}
    for name, flags, seed in (("strict", api.FLAG_STRICT_DIV, 2024), ("fast", 0, 2024), ("strict_reseeded", api.FLAG_STRICT_DIV, 2025)):
        m = api.matcher_from_scene(sc, box=box, n_best=n_best, seed=seed, flags=flags)
        depth = torch.empty((h, w), dtype=torch.float32, device=dev)
        normal = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
        cost = torch.empty((h, w), dtype=torch.float32, device=dev)
        m.pm_init()
        m.pm_iterate(6)
        m.compute_disp()
        m.get_result_device(depth=depth, normal=normal, cost=cost)
        m.close()
        maps[name] = (depth, normal, float(cost.mean()))
    agree = _agreement(maps["fast"][0], maps["fast"][1], maps["strict"][0], maps["strict"][1])
    control = _agreement(maps["strict_reseeded"][0], maps["strict_reseeded"][1], maps["strict"][0], maps["strict"][1])
    gt = sc.gt_depth
    for tag, (d, nrm, mc) in maps.items():
        agree[f"gt_1pct_{tag}"] = float(((d - gt).abs() / gt < 0.01).float().mean())
        agree[f"mean_cost_{tag}"] = mc
    agree["control_strict_vs_strict_reseeded"] = control
    _record(f"box{box}_nbest{n_best}_fast_vs_strict_{w}x{h}_{n}views_6iters", agree)
    _better_than_control(agree, control, floors=(0.98, 0.9995))           # measured 0.9900-0.9993 / >= 0.9997
    # f = 1129 px here: a 1e-3 depth change is 0.011 px of disparity (between cfg1's 0.004 and cfg2's 0.034)
    assert agree["valid_mismatch"] < 1e-3, agree
    assert abs(agree["gt_1pct_fast"] - agree["gt_1pct_strict"]) < 2e-3, agree
    assert abs(agree["mean_cost_fast"] - agree["mean_cost_strict"]) < 1e-4, agree


@pytest.mark.parametrize("colour", [0, 1])
def test_diverged_pixels_are_valid_patchmatch_steps(mid_scene, colour):
    """(c) one fast half-sweep from the oracle's state; where it lands on another plane than the oracle, the ORACLE's score of
    the GPU's plane must (i) not exceed the start cost beyond the cost tolerance — the GPU accepted it because its own
    score was lower — and (ii) equal the cost the GPU stored, within the tolerance"""
    sc = mid_scene
    orc = _oracle(sc, seed=3)
    orc.pm_init()
    orc.pm_iterate(1)
    if colour == 1:
        orc.pm_sweep(0)
    start_n, start_c = orc.norm4.copy(), orc.c.copy()
    launch = 2 + colour
    m = api.matcher_from_scene(sc, seed=3)
    m.set_plane(start_n, start_c)
    m.set_sweep_counter(launch)
    orc.pm_sweep(colour)
    m.pm_sweep(colour)
    planes, cost, _, _ = m.get_plane()
    m.close()
    swept = ((np.add.outer(np.arange(sc.h), np.arange(sc.w)) & 1) == colour)
    same = np.all(planes.view(np.uint32) == orc.norm4.view(np.uint32), axis=-1)
    assert same[~swept].all()
    div = swept & ~same
    rescored, _, _ = orc.pm_cost_planes(planes)              # the oracle's cost of every plane the GPU holds
    err = np.abs(rescored - cost)[swept]
    worse = (rescored - start_c)[div]                        # > 0: by the oracle's arithmetic the kept plane is worse than the start
    changed = div & ~np.all(planes.view(np.uint32) == start_n.view(np.uint32), axis=-1)
    # how far apart the two outcomes are where they differ, in the oracle's own cost: the near-tie statement, measured
    gap = np.abs(rescored - orc.c)[div]
    rec = {
        "swept": int(swept.sum()), "same_plane_frac": float(same[swept].mean()), "diverged": int(div.sum()),
        "diverged_gpu_changed_plane": int(changed.sum()),
        "stored_vs_oracle_cost_p50": float(np.percentile(err, 50)), "stored_vs_oracle_cost_p99": float(np.percentile(err, 99)),
        "stored_vs_oracle_cost_max": float(err.max()),
        "diverged_worse_than_start_max": float(worse.max()) if div.any() else 0.0,
        "diverged_strictly_worse_count": int((worse > 0).sum()),
        "diverged_outcome_gap_p50": float(np.percentile(gap, 50)) if div.any() else 0.0,
        "diverged_outcome_gap_p95": float(np.percentile(gap, 95)) if div.any() else 0.0,
        "diverged_outcome_gap_max": float(gap.max()) if div.any() else 0.0,
    }
    _record(f"half_sweep_colour{colour}_192x128_4views", rec)
    assert err.max() <= COST_MAX, rec                        # (ii) stored cost == oracle cost of the stored plane
    assert np.percentile(err, 99) <= COST_P99, rec
    assert (worse <= COST_MAX).all(), rec                    # (i) never worse than the start beyond the cost tolerance ...
    assert (worse > COST_P99).mean() < 0.01 if div.any() else True, rec      # ... and beyond its p99 in under 1 % of the diverged pixels
    assert np.percentile(gap, 50) <= COST_P99 if div.any() else True, rec    # the two outcomes are near-ties by the oracle's own score
    assert (cost[swept] <= start_c[swept]).all()             # greedy in the GPU's own arithmetic: strictly never up
    assert same[swept].mean() >= 0.75, rec                   # 85-86 % measured; the checks above carry the statement, this floor catches a several-fold regression


def test_cost_of_a_given_plane_percentiles(small_scene, mid_scene):
    """(d) |cost_fast - cost_oracle| on the same plane, as a distribution: ground-truth planes (low costs, strong cancellation
    in E[x^2]-E[x]^2) and random planes (costs spread over [0, 2])"""
    rec = {}
    for tag, sc in (("96x64_3views", small_scene), ("192x128_4views", mid_scene)):
        orc = _oracle(sc)
        m = api.matcher_from_scene(sc)
        rng = np.random.default_rng(11)
        h, w = sc.h, sc.w
        rnd = np.empty((h, w, 4), np.float32)
        for y in range(h):
            for x in range(w):
                nrm = rng.normal(size=3)
                nrm /= np.linalg.norm(nrm)
                if nrm @ orc.view_vector(x, y) > 0:
                    nrm = -nrm
                nrm = nrm.astype(np.float32)
                rnd[y, x, :3] = nrm
                rnd[y, x, 3] = orc.getD(nrm, x, y, rng.uniform(sc.depth_min, sc.depth_max))
        for kind, planes in (("gt", synth.gt_planes(sc).numpy()), ("random", rnd)):
            c_ref, bv_ref, _ = orc.pm_cost_planes(planes)
            c, bv, _ = m.pm_cost_planes(planes)
            d = np.abs(c - c_ref)
            rec[f"{tag}_{kind}"] = {"p50": float(np.percentile(d, 50)), "p90": float(np.percentile(d, 90)), "p99": float(np.percentile(d, 99)),
                                    "p999": float(np.percentile(d, 99.9)), "max": float(d.max()), "beview_agree": float((bv == bv_ref).mean()),
                                    "maxcost_flag_agree": float(((c == 2.0) == (c_ref == 2.0)).mean())}
            assert np.percentile(d, 50) <= COST_P50, rec
            assert np.percentile(d, 99) <= COST_P99, rec
            assert d.max() <= COST_MAX, rec
            assert (bv == bv_ref).mean() > 0.995, rec
        m.close()
    _record("cost_of_given_plane_fast_vs_oracle", rec)
