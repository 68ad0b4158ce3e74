"""CPU tests that pin the oracle (and with it the semantics the HIP path is checked against) on answers
known independently of this code base: published Philox vectors, closed-form geometry, libm, and
hand-constructed propagation cases.  The reference has no tests or fixtures of its own and
cannot be built here, so these — not reference outputs — are what the oracle is pinned by
("parity unpinned")."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as ol
from tsar_mvs_amd import synth


def _orc(scene, **kw):
    return ol.Oracle([im.numpy() for im in scene.images], scene.K, scene.R, scene.t, scene.depth_min, scene.depth_max, **kw)


def test_philox_published_vectors():
    """Random123 known-answer vectors for philox4x32-10"""
    L = ol.lib()
    out = (C.c_uint32 * 4)()
    L.orc_philox_raw(0, 0, 0, 0, 0, 0, out)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    L.orc_philox_raw(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, out)
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    L.orc_philox_raw(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0, out)
    assert [hex(v) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_uniforms_are_in_half_open_unit_interval():
    u = np.stack([ol.rng4(123, p, 1, s) for p in range(200) for s in range(4)])
    assert u.min() > 0.0 and u.max() <= 1.0
    assert abs(u.mean() - 0.5) < 0.03
    # streams are distinct
    assert not np.array_equal(ol.rng4(1, 5, 1, 0), ol.rng4(1, 5, 2, 0))
    assert not np.array_equal(ol.rng4(1, 5, 1, 0), ol.rng4(2, 5, 1, 0))


def test_expf_against_libm():
    xs = np.linspace(-16.0, 0.0, 4001, dtype=np.float32)
    got = np.array([ol.expf(float(x)) for x in xs], np.float32)
    ref = np.exp(xs.astype(np.float64))
    ulp = np.abs(got.astype(np.float64) - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 1.5, ulp.max()
    assert ol.expf(0.0) == 1.0


def _two_view_wall(w=64, h=48, f=100.0, b=0.5, Z=5.0, seed=0):
    """fronto-parallel textured wall at depth Z, source camera translated by b along x (the known-answer
    configuration recorded in SURVEY §8c)"""
    rng = np.random.default_rng(seed)
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    big = rng.uniform(0, 255, size=(h, w + 64)).astype(np.float32)
    # smooth a little so that bilinear sampling is meaningful, then quantise like an 8-bit image
    big = (big + np.roll(big, 1, 1) + np.roll(big, 1, 0) + np.roll(big, -1, 1) + np.roll(big, -1, 0)) / 5.0
    big = np.round(big)
    d = int(round(f * b / Z))     # integer disparity: 10 px
    ref = big[:, 32:32 + w].copy()
    src = big[:, 32 - d:32 - d + w].copy()       # x_src = x_ref + d ... the point at ref x appears at x + f*t_x/Z
    Ks = np.stack([K, K])
    Rs = np.stack([np.eye(3, dtype=np.float32)] * 2)
    ts = np.array([[0, 0, 0], [b, 0, 0]], np.float32)
    return ref, src, Ks, Rs, ts, d


def test_homography_of_fronto_parallel_plane_is_the_disparity_shift():
    ref, src, K, R, t, d = _two_view_wall()
    o = ol.Oracle([ref, src], K, R, t, 2.0, 20.0)
    n4 = np.array([0, 0, -1, 5.0], np.float32)        # n.X + d = 0  ->  -Z + 5 = 0
    H = o.homography(1, n4)
    H = H / H[2, 2]
    assert np.allclose(H, [[1, 0, 10.0], [0, 1, 0], [0, 0, 1]], atol=1e-4)
    assert abs(o.depth_from_plane(n4, 20, 30) - 5.0) < 1e-5


def test_true_plane_scores_zero_and_wrong_plane_does_not():
    ref, src, K, R, t, d = _two_view_wall()
    o = ol.Oracle([ref, src], K, R, t, 2.0, 20.0)
    true_plane = np.array([0, 0, -1, 5.0], np.float32)
    wrong_plane = np.array([0, 0, -1, 6.5], np.float32)   # 1.3 x depth
    c_true = [o.pm_cost(1, x, y, true_plane) for x in range(12, 40, 3) for y in range(10, 38, 3)]
    c_wrong = [o.pm_cost(1, x, y, wrong_plane) for x in range(12, 40, 3) for y in range(10, 38, 3)]
    assert max(c_true) < 1e-5
    assert np.median(c_wrong) > 0.5
    # textureless window -> MAXCOST (kMinVar, reference gipuma.cu:289-291)
    flat = np.zeros_like(ref)
    o2 = ol.Oracle([flat, src], K, R, t, 2.0, 20.0)
    assert o2.pm_cost(1, 30, 20, true_plane) == 2.0


def test_window_taps_skip_the_centre_and_clamp_at_borders():
    """WIN_INCREMENT 2 with an odd radius never samples the centre pixel (reference gipuma.cu:37,259-260):
    changing only the centre pixel changes the cost only through the colour weights"""
    ref, src, K, R, t, d = _two_view_wall()
    o = ol.Oracle([ref, src], K, R, t, 2.0, 20.0)
    n4 = np.array([0, 0, -1, 5.0], np.float32)
    # border pixel: taps fall outside the image and are clamped, result finite and in range
    c = o.pm_cost(1, 0, 0, n4)
    assert 0.0 <= c <= 2.0
    c = o.pm_cost(1, 63, 47, n4)
    assert 0.0 <= c <= 2.0


def test_plane_offset_depth_round_trip(small_scene):
    o = _orc(small_scene)
    rng = np.random.default_rng(1)
    for _ in range(200):
        x, y = int(rng.integers(0, small_scene.w)), int(rng.integers(0, small_scene.h))
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        if n @ o.view_vector(x, y) > 0:
            n = -n
        if abs(n @ o.view_vector(x, y)) < 0.2:
            continue
        depth = float(rng.uniform(small_scene.depth_min, small_scene.depth_max))
        d = o.getD(n.astype(np.float32), x, y, depth)
        back = o.depth_from_plane(np.array([*n, d], np.float32), x, y)
        assert abs(back - depth) / depth < 2e-5
    v = o.view_vector(10, 10)
    assert abs(np.linalg.norm(v) - 1) < 1e-6 and v[2] > 0
    assert o.depth_from_plane(np.array([0, 0, -1, np.nan], np.float32), 3, 3) == 1000.0   # gipuma.cu:449-450


def test_camera_reorigin(small_scene):
    """reference camera becomes K[I|0]; relative poses reproduce the original projections"""
    sc = small_scene
    o = _orc(sc)
    c0 = o.camera(0)
    assert np.allclose(np.array(c0.R).reshape(3, 3), np.eye(3))
    assert np.allclose(c0.t, 0) and np.allclose(c0.C, 0) and np.allclose(c0.P34, 0)
    assert np.allclose(np.array(c0.Minv).reshape(3, 3) @ sc.K[0], np.eye(3), atol=1e-5)
    assert c0.baseline == 1.0 and abs(c0.alpha - 1.0) < 1e-6
    # a world point seen by view 2: projecting through (R_rel, t_rel) from reference-camera coordinates
    Xw = np.array([0.3, -0.2, 0.4])
    Xr = sc.R[0].astype(np.float64) @ Xw + sc.t[0]
    c2 = o.camera(2)
    Xc = np.array(c2.R).reshape(3, 3) @ Xr + np.array(c2.t)
    Xc_direct = sc.R[2].astype(np.float64) @ Xw + sc.t[2]
    assert np.allclose(Xc, Xc_direct, atol=1e-5)
    assert abs(o.min_disp - c0.f / sc.depth_max) < 1e-4 and abs(o.max_disp - c0.f / sc.depth_min) < 1e-4


def test_multiview_best_n(small_scene):
    sc = small_scene
    gp = synth.gt_planes(sc).numpy()
    x, y = 40, 30
    per_view = sorted(min(_orc(sc).pm_cost(v, x, y, gp[y, x]), 2.0) for v in (1, 2, 3))
    c1, bv1, rt1 = _orc(sc, n_best=1).pm_cost_multiview(x, y, gp[y, x])
    c2, _, _ = _orc(sc, n_best=2).pm_cost_multiview(x, y, gp[y, x])
    call, _, _ = _orc(sc, n_best=2, cost_comb=0).pm_cost_multiview(x, y, gp[y, x])
    assert c1 == pytest.approx(per_view[0], abs=0)
    assert c2 == pytest.approx(np.float32((np.float32(per_view[0]) + np.float32(per_view[1])) / np.float32(2)), abs=0)
    assert call == pytest.approx(float(np.float32(np.float32(np.float32(per_view[0]) + np.float32(per_view[1])) + np.float32(per_view[2])) / np.float32(3)), rel=1e-6)
    assert rt1 == pytest.approx(per_view[0] / per_view[1], rel=1e-6)
    assert bv1 in (1, 2, 3)
    # view subset restricts the views used
    cs, bvs, _ = _orc(sc, n_best=1, subset=[2]).pm_cost_multiview(x, y, gp[y, x])
    assert bvs == 2 and cs == _orc(sc).pm_cost(2, x, y, gp[y, x])


def _cost_map(h, w, fill=1.0):
    return np.full((h, w), fill, np.float32)


def test_candidate_selection_far_arms_and_quirks(small_scene):
    sc = small_scene
    h, w = sc.h, sc.w
    x, y = 40, 30
    p = y * w + x
    c = _cost_map(h, w)
    c[y - 9, x] = 0.1      # up_far offsets are 3,5,...,23
    c[y - 8, x] = 0.0      # even offset: never looked at
    c[y, x - 23] = 0.2
    c[y, x - 25] = 0.0     # beyond the 11th tap
    o = _orc(sc)
    cand = o.select_candidates(c, x, y)
    assert cand[0] == p - 9 * w
    assert cand[2] == p - 23
    # right_far: the reference's comparison is inverted -> it walks to the LARGEST cost
    c2 = _cost_map(h, w)
    c2[y, x + 3] = 0.5
    c2[y, x + 7] = 1.7
    c2[y, x + 9] = 0.1
    assert o.select_candidates(c2, x, y)[3] == p + 7
    fixed = _orc(sc, flags=2)
    assert fixed.select_candidates(c2, x, y)[3] == p + 9
    # down_far: the reference seeds the running minimum with c[up_far]
    c3 = _cost_map(h, w)
    c3[y - 3, x] = 0.05          # up_far seed, smaller than anything below
    c3[y + 5, x] = 0.3
    assert o.select_candidates(c3, x, y)[1] == p + 3 * w        # nothing beats the (foreign) seed -> stays at down_far
    assert _orc(sc, flags=1).select_candidates(c3, x, y)[1] == p + 5 * w


def test_candidate_selection_near_arms_and_borders(small_scene):
    sc = small_scene
    h, w = sc.h, sc.w
    o = _orc(sc)
    x, y = 40, 30
    p = y * w + x
    c = _cost_map(h, w)
    c[y - 3, x - 1] = 0.2      # up_near V tap i=1: (x-1, y-3)
    c[y + 2, x] = 0.3          # down_near V tap i=0: (x, y+2)
    c[y + 2, x - 4] = 0.1      # left_near V tap i=2: (x-4, y+2)
    c[y - 1, x + 3] = 0.4      # right_near V tap i=1: (x+3, y-1)
    cand = o.select_candidates(c, x, y)
    assert cand[4] == p - 3 * w - 1
    assert cand[5] == p + 2 * w
    assert cand[6] == p - 4 + 2 * w
    assert cand[7] == p + 3 - w
    # image corners: arms that would leave the image are skipped
    cand = o.select_candidates(c, 0, 0)
    assert cand[0] == -1 and cand[2] == -1 and cand[4] == -1 and cand[6] == -1
    assert cand[1] >= 0 and cand[3] >= 0 and cand[5] == w and cand[7] == 1
    cand = o.select_candidates(c, w - 1, h - 1)
    assert cand[1] == -1 and cand[3] == -1 and cand[5] == -1 and cand[7] == -1


def test_bilinear_semantics():
    img = np.arange(12, dtype=np.float32).reshape(3, 4) * 10
    L = ol.lib()
    f = lambda u, v: L.orc_bilinear(img.ctypes.data_as(C.c_void_p), 4, 3, C.c_float(u), C.c_float(v))
    assert f(1, 1) == 50.0
    assert f(1.5, 1) == 55.0
    assert f(1, 1.5) == 70.0
    assert f(-5, -5) == 0.0 and f(10, 10) == 110.0          # clamp addressing
    assert f(3.5, 0) == 30.0                                # beyond the last column the edge texel repeats
    assert f(float("nan"), 0) == 0.0                        # NaN coordinates are defined (clamped low)


def test_bilinear_8bit_filter_mode():
    """S3' (TSAR_FLAG_TEX_FILTER_8BIT): fractions rounded to 8 fractional bits before the blend, as a CUDA texture fetch with
    linear filtering stores them (1.8 fixed point; CUDA C Programming Guide, "Linear Filtering")"""
    img = np.array([[100, 200, 100, 100], [100, 200, 100, 100]], np.float32)
    L = ol.lib()
    q = lambda u, v: L.orc_bilinear_q8(img.ctypes.data_as(C.c_void_p), 4, 2, C.c_float(u), C.c_float(v))
    f = lambda u, v: L.orc_bilinear(img.ctypes.data_as(C.c_void_p), 4, 2, C.c_float(u), C.c_float(v))
    assert q(0.5, 0) == f(0.5, 0) == 150.0                  # 128/256 is exact
    assert abs(f(0.3, 0) - 130.0) < 1e-4
    assert q(0.3, 0) == np.float32(100.0) + np.float32(77.0 / 256.0) * np.float32(100.0)     # round(0.3 * 256) = 77
    assert q(0.999, 0) == 200.0                             # 255.7 -> 256: the weight 1.0 is representable
    assert q(0.001, 0) == 100.0


def test_8bit_filter_mode_changes_costs_slightly(small_scene):
    sc = small_scene
    a, b = _orc(sc), _orc(sc, flags=32)
    planes = synth.gt_planes(sc).numpy()
    ca, _, _ = a.pm_cost_planes(planes)
    cb, _, _ = b.pm_cost_planes(planes)
    d = np.abs(ca - cb)
    assert 0 < d.max() < 0.05 and np.median(d) < 2e-3       # same matcher, marginally different samples


def test_init_window_radius_is_box_over_two(small_scene):
    """gipuma_init_cu2 derives its window radius as box / 2 (gipuma.cu:693-694), every other kernel as (box - 1) / 2 (:858-859,
    :1065-1066, :1175-1176).  An even box therefore initialises on the window of the next odd box and sweeps on the window of the
    previous one: box 12 -> init like box 13 (radius 6), cost evaluation like box 11 (radius 5)."""
    sc = small_scene
    args = ([im.numpy() for im in sc.images], sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max)
    o12, o13, o11 = (ol.Oracle(*args, box=b, seed=9) for b in (12, 13, 11))
    o12fix = ol.Oracle(*args, box=12, seed=9, flags=64)          # ORC_FLAG_FIX_INIT_RADIUS
    for o in (o12, o13, o11, o12fix):
        o.pm_init()
    assert np.array_equal(o12.norm4.view(np.uint32), o13.norm4.view(np.uint32))      # the random planes do not depend on the window
    assert np.array_equal(o12.c, o13.c) and not np.array_equal(o12.c, o11.c)
    assert np.array_equal(o12fix.c, o11.c)
    planes = o12.norm4.copy()
    assert np.array_equal(o12.pm_cost_planes(planes)[0], o11.pm_cost_planes(planes)[0])   # everything after init: radius 5
    # rectangular: each axis on its own
    oa, ob = ol.Oracle(*args, box=8, box_v=11, seed=9), ol.Oracle(*args, box=9, box_v=11, seed=9)
    oa.pm_init(); ob.pm_init()
    assert np.array_equal(oa.c, ob.c)


def test_refinement_step_count(small_scene):
    o = _orc(small_scene)
    n, dz = 0, o.max_disp / 2
    while dz >= 0.01:
        n += 1
        dz /= 10
    assert o.refine_steps() == n


def test_patchmatch_converges_on_the_synthetic_scene(mid_scene):
    sc = mid_scene
    o = _orc(sc, seed=1)
    o.pm_init()
    c0 = float(o.c.mean())
    o.pm_iterate(3)
    assert float(o.c.mean()) < 0.25 * c0
    d = o.compute_disp()[..., 3]
    gt = sc.gt_depth.numpy()
    assert (np.abs(d - gt) / gt < 0.02).mean() > 0.75
    # normals of accepted planes face the camera
    n = o.norm4[..., :3]
    assert (n[..., 2] < 0.2).mean() > 0.95


def test_sweep_is_jacobi_and_colour_restricted(small_scene):
    """a black sweep leaves every red pixel untouched and does not depend on the visiting order"""
    sc = small_scene
    o = _orc(sc, seed=2)
    o.pm_init()
    before_c, before_n = o.c.copy(), o.norm4.copy()
    o.pm_sweep(0)
    yy, xx = np.mgrid[0:sc.h, 0:sc.w]
    red = ((xx + yy) & 1) == 1
    assert np.array_equal(o.c[red], before_c[red]) and np.array_equal(o.norm4[red], before_n[red])
    assert (o.c[~red] <= before_c[~red]).all()              # greedy: cost never increases


def test_load_planes_compute_disp_round_trip(small_scene):
    sc = small_scene
    o = _orc(sc)
    n_world = np.ascontiguousarray((sc.gt_normal.numpy() @ sc.R[0]).astype(np.float32))
    o.load_planes(sc.gt_depth.numpy(), n_world)
    assert (o.c == 1.0).all()
    out = o.compute_disp()
    assert np.allclose(out[..., 3], sc.gt_depth.numpy(), rtol=2e-5)
    assert np.allclose(out[..., :3], n_world, atol=2e-6)
    o.c[0, 0] = 2.0
    assert o.compute_disp()[0, 0, 3] == 0.0                 # MAXCOST pixels export depth 0 (gipuma.cu:838-841)


def _merge_case(sc, o):
    """state + inputs that drive every branch of gipuma_compute_disp_final (reference gipuma.cu:757-808): fronto-parallel
    planes facing the camera at per-column-band depths, so that the expected output is known in closed form.
    Returns (resize4, text, expected depth, expected 'took resize' mask)."""
    h, w = sc.h, sc.w
    c0 = o.camera(0)
    f = float(c0.f)
    dmin, dmax = float(c0.depthMin), float(c0.depthMax)
    n = np.array([0.0, 0.0, -1.0], np.float32)       # camera looks along +z: n . view < 0
    mid = 0.5 * (dmin + dmax)
    # depth whose disparity differs from mid's by more / less than the 6-unit threshold of :778
    near6 = f / (f / mid + 8.0)
    close = f / (f / mid + 3.0)
    bands = [  # (depth_now, depth_resize, text, expected depth, resize taken)
        (mid, near6, 0.0, mid, False),          # text 0: never merged
        (mid, near6, 1.0, near6, True),         # text 1 and |disp_now - disp_org| = 8 > 6: take the upsampled plane
        (mid, close, 1.0, mid, False),          # text 1, difference 3 <= 6: keep
        (mid, near6, -1.0, near6, True),        # text -1: always take the upsampled plane
        (dmax * 1.5, mid, 0.0, dmax, False),    # beyond depthMax: plane offset re-derived at depthMax (:784-788)
        (dmin * 0.5, mid, 0.0, dmin, False),    # before depthMin (:789-793)
        (mid, dmax * 2.0, -1.0, dmax, True),    # merged AND clamped
    ]
    bw = w // len(bands)
    now = np.zeros((h, w, 4), np.float32)
    rs = np.zeros((h, w, 4), np.float32)
    text = np.zeros((h, w), np.float32)
    exp_d = np.zeros((h, w), np.float64)
    took = np.zeros((h, w), bool)
    for y in range(h):
        for x in range(w):
            d_now, d_rs, tx, d_exp, tk = bands[min(x // bw, len(bands) - 1)]
            now[y, x, :3] = n; now[y, x, 3] = o.getD(n, x, y, d_now)
            rs[y, x, :3] = n; rs[y, x, 3] = o.getD(n, x, y, d_rs)
            text[y, x] = tx; exp_d[y, x] = d_exp; took[y, x] = tk
    o.norm4[:] = now
    o.c[:] = 0.5
    o.c[0, :] = 2.0                                   # MAXCOST row: exported depth 0 (:802-805)
    return rs, text, exp_d, took


def test_compute_disp_final_known_answers(small_scene):
    """gipuma_compute_disp_final gipuma.cu:757-808: hi-res / upsampled plane merge by lines->text and the 6-disparity
    threshold, clamp of the merged plane into [depthMin, depthMax], world normal + depth (0 at MAXCOST) export"""
    sc = small_scene
    o = _orc(sc)
    rs, text, exp_d, took = _merge_case(sc, o)
    out = o.compute_disp_final(rs, text)
    d = out[..., 3]
    assert (d[0] == 0.0).all()
    assert np.allclose(d[1:], exp_d[1:], rtol=3e-5)
    assert np.allclose(o.depth, exp_d, rtol=3e-5)                       # lines->depth holds the merged depth (:797)
    # merged pixels carry the upsampled plane's bits unless the clamp re-derived the offset
    unclamped = took & (np.abs(exp_d - float(o.camera(0).depthMax)) > 1e-6)
    assert np.array_equal(o.norm4[unclamped], rs[unclamped])
    # normal export: R_orig^-1 n
    n_world = (np.asarray(o.camera(0).RorigInv, np.float64).reshape(3, 3) @ np.array([0, 0, -1.0]))
    assert np.allclose(out[..., :3], n_world, atol=2e-6)


def test_final_mode_sweep_skips_text_minus_one_and_side_writes(small_scene):
    """the kernels' `final == true` mode (gipuma.cu:856, :1063: return at text == -1; :559-562, :669-672: no ratio /
    beview writes); everywhere else it is the ordinary iteration"""
    sc = small_scene
    a, b = _orc(sc, seed=31), _orc(sc, seed=31)
    a.pm_init(); b.pm_init()
    text = np.zeros((sc.h, sc.w), np.float32)
    text[:, : sc.w // 3] = -1.0
    text[:, 2 * sc.w // 3:] = 1.0
    keep_c, keep_n = b.c.copy(), b.norm4.copy()
    b.ratio[:] = 7.0; b.beview[:] = 9
    a.pm_iterate(1)
    b.pm_iterate_final(1, text)
    frozen = text == -1.0
    assert np.array_equal(b.c[frozen], keep_c[frozen]) and np.array_equal(b.norm4[frozen], keep_n[frozen])
    assert (b.ratio == 7.0).all() and (b.beview == 9).all()
    assert (b.c[~frozen] <= keep_c[~frozen]).all() and (b.c[~frozen] < keep_c[~frozen]).mean() > 0.5
    # far from the frozen band (propagation reaches 23 px) both modes did exactly the same work
    far = np.zeros_like(frozen); far[:, sc.w // 3 + 48:] = True
    assert np.array_equal(a.c[far], b.c[far]) and np.array_equal(a.norm4[far], b.norm4[far])


def test_textureless_fill(small_scene):
    sc = small_scene
    h, w = sc.h, sc.w
    o = _orc(sc, seed=4)
    o.pm_init()
    labels = np.zeros((h, w), np.int32)
    labels[:, w // 2:] = 1
    o.set_regions(labels, np.array([1.0, -1.0], np.float32))
    plane = np.array([0.0, 0.0, 1.0, -6.0], np.float32)       # faces away from the camera: must be flipped
    o.set_region_planes(np.stack([np.zeros(4, np.float32), plane]))
    keep = o.norm4[:, : w // 2].copy()
    o.update_scale()
    assert np.array_equal(o.norm4[:, : w // 2], keep)
    assert np.allclose(o.norm4[:, w // 2:], [0, 0, -1, 6.0])
    assert (o.c[:, w // 2:] == 0).all() and (o.scale[:, w // 2:] == 1).all()
    c0 = o.camera(0)
    assert np.allclose(c0.f / o.depth[:, w // 2:], 6.0, rtol=1e-5)


# ---- TSAR refinement / SLIC oracle parts --------------------------------------------------------------
def test_cube_root_and_cielab_known_values():
    xs = np.linspace(0.009, 1.2, 500, dtype=np.float32)
    got = np.array([ol.pow_third(float(x)) for x in xs], np.float32)
    ref = np.power(xs.astype(np.longdouble), np.longdouble(np.float32(1.0) / np.float32(3.0))).astype(np.float32)   # pow(x, 1.0f / 3.0f), shared.h:41
    assert np.array_equal(got, ref)                 # correctly rounded (enumerated in tests/test_slic_reference_golden.py)
    white = ol.rgb2lab([255, 255, 255, 0])
    assert abs(white[0] - 100.0) < 0.01 and abs(white[1]) < 0.01 and abs(white[2]) < 0.01
    black = ol.rgb2lab([0, 0, 0, 0])
    assert abs(black[0]) < 1e-4
    red = ol.rgb2lab([0, 0, 255, 0])           # b, g, r order (gSLICr_seg_engine_shared.h:21-23)
    assert abs(red[0] - 53.24) < 0.05 and abs(red[1] - 80.09) < 0.1 and abs(red[2] - 67.20) < 0.1


def test_slic_on_two_flat_halves():
    h, w, S = 80, 120, 20
    img = np.zeros((h, w, 4), np.uint8)
    img[:, : w // 2] = (200, 50, 50, 0)
    img[:, w // 2:] = (30, 180, 220, 0)
    labels, lab, centers = ol.slic(img, S, 5, 5.0, 0, 0, want_centers=True)
    mw, mh = w // S, h // S
    assert labels.min() >= 0 and labels.max() < mw * mh
    # no superpixel straddles the colour edge
    left, right = set(np.unique(labels[:, : w // 2])), set(np.unique(labels[:, w // 2:]))
    assert not (left & right)
    # centres stay inside their half and their counts add up to the pixels their 3S x 48 windows can see
    assert np.all(centers[:, 7] > 0)


def test_ransac_recovers_a_plane_with_outliers():
    rng = np.random.default_rng(3)
    n_true = np.array([0.2, -0.3, -0.93])
    n_true /= np.linalg.norm(n_true)
    pts = rng.uniform(-1, 1, size=(4000, 3))
    d_true = 5.0                                                               # n.X + d = 0
    pts[:, 2] = -(d_true + n_true[0] * pts[:, 0] + n_true[1] * pts[:, 1]) / n_true[2]
    pts[:, 2] += rng.normal(0, 2e-4, 4000) / abs(n_true[2])
    pts[:600] += rng.uniform(-0.5, 0.5, size=(600, 3))                       # 15 % outliers
    plane, best = ol.ransac_points(pts.astype(np.float32), region_size=20 * (0.004 / 0.0003) ** 2)
    n = plane[:3] / np.linalg.norm(plane[:3])
    assert abs(float(n @ n_true)) > 0.9995
    assert abs(abs(plane[3]) - abs(d_true)) < 5e-3
    assert best > 0.7 * 4000
    # deterministic: same seed, same answer; different region id -> different draws
    plane2, best2 = ol.ransac_points(pts.astype(np.float32), region_size=20 * (0.004 / 0.0003) ** 2)
    assert np.array_equal(plane, plane2) and best == best2


def test_wmf_keeps_a_consistent_plane_and_flags_an_outlier(small_scene):
    sc = small_scene
    h, w = sc.h, sc.w
    o = _orc(sc)
    n_world = np.ascontiguousarray((sc.gt_normal.numpy() @ sc.R[0]).astype(np.float32))
    o.load_planes(sc.gt_depth.numpy(), n_world)          # ground-truth planes everywhere
    o.getview()                                          # lines->depth = f*b/depth
    o.scale[:] = 1.0
    # corrupt one interior pixel on the back plane: its plane disagrees with every neighbour's
    y, x = 8, 12
    bad = o.norm4[y, x].copy()
    bad[3] *= 1.6
    o.norm4[y, x] = bad
    for it in range(4):
        o.wmf_detect(it)
    assert o.scale[y, x] == 0.0
    assert o.scale.mean() > 0.8
    # the final pass refills it from its reliable neighbours
    o.set_regions(np.zeros((h, w), np.int32), np.array([1.0], np.float32))
    for it in range(3):
        o.wmf_fill(it)
    assert o.scale[y, x] == 1.0
    d = o.compute_disp()[y, x, 3]
    assert abs(d - sc.gt_depth.numpy()[y, x]) / sc.gt_depth.numpy()[y, x] < 0.02


# ---- row N2: weak-texture detection oracle ---------------------------------------------------------------
def _ccl(img, kind):
    import ctypes as C
    L = ol.lib()
    h, w = img.shape
    lab = np.empty((h, w), np.int32)
    fn = L.orc_connect_true if kind == "true" else L.orc_connect_literal
    fn.restype = C.c_int
    n = fn(img.ctypes.data_as(C.c_void_p), w, h, lab.ctypes.data_as(C.c_void_p), None, 0)
    return lab, n


def test_pyrdown_and_roberts_known_values():
    import ctypes as C
    L = ol.lib()
    src = np.full((20, 24), 77, np.uint8)
    dst = np.empty((10, 12), np.uint8)
    L.orc_pyrdown(src.ctypes.data_as(C.c_void_p), 24, 20, dst.ctypes.data_as(C.c_void_p))
    assert (dst == 77).all()                                   # the 5x5 kernel sums to 256
    ramp = np.tile(np.arange(24, dtype=np.uint8) * 8, (20, 1))
    L.orc_pyrdown(ramp.ctypes.data_as(C.c_void_p), 24, 20, dst.ctypes.data_as(C.c_void_p))
    assert list(dst[5, 1:11]) == [16 * k for k in range(1, 11)]   # linear ramps survive away from the border
    step = np.zeros((12, 12), np.uint8)
    step[:, 6:] = 10
    edge = np.empty_like(step)
    L.orc_roberts_threshold(step.ctypes.data_as(C.c_void_p), 12, 12, edge.ctypes.data_as(C.c_void_p))
    assert (edge[0] == 255).all() and (edge[:, 0] == 255).all()          # the 1-pixel frame is always "edge" (100*50 rule)
    assert (edge[1:-1, 5] == 255).all() and (edge[1:-1, 2] == 0).all() and (edge[1:-1, 8] == 0).all()
    big = np.zeros((4, 4), np.uint8)
    big[1, 1] = 182; big[2, 2] = 0; big[2, 1] = 181; big[1, 2] = 0      # sqrt(182^2 + 181^2) = 256.7 -> (uchar) 0
    L.orc_roberts_threshold(big.ctypes.data_as(C.c_void_p), 4, 4, edge[:4, :4].copy().ctypes.data_as(C.c_void_p))
    out = np.empty((4, 4), np.uint8)
    L.orc_roberts_threshold(big.ctypes.data_as(C.c_void_p), 4, 4, out.ctypes.data_as(C.c_void_p))
    assert out[1, 1] == 0                                       # the reference's uchar wrap makes this strong edge "flat"


def test_connected_components_numbering_and_the_reference_quirk():
    Z, E = 0, 255
    img = np.array([[Z, Z, E, Z, Z],
                    [E, E, E, E, Z],
                    [Z, Z, Z, E, Z],
                    [Z, E, Z, Z, Z]], np.uint8)
    lab, n = _ccl(img, "true")
    assert n == 3                                               # label 0 (edges) + two components
    assert lab[0, 0] == 1 and lab[0, 3] == 2 and lab[2, 0] == 2 and lab[0, 2] == 0
    assert np.array_equal(_ccl(img, "literal")[0], lab)
    # a shape where Connect()'s `connection[larger] = smaller` overwrites an earlier link (main.cpp:304-315):
    # provisional label 3 is first merged into 1, then re-parented to 2, and the 1-3 link is lost
    img = np.array([[Z, E, Z, E, Z, Z],
                    [Z, E, Z, E, Z, E],
                    [Z, E, Z, Z, Z, E],
                    [Z, Z, Z, E, E, E]], np.uint8)
    t, nt = _ccl(img, "true")
    assert nt == 2                                              # everything is one component
    l, nl = _ccl(img, "literal")
    assert nl >= nt                                             # the reference may keep more labels than there are components


def test_region_statistics_classification():
    import ctypes as C
    L = ol.lib()
    w4, h4 = 200, 120
    lab = np.zeros((h4, w4), np.int32)
    lab[10:90, 10:90] = 1            # 6400 px, bbox 79x79 = 6241 < 2*6400 -> true weak
    lab[100:110, 0:200] = 2          # 2000 px: below the 5000 px floor
    lab[0:5, 100:110] = 3
    n = 4
    text = np.empty(n, np.float32); size = np.empty(n, np.float32); cx = np.empty(n, np.int32); cy = np.empty(n, np.int32); cnt = np.empty(n, np.int32)
    L.orc_region_stats(lab.ctypes.data_as(C.c_void_p), w4, h4, n, *(a.ctypes.data_as(C.c_void_p) for a in (text, size, cx, cy, cnt)))
    assert list(text) == [1, -1, 1, 1] and size[1] == 79 and cnt[1] == 6400
    assert cx[1] == (sum(range(10, 90)) * 80 * 4) // 6400 and cy[2] == (sum(range(100, 110)) * 200 * 4) // 2000
    out = np.empty((h4 * 4 + 3, w4 * 4 + 2), np.int32)
    L.orc_upsample_labels(lab.ctypes.data_as(C.c_void_p), w4, h4, w4 * 4 + 2, h4 * 4 + 3, out.ctypes.data_as(C.c_void_p))
    assert out[40, 40] == 1 and out[-1, -1] == lab[-1, -1] and out[400, 0] == 2


# ---- row N3: fusion oracle -----------------------------------------------------------------------------
def _fusion_inputs(w=96, h=64, n_src=3):
    sc = synth.make_scene(w, h, n_src, seed=8, all_gt=True)
    depths = [d.numpy() for d, _ in sc.meta["gt_all"]]
    normals = [np.ascontiguousarray((n.numpy() @ sc.R[v]).astype(np.float32)) for v, (_, n) in enumerate(sc.meta["gt_all"])]   # world = R^T n_cam
    grays = [im.numpy() for im in sc.images]
    pairs = {v: [s for s in range(n_src + 1) if s != v] for v in range(n_src + 1)}
    return sc, depths, normals, grays, pairs


def test_fusion_of_ground_truth_maps_lands_on_the_surfaces():
    sc, depths, normals, grays, pairs = _fusion_inputs()
    pts = ol.fuse(depths, normals, grays, sc.K, sc.R, sc.t, pairs, num_consistent=2)
    assert len(pts) > 0.5 * sc.w * sc.h
    # every fused point is the mean of mutually consistent back-projections -> it lies on one of the analytic
    # surfaces (back plane n.X = d, slanted plane, sphere), up to the fp32 depth/pixel rounding of the inputs
    X = pts[:, :3].astype(np.float64)
    n0 = np.array([0.05, 0.02, -1.0]); n0 /= np.linalg.norm(n0)
    n1 = np.array([0.55, 0.10, -1.0]); n1 /= np.linalg.norm(n1)
    d_plane0 = np.abs(X @ n0 - (-1.6)); d_plane1 = np.abs(X @ n1 - 0.15)
    d_sph = np.abs(np.linalg.norm(X - np.array([-0.9, 0.35, -0.2]), axis=1) - 0.75)
    dist = np.minimum(np.minimum(d_plane0, d_plane1), d_sph)
    assert np.percentile(dist, 95) < 0.02
    assert np.allclose(np.linalg.norm(pts[:, 3:6], axis=1), 1.0, atol=1e-5)
    assert pts[:, 7].min() >= 2 and pts[:, 7].max() <= 3
    # used_list: without marking, every view re-emits the surface it shares with the others
    more = ol.fuse(depths, normals, grays, sc.K, sc.R, sc.t, pairs, num_consistent=2, used_list=0)
    assert len(more) > 1.5 * len(pts)
    # a stricter consistency count keeps fewer points; depth 0 pixels never fuse
    assert len(ol.fuse(depths, normals, grays, sc.K, sc.R, sc.t, pairs, num_consistent=3)) < len(pts)
    depths[0][:] = 0
    z = ol.fuse(depths, normals, grays, sc.K, sc.R, sc.t, pairs, num_consistent=2)
    assert not (z[:, 8] == 0).any()


def test_hough_closing_known_answers():
    """oracle of the boundary-closing step (deterministic stand-in for the reference's HoughLinesP, main.cpp:385-435):
    a horizontal boundary with a 10-pixel hole is closed by one line through the hole; a 30-pixel hole (> maxLineGap 18)
    and a short boundary (< minLineLength 160) are left alone"""
    import ctypes as C
    L = ol.lib()
    L.orc_hough_close.restype = C.c_int
    L.orc_connect_true.restype = C.c_int

    def run(hole, length):
        w, h = 400, 220
        edge = np.zeros((h, w), np.uint8)
        edge[0, :] = edge[-1, :] = 255
        edge[:, 0] = edge[:, -1] = 255
        x0 = (w - length) // 2
        edge[110, x0:x0 + length] = 255                     # a straight edge between two large flat areas
        edge[110, :x0] = 255 if length == w else edge[110, :x0]
        edge[110, 200:200 + hole] = 0                       # the hole
        lab0 = np.empty((h, w), np.int32)
        n0 = L.orc_connect_true(ol._p(edge), w, h, ol._p(lab0), None, 0)
        before = edge.copy()
        drawn = L.orc_hough_close(ol._p(edge), ol._p(lab0), n0, w, h)
        return drawn, before, edge

    drawn, before, after = run(10, 400)
    assert drawn >= 1 and (after[110, 200:210] == 255).all()       # hole closed
    assert (after >= before).all()                                   # closing only adds edge pixels
    drawn, before, after = run(30, 400)
    assert (after[110, 205:225] == 0).any()                          # a gap wider than maxLineGap is not bridged
    drawn, before, after = run(10, 120)
    assert np.array_equal(before, after) and (after[110, 200:210] == 0).all()   # a run shorter than minLineLength is not bridged
                                                                                  # (the frame's own long lines are re-drawn onto themselves)
