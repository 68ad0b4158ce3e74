"""GPU parity tests proper: the HIP path (through the C ABI, libtsar_hip.so) against the CPU oracle on
the same seeded inputs.

Two arithmetic modes are checked:
  strict (TSAR_FLAG_STRICT_DIV): every operation is a single IEEE fp32 op in the oracle's order, so the
      GPU must reproduce the oracle BIT FOR BIT — planes, costs, best views, after whole iterations.
  fast (default): the per-tap perspective divide uses v_rcp_f32 (1 ulp) instead of two IEEE divisions.
      Tolerance: |cost_gpu - cost_oracle| <= 1e-3 absolute on the same plane (cost lives in [0, 2]; the
      fp32 cancellation in var = E[x^2]-E[x]^2 amplifies a 1-ulp change of a tap position), and after
      one half-iteration from a common state >= 80 % of the swept pixels end on the bit-identical plane (~85 %
      measured; the late refinement steps perturb a plane so little that accept/reject is a near-tie that
      1e-6 of cost noise flips); whole runs are compared as distributions.
"""
import os
import numpy as np
import pytest

import oracle_lib as ol
from tsar_mvs_amd import api, synth

pytestmark = pytest.mark.gpu

FAST_COST_ATOL = 1e-3     # max; the distribution (p50 4e-6, p99 5e-5, max 3e-4 measured) is asserted in test_gpu_fast_mode.py


def _oracle(scene, **kw):
    return ol.Oracle([im.cpu().numpy() for im in scene.images], scene.K, scene.R, scene.t, scene.depth_min, scene.depth_max, **kw)


def _random_planes(scene, orc, seed):
    """random valid planes: random depth in range + random normal facing the camera"""
    rng = np.random.default_rng(seed)
    h, w = scene.h, scene.w
    planes = np.empty((h, w, 4), np.float32)
    for y in range(h):
        for x in range(w):
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            if n @ orc.view_vector(x, y) > 0:
                n = -n
            depth = rng.uniform(scene.depth_min, scene.depth_max)
            n = n.astype(np.float32)
            planes[y, x, :3] = n
            planes[y, x, 3] = orc.getD(n, x, y, depth)
    return planes


# boxes other than 11 (and more than two best views) run the general-window tap loop: weights from the shared table, lines in
# chunks of 4 / 5 / 6 taps (pm_core_lut.h).  (box, box_v): line lengths 1..16 taps cover every chunk length and padding count;
# even radii put the centre pixel among the taps; 25 and 31 are beyond what the per-thread weight table could hold.
@pytest.mark.parametrize("box,n_best,comb", [(11, 1, 1), (11, 2, 1), (7, 3, 1), (11, 1, 0), (19, 2, 1), (11, 1, 2), (11, 1, 3),
                                             (1, 1, 1), (3, 1, 1), (5, 2, 1), (9, 1, 1), (13, 1, 1), (15, 2, 1), (17, 1, 0), (21, 1, 1),
                                             (23, 4, 1), (25, 1, 1), (31, 2, 1), ((7, 13), 1, 1), ((19, 5), 2, 1), ((3, 27), 1, 1), (11, 3, 1), (11, 4, 1), (11, 5, 1), (9, 4, 0), (63, 1, 1), ((63, 9), 2, 1), ((5, 61), 1, 1), (45, 1, 1)])
def test_cost_planes_strict_bit_exact(small_scene, box, n_best, comb):
    sc = small_scene
    box, box_v = box if isinstance(box, tuple) else (box, box)
    orc = _oracle(sc, box=box, box_v=box_v, n_best=n_best, cost_comb=comb)
    m = api.matcher_from_scene(sc, box=box, box_v=box_v, n_best=n_best, cost_comb=comb, flags=api.FLAG_STRICT_DIV)
    for planes in (synth.gt_planes(sc).numpy(), _random_planes(sc, orc, 3)):
        c_ref, bv_ref, rt_ref = orc.pm_cost_planes(planes)
        c, bv, rt = m.pm_cost_planes(planes)
        assert np.array_equal(c, c_ref)
        assert np.array_equal(bv, bv_ref)
        assert np.array_equal(rt.view(np.uint32), rt_ref.view(np.uint32))
    m.close()


@pytest.mark.parametrize("box,n_best", [(11, 1), (19, 2), (7, 1), (9, 3), (11, 3), (11, 4), (11, 6), ((15, 9), 1), (25, 1), (63, 1)])
def test_cost_planes_fast_tolerance(small_scene, box, n_best):
    sc = small_scene
    box, box_v = box if isinstance(box, tuple) else (box, box)
    orc = _oracle(sc, box=box, box_v=box_v, n_best=n_best)
    m = api.matcher_from_scene(sc, box=box, box_v=box_v, n_best=n_best)
    for planes in (synth.gt_planes(sc).numpy(), _random_planes(sc, orc, 5)):
        c_ref, bv_ref, _ = orc.pm_cost_planes(planes)
        c, bv, _ = m.pm_cost_planes(planes)
        assert np.max(np.abs(c - c_ref)) <= FAST_COST_ATOL
        assert np.mean(bv == bv_ref) > 0.995
    m.close()


@pytest.mark.parametrize("box,n_best", [(11, 1), (19, 2), (7, 3), (13, 5)])
def test_8bit_texture_filter_mode_bit_exact(small_scene, box, n_best):
    """TSAR_FLAG_TEX_FILTER_8BIT (S3'): the CUDA texture unit's 8-fractional-bit filter weights; strict arithmetic, whole
    iterations and the reverse-direction cost.  The mode runs the general-window tap loop at every box (11 included)."""
    sc = small_scene
    fl = api.FLAG_TEX_FILTER_8BIT
    orc = _oracle(sc, seed=41, flags=fl, box=box, n_best=n_best)
    orc.pm_init()
    orc.pm_iterate(2)
    orc.lrdiff_op()
    orc.getview()
    m = api.matcher_from_scene(sc, seed=41, flags=fl | api.FLAG_STRICT_DIV, box=box, n_best=n_best)
    m.pm_init()
    m.pm_iterate(2)
    planes, cost, bv, _ = m.get_plane()
    assert np.array_equal(cost, orc.c) and np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32)) and np.array_equal(bv, orc.beview)
    m.lrdiff()
    m.getview()
    m.compute_disp()
    assert np.array_equal(m.get_result(("confid",))["confid"], orc.confid)
    m.close()
    plain = _oracle(sc, seed=41, box=box, n_best=n_best)
    plain.pm_init()
    plain.pm_iterate(2)
    assert not np.array_equal(plain.c, orc.c)              # the mode does change the numbers
    f = api.matcher_from_scene(sc, seed=41, flags=fl, box=box, n_best=n_best)        # fast arithmetic + 8-bit filter: within the fast-mode cost tolerance
    c_fast, _, _ = f.pm_cost_planes(orc.norm4.copy())
    f.close()
    # (a tap whose fraction sits on a rounding boundary of the 1/256 grid may land on the neighbouring step under the fast
    # mode's ~1-ulp different position: the cost then moves by one quantisation step of one tap, measured max 1.1e-3)
    assert np.max(np.abs(c_fast - orc.c)) <= 2e-3 and np.percentile(np.abs(c_fast - orc.c), 99) <= 2e-4


def test_float_image_path_matches_quad_path(small_scene):
    """non-integral images take the 4-load float path; on integral images both paths must agree exactly"""
    sc = small_scene
    orc = _oracle(sc)
    planes = _random_planes(sc, orc, 9)
    m = api.matcher_from_scene(sc, flags=api.FLAG_STRICT_DIV)
    c_quad, _, _ = m.pm_cost_planes(planes)
    m.close()
    # perturb one pixel of a source image by 0.5 -> library must fall back to the float path
    imgs = [im.clone() for im in sc.images]
    imgs[1][0, 0] += 0.5
    m2 = api.Matcher()
    m2.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max,
                                     flags=api.FLAG_STRICT_DIV, seed=2024))
    m2.set_views(imgs, sc.K, sc.R, sc.t)
    c_f, _, _ = m2.pm_cost_planes(planes)
    m2.close()
    orc2 = ol.Oracle([im.numpy() for im in imgs], sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max)
    c_ref, _, _ = orc2.pm_cost_planes(planes)
    assert np.array_equal(c_f, c_ref)
    # only windows that can touch source pixel (0,0) may differ between the two inputs
    assert np.mean(c_f != c_quad) < 0.2


# even boxes: gipuma_init_cu2 runs on radius box / 2 (gipuma.cu:693-694), the sweeps on (box - 1) / 2 (:858-859) — box 12 initialises
# on the general-window loop (radius 6) and sweeps on the box-11 loop, box 10 the other way round, (8, 11) differs on one axis only
@pytest.mark.parametrize("box,flags", [(11, 0), (12, 0), (10, 0), (12, api.FLAG_FIX_INIT_RADIUS), ((8, 11), 0), (20, 0), (2, 0)])
def test_init_strict_bit_exact(small_scene, box, flags):
    sc = small_scene
    box, box_v = box if isinstance(box, tuple) else (box, box)
    orc = _oracle(sc, seed=77, box=box, box_v=box_v, flags=flags)
    orc.pm_init()
    m = api.matcher_from_scene(sc, seed=77, box=box, box_v=box_v, flags=flags | api.FLAG_STRICT_DIV)
    m.pm_init()
    planes, cost, _, _ = m.get_plane()
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(cost, orc.c)
    if box % 2 == 0 or box_v % 2 == 0:
        # and the sweeps that follow run on the smaller window, from a state whose costs were NOT computed on it: a neighbour's
        # identical plane may score lower than the stored cost, so nothing may be skipped as "already held"
        orc.pm_iterate(1)
        m.pm_iterate(1)
        planes, cost, bv, _ = m.get_plane()
        assert np.array_equal(cost, orc.c)
        assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
        assert np.array_equal(bv, orc.beview)
    m.close()


@pytest.mark.parametrize("flags", [0, api.FLAG_FIX_DOWN_FAR_SEED | api.FLAG_FIX_RIGHT_FAR_CMP])
def test_iterations_strict_bit_exact(small_scene, flags):
    """two full red/black iterations: propagation (8 arms, snapshot reads) + refinement (Philox draws)"""
    sc = small_scene
    orc = _oracle(sc, seed=5, flags=flags)
    orc.pm_init()
    orc.pm_iterate(2)
    m = api.matcher_from_scene(sc, seed=5, flags=flags | api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(2)
    planes, cost, bv, rt = m.get_plane()
    assert np.array_equal(cost, orc.c)
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(bv, orc.beview)
    assert np.array_equal(rt.view(np.uint32), orc.ratio.view(np.uint32))
    m.close()


def test_half_sweeps_strict_bit_exact(small_scene):
    """the four reference kernels separately: black prop, black refine, red prop, red refine"""
    sc = small_scene
    orc = _oracle(sc, seed=9)
    orc.pm_init()
    m = api.matcher_from_scene(sc, seed=9, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    for colour, prop, refine in ((0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1)):
        orc.pm_sweep(colour, prop, refine)
        m.pm_sweep(colour, bool(prop), bool(refine))
        planes, cost, _, _ = m.get_plane()
        assert np.array_equal(cost, orc.c), (colour, prop, refine)
        assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32)), (colour, prop, refine)
    m.close()


def test_one_sweep_fast_agreement(mid_scene):
    """from the SAME state, one fast-mode half-iteration takes the same decisions as the oracle except
    where two hypotheses score within the fast-mode cost tolerance"""
    sc = mid_scene
    orc = _oracle(sc, seed=3)
    orc.pm_init()
    orc.pm_iterate(1)
    state_n, state_c = orc.norm4.copy(), orc.c.copy()
    m = api.matcher_from_scene(sc, seed=3)
    m.set_plane(state_n, state_c)
    m.set_sweep_counter(2)
    orc.pm_sweep(0)
    m.pm_sweep(0)
    planes, cost, _, _ = m.get_plane()
    same = np.all(planes.view(np.uint32) == orc.norm4.view(np.uint32), axis=-1)
    swept = (np.add.outer(np.arange(sc.h), np.arange(sc.w)) & 1) == 0          # colour 0: (x + y) even
    assert same[~swept].all()                                                   # the other colour is not touched
    assert same[swept].mean() >= 0.80, same[swept].mean()
    assert np.max(np.abs(cost - orc.c)[same]) <= FAST_COST_ATOL
    # pixels that end on another plane are checked in test_gpu_fast_mode.py::test_diverged_pixels_are_valid_patchmatch_steps
    # (re-scored by the oracle: valid steps, near-ties)
    m.close()


def test_iterations_fast_statistics(mid_scene):
    """whole runs: greedy accepts make trajectories diverge after a flipped decision, so full fast-mode
    runs are compared with the oracle as distributions: mean cost and the fraction of
    pixels converged to the analytic ground truth agree to within 1.5 points"""
    sc = mid_scene
    gt = sc.gt_depth.numpy()
    orc = _oracle(sc, seed=3)
    orc.pm_init()
    orc.pm_iterate(3)
    d_ref = orc.compute_disp()[..., 3]
    m = api.matcher_from_scene(sc, seed=3)
    m.pm_init()
    m.pm_iterate(3)
    m.compute_disp()
    res = m.get_result()
    d = res["depth"]
    conv_ref = (np.abs(d_ref - gt) / gt < 0.01).mean()
    conv = (np.abs(d - gt) / gt < 0.01).mean()
    assert abs(conv - conv_ref) < 0.015, (conv, conv_ref)
    assert abs(float(res["cost"].mean()) - float(orc.c.mean())) < 2e-3
    m.close()


def test_convergence_to_ground_truth(mid_scene):
    """property test at a size the oracle is not needed for: depth converges to the analytic scene"""
    sc = mid_scene
    m = api.matcher_from_scene(sc, seed=1)
    m.pm_init()
    m.pm_iterate(4)
    m.compute_disp()
    d = m.get_result(("depth",))["depth"]
    gt = sc.gt_depth.numpy()
    good = np.abs(d - gt) / gt < 0.02
    assert good.mean() > 0.80, good.mean()
    m.close()


def test_plane_depth_kernels_bit_exact(small_scene):
    sc = small_scene
    orc = _oracle(sc)
    gt_d = sc.gt_depth.numpy()
    n_cam = sc.gt_normal.numpy()
    Rt = sc.R[0].T
    n_world = np.ascontiguousarray((n_cam @ Rt.T).astype(np.float32))   # n_w = R^T n_c
    orc.load_planes(gt_d, n_world)
    m = api.matcher_from_scene(sc)
    m.load_planes(gt_d, n_world)
    planes, cost, _, _ = m.get_plane()
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(cost, orc.c)
    ref = orc.compute_disp()
    m.compute_disp()
    res = m.get_result()
    assert np.array_equal(res["depth"], ref[..., 3])
    assert np.array_equal(res["normal"], ref[..., :3])
    # the round trip reproduces the loaded depth / normals to fp32 accuracy
    assert np.allclose(res["depth"], gt_d, rtol=2e-5)
    assert np.allclose(res["normal"], n_world, atol=2e-6)
    # getview + depth_to_plane
    orc.getview()
    m.getview()
    orc.depth_to_plane()
    m.depth_to_plane()
    planes2, _, _, _ = m.get_plane()
    assert np.array_equal(planes2.view(np.uint32), orc.norm4.view(np.uint32))
    m.close()


def test_textureless_fill_bit_exact(small_scene):
    sc = small_scene
    h, w = sc.h, sc.w
    orc = _oracle(sc, seed=4)
    orc.pm_init()
    m = api.matcher_from_scene(sc, seed=4, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    labels = np.zeros((h, w), np.int32)
    labels[:, w // 3: 2 * w // 3] = 1
    labels[h // 2:, 2 * w // 3:] = 2
    text = np.array([1.0, -1.0, -1.0], np.float32)
    planes = np.array([[0.0, 0.0, 1.0, -5.0], [0.1, -0.2, -0.97, 6.0], [0, 0, 0, 0]], np.float32)
    planes[1, :3] /= np.linalg.norm(planes[1, :3])
    planes[2] = planes[1] * np.array([1, 1, 1, 1.1], np.float32)
    for o in (orc, m):
        o.set_regions(labels, text)
        o.set_region_planes(planes)
    orc.fake_depth()
    fd = m.fake_depth()
    msk = labels > 0
    assert np.array_equal(fd[msk], orc.fakedepth[msk])
    orc.update_scale()
    ref = orc.compute_disp()
    m.fill_textureless()
    res = m.get_result()
    assert np.array_equal(res["depth"], ref[..., 3])
    assert np.array_equal(res["normal"], ref[..., :3])
    assert np.array_equal(res["cost"], orc.c)
    m.close()


def test_lrdiff_confidence(small_scene):
    sc = small_scene
    orc = _oracle(sc, seed=6)
    orc.pm_init()
    orc.pm_iterate(1)
    orc.lrdiff_op()
    orc.getview()
    m = api.matcher_from_scene(sc, seed=6, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(1)
    m.lrdiff()
    m.getview()
    m.compute_disp()
    res = m.get_result()
    assert np.array_equal(res["confid"], orc.confid)
    m.close()


def test_compute_disp_final_bit_exact(small_scene):
    """gipuma_compute_disp_final (reference gipuma.cu:757-808) on the case that drives every branch: text 0 / 1 / -1,
    disparity difference above and below the threshold of 6, clamps at depthMin and depthMax, MAXCOST export"""
    from test_oracle_known_answers import _merge_case
    sc = small_scene
    orc = _oracle(sc)
    rs, text, exp_d, took = _merge_case(sc, orc)
    m = api.matcher_from_scene(sc)
    m.set_plane(orc.norm4.copy(), orc.c.copy())
    ref = orc.compute_disp_final(rs, text)
    m.compute_disp_final(rs, text)
    res = m.get_result(("depth", "normal", "cost"))
    assert np.array_equal(res["depth"], ref[..., 3])
    assert np.array_equal(res["normal"].view(np.uint32), ref[..., :3].view(np.uint32))
    planes, _, _, _ = m.get_plane()
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))       # merged / clamped camera-frame planes
    assert took.any() and (~took).any() and (res["depth"][0] == 0).all()
    # a second case on a PatchMatch state: random text, upsampled planes = the state shifted by one pixel
    orc2 = _oracle(sc, seed=14)
    orc2.pm_init(); orc2.pm_iterate(1)
    rng = np.random.default_rng(3)
    text2 = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(sc.h, sc.w))
    rs2 = np.roll(orc2.norm4, 1, axis=1).copy()
    m.set_plane(orc2.norm4.copy(), orc2.c.copy())
    ref2 = orc2.compute_disp_final(rs2, text2)
    m.compute_disp_final(rs2, text2)
    res2 = m.get_result(("depth", "normal"))
    assert np.array_equal(res2["depth"], ref2[..., 3])
    assert np.array_equal(res2["normal"].view(np.uint32), ref2[..., :3].view(np.uint32))
    m.close()


def test_final_mode_iterations_bit_exact(small_scene):
    """the kernels' `final == true` mode (reference gipuma.cu:856, :1063, :559-562, :669-672), strict arithmetic"""
    sc = small_scene
    rng = np.random.default_rng(8)
    text = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(sc.h, sc.w), p=[0.3, 0.4, 0.3])
    text[:, : sc.w // 4] = -1.0
    orc = _oracle(sc, seed=21)
    orc.pm_init()
    orc.pm_iterate(1)
    m = api.matcher_from_scene(sc, seed=21, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(1)
    orc.pm_iterate_final(2, text)
    m.pm_iterate_final(2, text)
    planes, cost, bv, rt = m.get_plane()
    assert np.array_equal(cost, orc.c)
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(bv, orc.beview) and np.array_equal(rt.view(np.uint32), orc.ratio.view(np.uint32))
    # ... and a normal iteration afterwards continues from it bit for bit
    orc.pm_iterate(1)
    m.pm_iterate(1)
    planes, cost, _, _ = m.get_plane()
    assert np.array_equal(cost, orc.c) and np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    m.close()


def test_view_subset_change_mid_run_bit_exact(mid_scene):
    """changing the scored views after the state exists: the stored costs no longer belong to the subset, so the sweep must
    re-score a neighbour that carries the pixel's own plane like the reference does (gipuma.cu:540-556), not skip it"""
    sc = mid_scene
    orc = _oracle(sc, seed=4, subset=[1, 2])
    orc.pm_init(); orc.pm_iterate(1)
    m = api.matcher_from_scene(sc, seed=4, flags=api.FLAG_STRICT_DIV, subset=[1, 2])
    m.pm_init(); m.pm_iterate(1)
    orc.set_subset([3, 4, 1])
    m.set_view_subset([3, 4, 1])
    orc.pm_iterate(1)
    m.pm_iterate(1)
    planes, cost, bv, _ = m.get_plane()
    assert np.array_equal(cost, orc.c)
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(bv, orc.beview)
    m.close()


def test_device_buffers_and_final_mode_with_device_inputs(small_scene):
    """the ABI pieces tsar_gipuma --all --fuse is built from (device alloc / host write / peer copy, here on one device), and
    tsar_pm_iterate_final fed from device memory"""
    import ctypes as C
    import torch
    L = api.load_library()
    n = 1 << 16
    src = np.arange(n, dtype=np.float32)
    a = L.tsar_device_alloc(0, n * 4)
    b = L.tsar_device_alloc(0, n * 4)
    assert a and b
    assert L.tsar_device_write(0, a, src.ctypes.data_as(C.c_void_p), n * 4) == api.TSAR_OK
    assert L.tsar_peer_copy(0, b, 0, a, n * 4) == api.TSAR_OK
    back = torch.empty(n, dtype=torch.float32, device="cuda")
    assert L.tsar_peer_copy(0, C.c_void_p(back.data_ptr()), 0, b, n * 4) == api.TSAR_OK
    assert np.array_equal(back.cpu().numpy(), src)
    assert L.tsar_peer_copy(0, None, 0, a, 4) == api.TSAR_ERR_INVALID
    L.tsar_device_free(0, a)
    L.tsar_device_free(0, b)
    pinned = api.pinned_empty((4, 5), np.float32)
    pinned[...] = 3.0
    view = pinned[1:]                              # a view keeps the page-locked block alive
    del pinned
    assert float(view.sum()) == 45.0
    # final-mode iterations with lines->text living on the device
    sc = small_scene
    text = np.zeros((sc.h, sc.w), np.float32)
    text[::3] = -1.0
    orc = _oracle(sc, seed=2)
    orc.pm_init()
    orc.pm_iterate_final(1, text)
    m = api.matcher_from_scene(sc, seed=2, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate_final(1, torch.from_numpy(text).cuda())
    planes, cost, _, _ = m.get_plane()
    assert np.array_equal(cost, orc.c) and np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    m.close()


def test_error_codes(small_scene):
    m = api.Matcher()
    with pytest.raises(api.TsarError) as e:
        m.pm_init()
    assert e.value.code == api.TSAR_ERR_STATE
    with pytest.raises(api.TsarError) as e:
        m.set_params(api.default_params(depth_min=5.0, depth_max=1.0))
    assert e.value.code == api.TSAR_ERR_INVALID
    # limits that used to fail late or silently: more best views than the 32-entry cost vector, a box beyond 63
    for bad in (dict(n_best=33), dict(box_hsize=65, box_vsize=11), dict(box_hsize=11, box_vsize=0)):
        with pytest.raises(api.TsarError) as e:
            m.set_params(api.default_params(**bad))
        assert e.value.code == api.TSAR_ERR_INVALID, bad
    m.set_params(api.default_params(box_hsize=63, box_vsize=63))      # 8-bit imagery: any box (shared weight table)
    m.close()
    sc = small_scene
    # images that are not an 8-bit decode keep S weights per thread in LDS: boxes beyond 23 are refused when the views arrive
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=25, box_vsize=25, depth_min=sc.depth_min, depth_max=sc.depth_max))
    with pytest.raises(api.TsarError) as e:
        m.set_views([im + 0.25 for im in sc.images], sc.K, sc.R, sc.t)
    assert e.value.code == api.TSAR_ERR_INVALID
    assert "not 8-bit" in str(e.value)
    m.set_params(api.default_params(box_hsize=23, box_vsize=23, depth_min=sc.depth_min, depth_max=sc.depth_max))
    m.set_views([im + 0.25 for im in sc.images], sc.K, sc.R, sc.t)   # the largest square box of the float path
    # box 24 on float imagery: the sweeps' window (radius 11) fits, gipuma_init_cu2's own (radius 12, gipuma.cu:693-694) does not
    m.set_params(api.default_params(box_hsize=24, box_vsize=24, depth_min=sc.depth_min, depth_max=sc.depth_max))
    with pytest.raises(api.TsarError) as e:
        m.set_views([im + 0.25 for im in sc.images], sc.K, sc.R, sc.t)
    assert e.value.code == api.TSAR_ERR_INVALID
    m.set_params(api.default_params(box_hsize=24, box_vsize=24, depth_min=sc.depth_min, depth_max=sc.depth_max, flags=api.FLAG_FIX_INIT_RADIUS))
    m.set_views([im + 0.25 for im in sc.images], sc.K, sc.R, sc.t)
    m.close()
    # 8-bit imagery, a rectangular box whose radii have mixed parity: 63 x 61 has 202 distinct tap distances, more than the shared
    # weight table holds — refused with that reason; the same box on the reference view alone (refinement operators only) is fine
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=63, box_vsize=61, depth_min=sc.depth_min, depth_max=sc.depth_max))
    with pytest.raises(api.TsarError) as e:
        m.set_views(sc.images, sc.K, sc.R, sc.t)
    assert e.value.code == api.TSAR_ERR_INVALID and "distinct tap distances" in str(e.value)
    m.set_views(sc.images[:1], sc.K[:1], sc.R[:1], sc.t[:1])
    m.close()
    m = api.matcher_from_scene(sc)
    with pytest.raises(api.TsarError) as e:
        m.set_view_subset(list(range(1, 3)) * 17)                      # 34 entries
    assert e.value.code == api.TSAR_ERR_INVALID
    # region labels index device tables: out-of-range labels are refused, host and device buffers alike, and leave the
    # context usable
    import torch
    labels = np.zeros((sc.h, sc.w), np.int32)
    text = np.array([1.0, -1.0], np.float32)
    for bad_value in (2, -1, 1 << 20):
        lb = labels.copy()
        lb[sc.h // 2, sc.w // 2] = bad_value
        with pytest.raises(api.TsarError) as e:
            m.set_regions(lb, text)
        assert e.value.code == api.TSAR_ERR_INVALID, bad_value
        with pytest.raises(api.TsarError) as e:
            m.set_regions(torch.from_numpy(lb).cuda(), text)
        assert e.value.code == api.TSAR_ERR_INVALID, bad_value
    labels[:, sc.w // 2:] = 1
    m.set_regions(torch.from_numpy(labels).cuda(), text)
    m.set_region_planes(np.array([[0, 0, 0, 0], [0, 0, -1, 5.0]], np.float32))
    m.pm_init()
    m.fill_textureless()
    assert (m.get_result(("cost",), pinned=True)["cost"][:, sc.w // 2:] == 0).all()
    m.close()


# ---- TSAR refinement: weighted median filter, region RANSAC, SLIC ---------------------------------
def _prepared_pair(sc, seed, flags=0):
    """oracle + matcher in the same converged-ish state with lines->depth filled (getview)"""
    orc = _oracle(sc, seed=seed, flags=flags)
    orc.pm_init()
    orc.pm_iterate(2)
    orc.getview()
    m = api.matcher_from_scene(sc, seed=seed, flags=flags | api.FLAG_STRICT_DIV)
    m.set_plane(orc.norm4.copy(), orc.c.copy())
    m.getview()
    return orc, m


def test_reference_only_context_serves_the_refinement_operators(small_scene):
    """tsar_set_views with the reference view alone: none of the textureless-refinement operators reads a source image, so
    they give the same result as with all views loaded; the matching entry points refuse"""
    sc = small_scene
    rng = np.random.default_rng(2)
    depth = rng.uniform(sc.depth_min, sc.depth_max, (sc.h, sc.w)).astype(np.float32)
    normal = rng.normal(size=(sc.h, sc.w, 3)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=-1, keepdims=True)
    good = (rng.uniform(size=(sc.h, sc.w)) < 0.7).astype(np.float32)
    outs = []
    for n in (1, len(sc.images)):
        m = api.Matcher()
        m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max))
        m.set_views(sc.images[:n], sc.K[:n], sc.R[:n], sc.t[:n])
        m.load_planes(depth, normal)
        m.set_reliable_mask(good)
        labels, text, size = m.detect_weak_texture()
        labels2, text2, size2 = m.detect_weak_texture()          # second call: temporaries recycled from the scratch arena
        assert np.array_equal(labels, labels2) and np.array_equal(text, text2) and np.array_equal(size, size2)
        m.getview()
        planes, ratio = m.ransac_regions()
        m.fake_depth()
        m.fill_textureless()
        res = m.get_result(("depth", "normal"))
        outs.append((labels, text, size, planes, ratio, res["depth"], res["normal"]))
        if n == 1:
            for call in (m.pm_init, lambda: m.pm_iterate(1), m.lrdiff):
                with pytest.raises(api.TsarError) as e:
                    call()
                assert e.value.code == api.TSAR_ERR_STATE
        m.close()
    for a, b in zip(*outs):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_wmf_detect_and_fill_bit_exact(small_scene):
    sc = small_scene
    h, w = sc.h, sc.w
    orc, m = _prepared_pair(sc, 8)
    rng = np.random.default_rng(0)
    scale = (rng.uniform(size=(h, w)) < 0.7).astype(np.float32)
    orc.scale[:] = scale
    m.set_reliable_mask(scale)
    labels = np.zeros((h, w), np.int32)
    labels[:, w // 2:] = 1
    text = np.array([1.0, -1.0], np.float32)
    orc.set_regions(labels, text)
    m.set_regions(labels, text)
    for it in range(4):
        orc.wmf_detect(it)
    m.wmf(4, False)
    # the reliability map the four detection passes leave in lines->scale
    got_scale = m.get_reliable_mask()
    assert np.array_equal(got_scale, orc.scale)
    assert 0.02 < (got_scale != scale).mean() < 0.98      # the passes did change it
    for it in range(3):
        orc.wmf_fill(it)
    m.wmf(3, True)
    planes, _, _, _ = m.get_plane()
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    m.compute_disp()
    ref = orc.compute_disp()
    assert np.array_equal(m.get_result(("depth",))["depth"], ref[..., 3])
    m.close()


def test_wmf_megapixel_all_tap_grids_interior_and_border():
    """1216 x 832: at every one of the four detection tap grids (radius 80 / 40 / 20 / 10) and the three fill grids most waves have their
    whole windows inside the image and read their taps through the scalar-offset buffer loads, the rest through the clamped path
    (wmf_kernels.hip tap_window); two lanes per pixel; against the oracle's bubble sort, bit for bit (reliability map after the
    four detection passes, planes / depths / reliability after the three fill passes)"""
    sc = synth.make_scene(1216, 832, 2, seed=14)
    h, w = sc.h, sc.w
    orc, m = _prepared_pair(sc, 6)
    rng = np.random.default_rng(4)
    scale = (rng.uniform(size=(h, w)) < 0.7).astype(np.float32)
    orc.scale[:] = scale
    m.set_reliable_mask(scale)
    labels = np.zeros((h, w), np.int32)
    labels[:, w // 2:] = 1
    text = np.array([1.0, -1.0], np.float32)
    orc.set_regions(labels, text)
    m.set_regions(labels, text)
    for it in range(4):
        orc.wmf_detect(it)
    m.wmf(4, False)
    got = m.get_reliable_mask()
    assert np.array_equal(got, orc.scale)
    assert 0.02 < (got != scale).mean() < 0.98
    for it in range(3):
        orc.wmf_fill(it)
    m.wmf(3, True)
    assert np.array_equal(m.get_plane()[0].view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(m.get_reliable_mask(), orc.scale)
    m.close()


def test_wmf_odd_size_partial_last_workgroup():
    """173 x 61 = 10 553 pixels: not a multiple of the 32 pixels a workgroup serves (two lanes per pixel), rows that do not align
    with waves, and — at the fine tap grids — waves whose windows are all inside the image next to waves at the border (the two
    addressing paths of wmf_kernels.hip): detection and fill bit for bit"""
    sc = synth.make_scene(173, 61, 2, seed=12)
    h, w = sc.h, sc.w
    orc, m = _prepared_pair(sc, 5)
    rng = np.random.default_rng(3)
    scale = (rng.uniform(size=(h, w)) < 0.75).astype(np.float32)
    orc.scale[:] = scale
    m.set_reliable_mask(scale)
    labels = np.zeros((h, w), np.int32)
    text = np.array([1.0], np.float32)
    orc.set_regions(labels, text)
    m.set_regions(labels, text)
    for it in range(4):
        orc.wmf_detect(it)
    m.wmf(4, False)
    assert np.array_equal(m.get_reliable_mask(), orc.scale)
    for it in range(3):
        orc.wmf_fill(it)
    m.wmf(3, True)
    assert np.array_equal(m.get_plane()[0].view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(m.get_reliable_mask(), orc.scale)
    m.close()


def test_wmf_on_float_imagery_keeps_the_direct_weights(small_scene):
    """the filter's bilateral weight comes from two tables on 8-bit imagery (spatial factor per tap slot, colour factor per integer
    difference: wmf_kernels.hip WmfLds); images that are not an 8-bit decode take the direct form exp(-cd / 9): same oracle, bit for bit"""
    import dataclasses
    sc = dataclasses.replace(small_scene, images=[im + 0.25 for im in small_scene.images])
    h, w = sc.h, sc.w
    orc, m = _prepared_pair(sc, 8)
    rng = np.random.default_rng(0)
    scale = (rng.uniform(size=(h, w)) < 0.7).astype(np.float32)
    orc.scale[:] = scale
    m.set_reliable_mask(scale)
    labels = np.zeros((h, w), np.int32)
    text = np.array([1.0], np.float32)
    orc.set_regions(labels, text)
    m.set_regions(labels, text)
    for it in range(4):
        orc.wmf_detect(it)
    m.wmf(4, False)
    assert np.array_equal(m.get_reliable_mask(), orc.scale)
    for it in range(2):
        orc.wmf_fill(it)
    m.wmf(2, True)
    assert np.array_equal(m.get_plane()[0].view(np.uint32), orc.norm4.view(np.uint32))
    m.close()


def test_wmf_infinite_values_at_reliable_taps_sort_like_the_reference(small_scene):
    """an infinite normal component inside a reliable window: the 64-bit (value, slot) sort key classifies infinities before the
    slot bits go into the mantissa (they would make a signalling NaN, which v_min_f64 quiets instead of replacing, and the
    network would lose a slot): -inf first, +inf last, ties in tap order, like the reference's `>` bubble sort (gipuma.cu:1560-1616)"""
    sc = small_scene
    h, w = sc.h, sc.w
    orc, m = _prepared_pair(sc, 8)
    rng = np.random.default_rng(5)
    planes = orc.norm4.copy()
    r = rng.uniform(size=(h, w))
    planes[..., 0][r < 0.03] = np.inf
    planes[..., 0][(r >= 0.03) & (r < 0.06)] = -np.inf
    planes[..., 2][(r >= 0.06) & (r < 0.08)] = np.inf
    planes[..., 1][(r >= 0.08) & (r < 0.09)] = -np.inf
    orc.norm4[:] = planes                         # lines->depth stays as getview left it: the depth list is finite
    m.set_plane(planes, orc.c.copy())
    scale = (rng.uniform(size=(h, w)) < 0.8).astype(np.float32)
    orc.scale[:] = scale
    m.set_reliable_mask(scale)
    labels = np.zeros((h, w), np.int32)
    text = np.array([1.0], np.float32)
    orc.set_regions(labels, text)
    m.set_regions(labels, text)
    for it in range(4):
        orc.wmf_detect(it)
    m.wmf(4, False)
    got = m.get_reliable_mask()
    assert np.array_equal(got, orc.scale)
    assert 0.02 < (got != scale).mean() < 0.98
    for it in range(2):
        orc.wmf_fill(it)
    m.wmf(2, True)
    got_planes = m.get_plane()[0]
    same = (got_planes.view(np.uint32) == orc.norm4.view(np.uint32)) | (np.isnan(got_planes) & np.isnan(orc.norm4))
    assert same.all(), int((~same).sum())
    m.close()


@pytest.mark.parametrize("knob,value", [("TSAR_RANSAC_CHAIN", "8"), ("TSAR_RANSAC_CHAIN", "4"), ("TSAR_RANSAC_CHAIN", "16"),
                                        ("TSAR_RANSAC_WGS", "1"), ("TSAR_RANSAC_WGS", "2"), ("TSAR_RANSAC_WGS", "4"), ("TSAR_RANSAC_WGS", "7"),
                                        ("TSAR_RANSAC_LOOKAHEAD", "1"), ("TSAR_RANSAC_LOOKAHEAD", "2"), ("TSAR_RANSAC_LOOKAHEAD", "3"),
                                        ("TSAR_RANSAC_WGS+TSAR_RANSAC_CHAIN", "1+4"), ("TSAR_RANSAC_WGS+TSAR_RANSAC_CHAIN", "1+16"),
                                        ("TSAR_RANSAC_FORCE_FALLBACK", "1"), ("TSAR_RANSAC_POLL_LIMIT", "0"), ("TSAR_RANSAC_COOPERATIVE", "0")])
def test_ransac_regions_bit_exact(mid_scene, monkeypatch, knob, value):
    """how stage 2 shares passes over the points between perturbation steps (ransac_kernels.hip): a speculative chain of K steps
    (the default, K = 8; TSAR_RANSAC_CHAIN picks the length in the multi-workgroup kernel, with TSAR_RANSAC_WGS=1 in the
    single-workgroup one) or a tree of the accept / reject histories of G steps; every setting must replay the reference's
    sequential accept order exactly.  TSAR_RANSAC_FORCE_FALLBACK=1 / TSAR_RANSAC_POLL_LIMIT=0: the give-up path of the
    multi-workgroup kernel (flag pre-set / raised by the first workgroup that has to wait) — the single-workgroup kernel then
    produces the same bits.  TSAR_RANSAC_COOPERATIVE=0: a plain launch instead of the cooperative one."""
    for k, v in zip(knob.split("+"), value.split("+")):
        monkeypatch.setenv(k, v)
    sc = mid_scene
    h, w = sc.h, sc.w
    orc, m = _prepared_pair(sc, 12)
    # regions from the analytic primitives: back plane (0) textured, slanted plane (1) and sphere (2) "textureless"
    labels = sc.gt_prim.numpy().astype(np.int32)
    text = np.array([1.0, -1.0, -1.0], np.float32)
    size = np.array([(labels == k).sum() for k in range(3)], np.float32)
    gt = sc.gt_depth.numpy()
    # reliable = pixels whose depth converged (like weak.png in the reference)
    d = orc.compute_disp()[..., 3]
    scale = (np.abs(d - gt) / gt < 0.01).astype(np.float32)
    orc.scale[:] = scale
    m.set_reliable_mask(scale)
    orc.set_regions(labels, text, size)
    m.set_regions(labels, text, size)
    planes_ref, ratio_ref = orc.ransac_regions()
    planes, ratio = m.ransac_regions()
    assert np.array_equal(planes[1:].view(np.uint32), planes_ref[1:].view(np.uint32))
    assert np.array_equal(ratio, ratio_ref)
    # again in the same context: the second call takes its temporaries from the context's scratch arena (recycled, not zeroed)
    planes2, ratio2 = m.ransac_regions()
    assert np.array_equal(planes2.view(np.uint32), planes.view(np.uint32)) and np.array_equal(ratio2, ratio)
    # the slanted plane is recovered: normal parallel to the analytic one
    n_gt = sc.gt_normal.numpy()[labels == 1].mean(0)
    n_gt /= np.linalg.norm(n_gt)
    n = planes[1, :3] / np.linalg.norm(planes[1, :3])
    assert abs(float(n @ n_gt)) > 0.995
    assert ratio[1] > 0.05    # inlier band 0.0003*sqrt(size/20) is tight for a 192x128 depth map
    # fill the textureless regions with the fitted planes and export
    orc.fake_depth(); orc.update_scale()
    ref = orc.compute_disp()
    m.fake_depth(); m.fill_textureless()
    res = m.get_result()
    assert np.array_equal(res["depth"], ref[..., 3])
    m.close()


def _ransac_child_inputs(sc, tmp_path):
    """the oracle's converged state + reliability mask + regions of the RANSAC tests as an .npz for tests/ransac_contexts_child.py,
    and the oracle's fit as the expected answer"""
    labels = sc.gt_prim.numpy().astype(np.int32)
    text = np.array([1.0, -1.0, -1.0], np.float32)
    size = np.array([(labels == k).sum() for k in range(3)], np.float32)
    gt = sc.gt_depth.numpy()
    orc = _oracle(sc, seed=12)
    orc.pm_init()
    orc.pm_iterate(2)
    orc.getview()
    d = orc.compute_disp()[..., 3]
    scale = (np.abs(d - gt) / gt < 0.01).astype(np.float32)
    state = str(tmp_path / "state.npz")
    np.savez(state, w=sc.w, h=sc.h, n_src=len(sc.images) - 1, scene_seed=11, seed=12, norm4=orc.norm4, c=orc.c, scale=scale, labels=labels, text=text, size=size)
    orc.scale[:] = scale
    orc.set_regions(labels, text, size)
    planes_ref, ratio_ref = orc.ransac_regions()
    return state, planes_ref, ratio_ref


def _start_ransac_child(state, out, threads, reps, start_at=0.0):
    import subprocess
    import sys
    env = dict(os.environ)
    env["PYTHONFAULTHANDLER"] = "1"
    return subprocess.Popen([sys.executable, "-X", "faulthandler", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ransac_contexts_child.py"),
                             state, out, "--threads", str(threads), "--reps", str(reps), "--start-at", repr(start_at)],
                            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _check_ransac_child(proc, out, planes_ref, ratio_ref, limit_s):
    so, se = proc.communicate(timeout=300)
    # the exit status covers interpreter and HIP-runtime teardown: a crash there (the round-3 core dumps came after pytest's
    # "passed") is a negative status / 134 / 139 here, with faulthandler's report in stderr
    assert proc.returncode == 0, "child exited with %r\n--- stdout\n%s\n--- stderr\n%s" % (proc.returncode, so[-2000:], se[-6000:])
    got = np.load(out)
    for k in range(got["planes"].shape[0]):
        for planes, ratio in zip(got["planes"][k], got["ratio"][k]):
            assert np.array_equal(planes[1:].view(np.uint32), planes_ref[1:].view(np.uint32)) and np.array_equal(ratio, ratio_ref)
    assert float(got["seconds"]) < limit_s, "the region fits took %.1f s: a spin barrier ran into its limit" % float(got["seconds"])     # ~0.1 s when nothing stalls


def test_ransac_two_contexts_share_the_device(mid_scene, tmp_path):
    """two contexts of ONE process on one device (tsar_gipuma --workers=2) fit their regions at the same time from two host threads:
    the multi-workgroup stage 2 needs its workgroups resident together, which a neighbour's kernels can prevent — the cooperative
    launch (one at a time per process, ransac_kernels.hip) or, failing that, the bounded wait and the single-workgroup kernel
    must give the oracle's planes without stalling.  Runs in a child interpreter so that its exit status — teardown included —
    is part of the assertion ("the round-3 exit-time core dumps")."""
    state, planes_ref, ratio_ref = _ransac_child_inputs(mid_scene, tmp_path)
    out = str(tmp_path / "out.npz")
    _check_ransac_child(_start_ransac_child(state, out, threads=2, reps=6), out, planes_ref, ratio_ref, 5.0)


def test_ransac_two_processes_share_the_device(mid_scene, tmp_path):
    """the form the per-process cooperative-launch lock does NOT cover: two PROCESSES on one device (tsar_gipuma one process per
    view, ranks sharing a GPU), two contexts each, all fitting at the same time.  Each process's runtime places its cooperative
    grid against its own occupancy only, so co-residency across processes rests on the second line — the bounded wait and the
    single-workgroup fallback: same planes, no stall, clean exit of both."""
    import time
    state, planes_ref, ratio_ref = _ransac_child_inputs(mid_scene, tmp_path)
    outs = [str(tmp_path / ("out%d.npz" % k)) for k in range(2)]
    start_at = time.time() + 20.0              # both children have imported torch-free api + built their contexts by then, or start late: still valid
    procs = [_start_ransac_child(state, outs[k], threads=2, reps=6, start_at=start_at) for k in range(2)]
    for k in range(2):
        _check_ransac_child(procs[k], outs[k], planes_ref, ratio_ref, 10.0)


@pytest.mark.parametrize("S,iters,conn,space", [(20, 5, 0, 0), (16, 3, 1, 0), (20, 2, 0, 1), (12, 2, 1, 2)])
def test_slic_labels_exact(S, iters, conn, space):
    rng = np.random.default_rng(S)
    h, w = 150, 212
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (127 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)
    img[..., 1] = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
    img[..., 2] = (rng.integers(0, 30, size=(h, w)) + 100 * ((xx // 40 + yy // 30) % 2)).astype(np.uint8)
    m = api.Matcher()
    st = api.SlicSettings(S, iters, 5.0, conn, space)
    got = m.slic(img, st)
    ref = ol.slic(img, S, iters, 5.0, conn, space)
    assert np.array_equal(got, ref)
    assert got.min() >= 0 and got.max() < (w // S) * (h // S)
    m.close()


def test_slic_labels_exact_at_the_quarter_resolution_of_an_eth3d_view():
    """BASELINE configs[3]: gSLICr runs on the 1512 x 1008 quarter-resolution image, superpixel size 20, 5 iterations, CIELAB
    (main.cpp:1506-1517)"""
    rng = np.random.default_rng(20)
    h, w = 1008, 1512
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (127 + 100 * np.sin(xx / 37.0) * np.cos(yy / 53.0)).astype(np.uint8)
    img[..., 1] = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
    img[..., 2] = (rng.integers(0, 30, size=(h, w)) + 100 * ((xx // 90 + yy // 70) % 2)).astype(np.uint8)
    img[..., 3] = 255
    m = api.Matcher()
    got = m.slic(img, api.SlicSettings(20, 5, 5.0, 1, 0))
    ref = ol.slic(img, 20, 5, 5.0, 1, 0)
    assert np.array_equal(got, ref)
    m.close()


# ---- row N2: weak-texture region detection -------------------------------------------------------------
def _weak_scene(w=1216, h=832):
    """a scene with large constant-albedo patches (the 'textureless' variant of the synthetic scene)"""
    return synth.make_scene(w, h, 2, seed=5, textureless=True, flat_cell=3.0)


def test_weak_texture_detection_matches_oracle():
    sc = _weak_scene()
    for flags, close in ((0, True), (api.FLAG_NO_LINE_CLOSING, False)):
        ref = ol.weak_texture(sc.images[0].numpy().astype(np.uint8), connect="true", close_lines=close)
        m = api.matcher_from_scene(sc, flags=flags)
        labels, text, size = m.detect_weak_texture()
        assert np.array_equal(labels, ref["labels"])
        assert np.array_equal(text, ref["text"]) and np.array_equal(size, ref["size"])
        assert (text == -1).sum() >= 1                     # the flat patches are found ...
        weak_px = np.isin(labels, np.nonzero(text == -1)[0])
        flat = ~sc.textured.numpy()
        assert (weak_px & flat).sum() > 0.5 * weak_px.sum()  # ... and they are mostly the constant-albedo areas
        m.close()
    # the reference's own two-pass labelling (with its lossy parent overwrite) gives the same partition here
    ref = ol.weak_texture(sc.images[0].numpy().astype(np.uint8), connect="true")
    lit = ol.weak_texture(sc.images[0].numpy().astype(np.uint8), connect="literal")
    assert len(lit["text"]) >= len(ref["text"])


def test_line_closing_separates_regions_that_leak_through_a_gap():
    """two flat areas joined through a 40-pixel gap in a straight textured stripe: one region without the boundary
    closing, two with it (what the reference's HoughLinesP step is for, main.cpp:385-435); GPU == oracle in both modes"""
    import torch
    w, h = 1600, 1200
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    img[100:1100, 80:780] = 120
    img[100:1100, 820:1520] = 120
    img[580:620, 780:820] = 120
    K = np.array([[900.0, 0, w / 2], [0, 900.0, h / 2], [0, 0, 1]], np.float32)
    Ks, Rs = np.stack([K, K]), np.stack([np.eye(3, dtype=np.float32)] * 2)
    ts = np.array([[0, 0, 0], [-0.2, 0, 0]], np.float32)
    big = {}
    for flags, close in ((0, True), (api.FLAG_NO_LINE_CLOSING, False)):
        ref = ol.weak_texture(img, connect="true", close_lines=close)
        m = api.Matcher()
        m.set_params(api.default_params(depth_min=1.0, depth_max=10.0, flags=flags))
        m.set_views([torch.from_numpy(img.astype(np.float32))] * 2, Ks, Rs, ts)
        labels, text, size = m.detect_weak_texture()
        assert np.array_equal(labels, ref["labels"]) and np.array_equal(text, ref["text"]) and np.array_equal(size, ref["size"])
        big[close] = int((ref["count"][1:] > 40000).sum())
        if close:
            assert ref["segments"] > 0
        m.close()
    assert big[False] == 1 and big[True] == 2


# ---- row N3: fusion ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,n_src", [(160, 120, 4), (2016, 1344, 5)])
def test_fusion_bit_exact(w, h, n_src):
    sc = synth.make_scene(w, h, n_src, seed=8, all_gt=True)
    n = len(sc.images)
    depths = [d.numpy() for d, _ in sc.meta["gt_all"]]
    # perturb: a noisy band and some holes so that the consistency tests actually reject things
    rng = np.random.default_rng(1)
    for v in range(n):
        depths[v] = depths[v] * (1 + rng.normal(0, 0.004, depths[v].shape).astype(np.float32))
        depths[v][rng.uniform(size=depths[v].shape) < 0.05] = 0
    normals = [np.ascontiguousarray((nc.numpy() @ sc.R[v]).astype(np.float32)) for v, (_, nc) in enumerate(sc.meta["gt_all"])]
    grays = [im.numpy() for im in sc.images]
    pairs = {v: [s for s in range(n) if s != v] for v in range(n)}
    for num_consistent, used in ((1, 1), (2, 1), (2, 0)):
        ref = ol.fuse(depths, normals, grays, sc.K, sc.R, sc.t, pairs, num_consistent=num_consistent, used_list=used)
        prm = api.FusionParams(num_consistent, 2.0, 0.01, 15.0, used)
        got = api.fuse(depths, normals, grays, sc.K, sc.R, sc.t, pairs, prm)
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        # on a context (tsar_fuse_ctx: its stream, temporaries from its arena); three calls: the first overflows the empty arena,
        # the second sizes it, the third allocates nothing — same bits every time
        m = api.Matcher()
        for _ in range(3):
            got = api.fuse(depths, normals, grays, sc.K, sc.R, sc.t, pairs, prm, matcher=m)
            assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        m.close()


def test_end_to_end_match_then_fuse():
    """PatchMatch every view on the GPU, fuse on the GPU, check the cloud against the analytic scene"""
    sc = synth.make_scene(192, 128, 4, seed=8)
    n = len(sc.images)
    depths, normals = [], []
    for ref in range(n):
        order = [ref] + [v for v in range(n) if v != ref]
        m = api.Matcher()
        m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max, seed=3 + ref))
        m.set_views([sc.images[v] for v in order], sc.K[order], sc.R[order], sc.t[order])
        m.pm_init(); m.pm_iterate(4); m.compute_disp()
        res = m.get_result(("depth", "normal"))
        depths.append(res["depth"]); normals.append(res["normal"])
        m.close()
    pairs = {v: [s for s in range(n) if s != v] for v in range(n)}
    pts = api.fuse(depths, normals, [im.numpy() for im in sc.images], sc.K, sc.R, sc.t, pairs, api.FusionParams(2, 2.0, 0.01, 15.0, 1))
    assert len(pts) > 0.3 * sc.w * sc.h
    X = pts[:, :3].astype(np.float64)
    n0 = np.array([0.05, 0.02, -1.0]); n0 /= np.linalg.norm(n0)
    n1 = np.array([0.55, 0.10, -1.0]); n1 /= np.linalg.norm(n1)
    dist = np.minimum(np.minimum(np.abs(X @ n0 + 1.6), np.abs(X @ n1 - 0.15)), np.abs(np.linalg.norm(X - np.array([-0.9, 0.35, -0.2]), axis=1) - 0.75))
    assert np.percentile(dist, 90) < 0.05


@pytest.mark.parametrize("strict", [True, False])
def test_set_views_u8_is_set_views_on_the_widened_bytes(small_scene, strict):
    """tsar_set_views_u8 (the 8-bit decode itself, widened on the device — what tsar_gipuma hands over) against tsar_set_views on
    (float)bytes: the same planes, costs and output maps bit for bit, from host bytes and from device-resident bytes; an image
    whose size is not a multiple of 16 takes the kernel's scalar tail"""
    import torch
    sc = small_scene
    flags = api.FLAG_STRICT_DIV if strict else 0
    outs = []
    for form in ("float", "u8 host", "u8 device"):
        m = api.Matcher()
        m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max, seed=9, flags=flags))
        if form == "float":
            m.set_views(sc.images, sc.K, sc.R, sc.t)
        elif form == "u8 host":
            m.set_views([im.numpy().astype(np.uint8) for im in sc.images], sc.K, sc.R, sc.t, u8=True)
        else:
            m.set_views([im.to(torch.uint8).cuda().contiguous() for im in sc.images], sc.K, sc.R, sc.t, u8=True)
        m.pm_init()
        m.pm_iterate(2)
        planes, cost, bv, ratio = m.get_plane()
        m.compute_disp()
        outs.append((planes.view(np.uint32).copy(), cost.copy(), bv.copy(), m.get_result(("depth",))["depth"].copy()))
        m.close()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)
    # odd size (93 x 61 = 5673 pixels, not a multiple of 16) and the refinement operators' reference-only form
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(61, 93)).astype(np.uint8)
    res = []
    for u8 in (False, True):
        m = api.Matcher()
        m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=1.0, depth_max=10.0))
        m.set_views([img if u8 else img.astype(np.float32)], sc.K[:1], sc.R[:1], sc.t[:1], u8=u8)
        labels, text, size = m.detect_weak_texture()
        res.append((labels.copy(), text.copy()))
        m.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
