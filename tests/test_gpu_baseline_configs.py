"""GPU tests shaped like BASELINE.json's configs (the bench line only measures configs[1]; the others are
parity cases):

  cfg1  640x480, 4 source views, 8 iterations      -> whole run bit-exact against the oracle (strict mode)
  cfg2  6048x4032, 10 source views (the bench size) -> size-independent properties of the production (fast) path:
        determinism, never-increasing per-pixel cost, stored cost == re-scored cost of the stored plane,
        planes valid (unit normal facing the camera, depth in range), a checksum of checksums over halves
  cfg5  20 source views, 12 iterations               -> 20-view selection at reduced image size, bit-exact
  odd sizes / strip order: image widths that leave a partial tile, a partial strip, and fewer tiles than a strip
"""
import numpy as np
import pytest
import torch

import oracle_lib as ol
from tsar_mvs_amd import api, synth

pytestmark = pytest.mark.gpu


def _oracle(scene, **kw):
    return ol.Oracle([im.cpu().numpy() for im in scene.images], scene.K, scene.R, scene.t, scene.depth_min, scene.depth_max, **kw)


def _assert_state_equal(m, orc):
    planes, cost, bv, rt = m.get_plane()
    assert np.array_equal(cost, orc.c)
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(bv, orc.beview)
    assert np.array_equal(rt.view(np.uint32), orc.ratio.view(np.uint32))


def test_cfg1_640x480_4views_8iters_bit_exact():
    """BASELINE configs[0] (the reference's CPU-runnable case): the complete run, strict arithmetic"""
    sc = synth.make_scene(640, 480, 4, seed=21)
    orc = _oracle(sc, seed=13)
    orc.pm_init()
    orc.pm_iterate(8)
    m = api.matcher_from_scene(sc, seed=13, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(8)
    _assert_state_equal(m, orc)
    d_ref = orc.compute_disp()
    m.compute_disp()
    res = m.get_result()
    assert np.array_equal(res["depth"], d_ref[..., 3])
    assert np.array_equal(res["normal"], d_ref[..., :3])
    gt = sc.gt_depth.numpy()
    assert (np.abs(res["depth"] - gt) / gt < 0.01).mean() > 0.9
    m.close()


def test_cfg5_20_views_12_iterations_bit_exact():
    """BASELINE configs[4] selects 20 source views and runs 12 iterations; image reduced so the oracle finishes"""
    sc = synth.make_scene(160, 96, 20, seed=23)
    orc = _oracle(sc, seed=17)
    orc.pm_init()
    orc.pm_iterate(12)
    m = api.matcher_from_scene(sc, seed=17, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(12)
    _assert_state_equal(m, orc)
    m.close()


@pytest.mark.parametrize("w,h", [(101, 67), (544, 40), (672, 48), (33, 17)])
def test_odd_sizes_and_partial_strips_bit_exact(w, h):
    """partial tiles on both borders; 17 tiles across = one full 16-tile strip + a 1-tile strip (544), 21 tiles
    (672), fewer tiles than one strip (101, 33)"""
    sc = synth.make_scene(w, h, 3, seed=29)
    orc = _oracle(sc, seed=5)
    orc.pm_init()
    orc.pm_iterate(2)
    m = api.matcher_from_scene(sc, seed=5, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(2)
    _assert_state_equal(m, orc)
    m.close()


def test_cfg2_full_size_properties():
    """BASELINE configs[1] at its full size, through the production (fast-mode) kernels"""
    w, h, n = 6048, 4032, 10
    sc = synth.make_scene(w, h, n, device="cuda", seed=1234)
    m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024)
    dev = torch.device("cuda")
    cost = [torch.empty((h, w), dtype=torch.float32, device=dev) for _ in range(3)]
    depth = torch.empty((h, w), dtype=torch.float32, device=dev)
    normal = torch.empty((h, w, 3), dtype=torch.float32, device=dev)

    def snapshot(k):
        m.compute_disp()
        m.get_result_device(depth=depth, normal=normal, cost=cost[k])

    m.pm_init()
    snapshot(0)
    m.pm_iterate(1)
    snapshot(1)
    m.pm_iterate(1)
    snapshot(2)
    # greedy accepts: a pixel's cost never goes up
    assert bool((cost[1] <= cost[0]).all()) and bool((cost[2] <= cost[1]).all())
    assert float(cost[2].mean()) < float(cost[0].mean()) * 0.5
    # planes are valid: depth 0 only where the cost is MAXCOST, otherwise inside the range; world normals unit length
    valid = cost[2] < 2.0
    assert bool(((depth > 0) == valid).all())
    dv = depth[valid]
    assert float(dv.min()) >= sc.depth_min * (1 - 1e-5) and float(dv.max()) <= sc.depth_max * (1 + 1e-5)
    nn = normal[valid].norm(dim=-1)
    assert float((nn - 1).abs().max()) < 1e-4
    # after two iterations most of the analytic scene is already found
    gt = sc.gt_depth
    assert float(((depth - gt).abs() / gt < 0.01).float().mean()) > 0.9
    # determinism: the same seed reproduces the run bit for bit (checksum of checksums over image halves)
    def checksums():
        d = depth.view(torch.int32).to(torch.int64)
        return [int(d[: h // 2].sum()), int(d[h // 2:].sum()), int(cost[2].view(torch.int32).to(torch.int64).sum())]
    first = checksums()
    m.pm_init()
    m.pm_iterate(2)
    snapshot(2)
    assert checksums() == first
    # idempotence of scoring: the stored cost is the cost of the stored plane (re-scored by the full-cost kernel,
    # which evaluates the same arithmetic with a different instruction selection: fast-mode tolerance)
    planes, c_host, _, _ = m.get_plane()
    c_again, _, _ = m.pm_cost_planes(planes)
    assert np.max(np.abs(c_again - c_host)) <= 1e-3
    m.close()


@pytest.mark.parametrize("block", ["128", "256"])
@pytest.mark.parametrize("w,h", [(192, 128), (101, 67)])
def test_both_workgroup_shapes_bit_exact(monkeypatch, block, w, h):
    """the sweep runs as 256-thread (32 x 16 pixel) workgroups, or 128-thread (32 x 8) ones on small images; small test scenes
    would only ever see the latter, so both shapes are forced here (TSAR_BLOCK is read by tsar_create: set before the matcher exists)"""
    monkeypatch.setenv("TSAR_BLOCK", block)
    sc = synth.make_scene(w, h, 4, seed=31)
    orc = _oracle(sc, seed=23)
    orc.pm_init()
    orc.pm_iterate(2)
    m = api.matcher_from_scene(sc, seed=23, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(2)
    _assert_state_equal(m, orc)
    m.close()
    f = api.matcher_from_scene(sc, seed=23)                 # fast mode: same decisions up to near-ties
    f.set_plane(orc.norm4.copy(), orc.c.copy())
    f.set_sweep_counter(4)
    orc.pm_sweep(0)
    f.pm_sweep(0)
    planes, cost, _, _ = f.get_plane()
    f.close()
    same = np.all(planes.view(np.uint32) == orc.norm4.view(np.uint32), axis=-1)
    assert same.mean() > 0.85
    assert np.max(np.abs(cost - orc.c)[same]) <= 1e-3


@pytest.mark.parametrize("variant", ["114"])
def test_tap_loop_without_d16_loads_bit_exact(monkeypatch, variant):
    """the tap-loop variant tsar_create falls back to when the D16 probe fails (no ds_read_u16_d16_hi): forced here"""
    monkeypatch.setenv("TSAR_VARIANT", variant)
    sc = synth.make_scene(192, 128, 4, seed=11)
    orc = _oracle(sc, seed=19)
    orc.pm_init()
    orc.pm_iterate(2)
    m = api.matcher_from_scene(sc, seed=19, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(2)
    _assert_state_equal(m, orc)
    m.close()
    f = api.matcher_from_scene(sc, seed=19)
    c_fast, _, _ = f.pm_cost_planes(orc.norm4.copy())
    f.close()
    assert np.max(np.abs(c_fast - orc.c)) <= 1e-3




@pytest.mark.parametrize("skew_ref,skew_src", [(0.0, 1.7), (0.9, 0.0), (0.6, -1.2)])
def test_skewed_intrinsics_take_the_full_homography_products(skew_ref, skew_src):
    """plane_homography (tsar_device_math.h) skips the products with the structural zeros of K_src and K_ref^-1 when NO camera of the
    scene has skew (DevScene.k_sparse) — every synthetic scene, and every MVSNet-format cam file.  A skewed K in the reference
    camera, in a source camera, or in both must fall back to the reference's two full 3x3 products: same bits as the oracle in
    strict mode, box-11 loop and general-window loop, and the fast mode's restated arithmetic too."""
    sc = synth.make_scene(128, 96, 3, seed=17)
    sc.K = sc.K.copy()
    sc.K[0, 0, 1] = skew_ref
    sc.K[2, 0, 1] = skew_src
    for box in (11, 7):
        orc = _oracle(sc, seed=29, box=box)
        orc.pm_init()
        orc.pm_iterate(2)
        m = api.matcher_from_scene(sc, seed=29, box=box, flags=api.FLAG_STRICT_DIV)
        m.pm_init()
        m.pm_iterate(2)
        _assert_state_equal(m, orc)
        m.lrdiff()                # rlCost (tsar_refine.hip) builds the same homography
        m.getview()
        m.compute_disp()
        orc.lrdiff_op()
        orc.getview()
        assert np.array_equal(m.get_result()["confid"], orc.confid)
        m.close()


@pytest.mark.parametrize("box,n_best,n_src", [(19, 2, 3), (7, 1, 3), (11, 1, 1), (11, 3, 5), (5, 1, 2), (9, 1, 2), (15, 4, 5), (27, 1, 2), ((13, 7), 2, 3), (11, 4, 5), (11, 5, 5), (19, 6, 6), (7, 5, 5)])
def test_other_windows_and_view_counts_bit_exact(box, n_best, n_src):
    """the general-window kernels (pm_core_lut.h: runtime radius — the reference's default box is 19 = 100 taps —, weights from
    the shared table, chunked lines), best-N lists longer than two, and a single source view (ratio is defined as 0 there,
    DESIGN.md §3)"""
    sc = synth.make_scene(112, 80, n_src, seed=31)
    box, box_v = box if isinstance(box, tuple) else (box, box)
    orc = _oracle(sc, seed=9, box=box, box_v=box_v, n_best=n_best)
    orc.pm_init()
    orc.pm_iterate(2)
    m = api.matcher_from_scene(sc, seed=9, box=box, box_v=box_v, n_best=n_best, flags=api.FLAG_STRICT_DIV)
    m.pm_init()
    m.pm_iterate(2)
    _assert_state_equal(m, orc)
    m.close()


@pytest.mark.parametrize("box", [11, 7, 19])
def test_non_integral_images_bit_exact(box):
    """images that are not an 8-bit decode take the float path (four loads per bilinear tap, float reference window, S hoisted
    weights per thread in LDS), whatever the box"""
    sc = synth.make_scene(112, 80, 3, seed=33)
    rng = np.random.default_rng(0)
    sc.images = [im + torch.from_numpy(rng.uniform(0, 0.5, size=tuple(im.shape)).astype(np.float32)) for im in sc.images]
    orc = _oracle(sc, seed=9, box=box)
    orc.pm_init()
    orc.pm_iterate(2)
    for flags in (api.FLAG_STRICT_DIV,):
        m = api.matcher_from_scene(sc, seed=9, box=box, flags=flags)
        m.pm_init()
        m.pm_iterate(2)
        _assert_state_equal(m, orc)
        m.close()
    # fast mode on the float path: same plane scored within the documented tolerance
    f = api.matcher_from_scene(sc, seed=9, box=box)
    c_fast, _, _ = f.pm_cost_planes(orc.norm4)
    c_ref, _, _ = orc.pm_cost_planes(orc.norm4)
    assert np.max(np.abs(c_fast - c_ref)) <= 2e-3
    f.close()


def test_contexts_release_their_device_memory():
    """create / run / destroy in a loop: device memory returns to the baseline (every buffer is owned by the context)"""
    sc = synth.make_scene(640, 480, 3, seed=3)
    torch.cuda.synchronize()

    def cycle():
        m = api.matcher_from_scene(sc, seed=1)
        m.pm_init()
        m.pm_iterate(1)
        m.lrdiff()
        m.getview()
        m.compute_disp()
        m.get_result()
        lab = np.zeros((sc.h, sc.w), np.int32)
        m.set_regions(lab, np.array([1.0], np.float32))
        m.set_reliable_mask(np.ones((sc.h, sc.w), np.float32))
        m.wmf(1, False)
        m.close()

    cycle()                                   # first use pays one-time runtime allocations
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        cycle()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"leaked {(free0 - free1) >> 20} MiB over 12 contexts"


def test_two_contexts_on_one_device_from_two_threads():
    """the CLI's --workers: two host threads, one context each, same GPU, interleaved kernels -> same bits as alone"""
    import threading
    sc = synth.make_scene(320, 240, 3, seed=4)
    ref = api.matcher_from_scene(sc, seed=2)
    ref.pm_init(); ref.pm_iterate(2)
    want = ref.get_plane()
    ref.close()
    got = [None, None]

    def work(k):
        m = api.matcher_from_scene(sc, seed=2)
        m.pm_init(); m.pm_iterate(2)
        got[k] = m.get_plane()
        m.close()
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got[k], want))


def test_random_shapes_bit_exact():
    """a fixed pseudo-random list of image sizes, view counts, windows and best-N settings: init + one full iteration"""
    rng = np.random.default_rng(2024)
    for case in range(10):
        w, h = int(rng.integers(40, 300)), int(rng.integers(30, 200))
        n_src = int(rng.integers(1, 7))
        box = int(rng.choice([5, 7, 9, 11, 13]))
        n_best = int(rng.integers(1, 4))
        comb = int(rng.integers(0, 2))
        flags = int(rng.choice([0, api.FLAG_FIX_DOWN_FAR_SEED | api.FLAG_FIX_RIGHT_FAR_CMP]))
        sc = synth.make_scene(w, h, n_src, seed=100 + case)
        orc = _oracle(sc, seed=case, box=box, n_best=n_best, cost_comb=comb, flags=flags)
        orc.pm_init()
        orc.pm_iterate(1)
        m = api.matcher_from_scene(sc, seed=case, box=box, n_best=n_best, cost_comb=comb, flags=flags | api.FLAG_STRICT_DIV)
        m.pm_init()
        m.pm_iterate(1)
        planes, cost, bv, rt = m.get_plane()
        ctx = (case, w, h, n_src, box, n_best, comb, flags)
        assert np.array_equal(cost, orc.c), ctx
        assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32)), ctx
        assert np.array_equal(bv, orc.beview), ctx
        m.close()


def test_degenerate_inputs_bit_exact():
    """edge cases of the boundary: the smallest accepted image, a textureless reference (every hypothesis scores MAXCOST, the
    exported depth is 0 everywhere), 32 scored views out of 33 supplied (the width of the reference's cost vector)"""
    # 8 x 8, one source view
    sc = synth.make_scene(8, 8, 1, seed=3)
    orc = _oracle(sc, seed=1)
    orc.pm_init(); orc.pm_iterate(2)
    m = api.matcher_from_scene(sc, seed=1, flags=api.FLAG_STRICT_DIV)
    m.pm_init(); m.pm_iterate(2)
    _assert_state_equal(m, orc)
    m.close()
    # flat reference image
    sc = synth.make_scene(64, 48, 2, seed=4)
    imgs = [im.clone() for im in sc.images]
    imgs[0][:] = 77.0
    orc = ol.Oracle([im.numpy() for im in imgs], sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, seed=2)
    orc.pm_init(); orc.pm_iterate(1)
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max, seed=2, flags=api.FLAG_STRICT_DIV))
    m.set_views(imgs, sc.K, sc.R, sc.t)
    m.pm_init(); m.pm_iterate(1)
    _assert_state_equal(m, orc)
    m.compute_disp()
    res = m.get_result(("depth", "cost"))
    assert (res["cost"] == 2.0).all() and (res["depth"] == 0.0).all()
    m.close()
    # 33 source views: the default subset is the first 32; cost_comb ALL averages all 32 (the NB = 32 kernels)
    sc = synth.make_scene(48, 32, 33, seed=5)
    orc = _oracle(sc, seed=3, cost_comb=api.COMB_ALL, subset=list(range(1, 33)))
    orc.pm_init(); orc.pm_iterate(1)
    m = api.matcher_from_scene(sc, seed=3, cost_comb=api.COMB_ALL, flags=api.FLAG_STRICT_DIV)
    m.pm_init(); m.pm_iterate(1)
    _assert_state_equal(m, orc)
    with pytest.raises(api.TsarError):
        m.set_view_subset(list(range(1, 34)))
    m.close()


@pytest.mark.parametrize("w,h,n_best,box", [(101, 67, 1, 11), (640, 480, 1, 11), (2016, 1344, 3, 11), (333, 222, 2, 19), (640, 480, 5, 7), (320, 200, 1, 9)])
def test_structured_buffer_gathers_change_no_bit(monkeypatch, w, h, n_best, box):
    """from the third sweep of a run on the fast tap loops issue their gathers as structured buffer loads (pm_tap_r5.h BUF: the
    addresser scales the element index) from the half-float difference texture (MIX: v_fma_mix_f32 blend, no byte converts);
    TSAR_MIX_GATHER=0 keeps the byte texture for them, TSAR_BUFFER_GATHER=0 keeps global loads and a shift in every launch.  Same
    arithmetic: whole fast-mode runs must agree bit for bit in all three forms, for both workgroup shapes and both best-N
    selections of the box-11 loop and for the general-window loop (chunks of 4 / 5 / 6 taps)"""
    sc = synth.make_scene(w, h, 4, device="cuda" if w > 1000 else "cpu", seed=19)
    outs = []
    for buf, mix in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("TSAR_BUFFER_GATHER", buf)
        monkeypatch.setenv("TSAR_MIX_GATHER", mix)
        m = api.matcher_from_scene(sc, seed=5, n_best=n_best, box=box)
        m.pm_init()
        m.pm_iterate(3)
        planes, cost, bv, ratio = m.get_plane()
        outs.append((planes.view(np.uint32).copy(), cost.view(np.uint32).copy(), bv.copy(), ratio.view(np.uint32).copy()))
        m.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)


def test_live_path_full_size_bit_exact():
    """BASELINE configs[3] at its full size: the reference's live path (external planes -> weak-texture regions -> region RANSAC ->
    plane fill) on one 6048x4032 view, operator by operator against the oracle (tools/full_size_refine_check.py: ~10 s, most of
    it the oracle's RANSAC).  The PatchMatch counterpart at this size (tools/full_size_oracle_check.py) needs ~2 minutes of 16 host
    cores per iteration and is kept as a recorded run (profiles/r02/full_size_oracle_check.json), not a test."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "full_size_refine_check.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rep = json.loads(out.stdout)
    assert rep["all_bit_identical"] is True and len(rep["steps"]) == 5
    assert rep["median_relative_depth_error_there"]["after"] < 1e-3 < rep["median_relative_depth_error_there"]["before"]


def test_bench_workload_full_size_first_steps_bit_exact():
    """BASELINE configs[1] at its full size (6048x4032, 1 + 10 views), strict mode against the oracle: the random initialisation
    (one scored hypothesis per pixel and view: the whole cost function on 24.4 M pixels) and the exported maps, bit for bit
    (tools/full_size_oracle_check.py --iters 0: ~11 s of oracle time).  With iterations the same script needs ~2 minutes of 16
    host cores each and is kept as a recorded run (profiles/r02/full_size_oracle_check.json: five iterations, all bit-identical)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "full_size_oracle_check.py"), "--iters", "0"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rep = json.loads(out.stdout)
    assert rep["all_bit_identical"] is True and rep["steps"][0]["after"] == "pm_init" and rep["steps"][0]["pixels"] == 6048 * 4032


def _every_half_sweep_on_windows(mode, W, H, n_src, iters, min_changed, box=11, n_best=1, converged=0.99, keep=None):
    """every half-sweep of a full-size run against the oracle on twelve 192 x 160 windows (see the tests below); keep: a subset of
    the twelve (row-major on the 4 x 3 grid), for the configurations whose oracle time per window is several times the bench workload's"""
    import torch
    sc = synth.make_scene(W, H, n_src, device="cuda", seed=1234)
    images = [im.cpu().numpy() for im in sc.images]
    orc = ol.Oracle(images, sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, seed=2024, box=box, n_best=n_best, flags=ol.FLAGS_FAST_8BIT_IMAGERY if mode == "fast" else 0)
    m = api.matcher_from_scene(sc, box=box, n_best=n_best, seed=2024, flags=0 if mode == "fast" else api.FLAG_STRICT_DIV)
    if mode == "fast":
        orc.set_rcp_table(ol.rcp_table_from_device(m))
    rw, rh = 192, 160
    xs, ys = [0, W // 3 - 7, 2 * W // 3 + 5, W - rw], [0, H // 2 - 3, H - rh]
    rects = [(x, y, x + rw, y + rh) for y in ys for x in xs]
    if keep is not None:
        rects = [rects[k] for k in keep]
    mask = np.zeros((H, W), bool)
    for x0, y0, x1, y1 in rects:
        mask[y0:y1, x0:x1] = True
    yy, xx = np.nonzero(mask)
    m.pm_init()
    planes, cost, _, _ = m.get_plane()
    changed = 0
    for sweep in range(2 * iters):
        colour = sweep & 1
        orc.norm4[...] = planes
        orc.c[...] = cost
        orc.set_launch(sweep)
        orc.pm_sweep_rects(colour, rects)
        m.pm_sweep(colour)
        planes_after, cost_after, _, _ = m.get_plane()
        sel = ((xx + yy) & 1) == colour
        py, px = yy[sel], xx[sel]
        assert np.array_equal(cost_after[py, px], orc.c[py, px]), "costs differ after half-sweep %d" % sweep
        assert np.array_equal(planes_after[py, px].view(np.uint32), orc.norm4[py, px].view(np.uint32)), "planes differ after half-sweep %d" % sweep
        changed += int((cost_after[py, px] != cost[py, px]).sum())
        planes, cost = planes_after, cost_after
    assert changed > min_changed                  # the sweeps did work inside the windows
    gt = sc.gt_depth.cpu().numpy()
    m.compute_disp()
    depth = m.get_result(("depth",))["depth"]
    assert (np.abs(depth - gt) / gt < 0.01).mean() > converged      # and the run converged (bench.py reports the same figure)
    assert not orc.rcp_out_of_range
    m.close()


@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_bench_workload_full_size_every_half_sweep_on_windows(mode):
    """BASELINE configs[1] at its full size (6048x4032, 1 + 10 views, 8 iterations), in the arithmetic bench.py times ("fast":
    against the oracle's restatement of that arithmetic, oracle/tsar_oracle.c S7, with the device's v_rcp_f32 table) and in the
    reference's arithmetic ("strict"): EVERY one of the 16 half-sweeps
    of the run the bench times is checked against the oracle, bit for bit, on twelve 192 x 160 windows of the image (corners,
    borders, interior: 0.37 Mpixel).  Before each launch the oracle takes the GPU's state (so each launch is judged on its own
    inputs: propagation reads up to 23 pixels beyond a window), runs the same launch restricted to the windows
    (orc_pm_sweep_rects: ~0.6 s of 16 host cores instead of a minute for the whole image) and the planes and costs of the swept
    colour inside the windows must match.  The whole-image version of this comparison is tools/full_size_oracle_check.py
    (profiles/r02/full_size_oracle_check.json: five iterations, ~2 minutes of oracle time each)."""
    _every_half_sweep_on_windows(mode, 6048, 4032, 10, 8, 100000)


@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_cfg5_full_size_every_half_sweep_on_windows(mode):
    """BASELINE configs[4] at its full size — 3840 x 2160, 1 + 20 views, 12 iterations: all 24 half-sweeps, both arithmetic modes,
    bit for bit against the oracle on six of the twelve windows — the four corners and the two interior ones (round 4 had this
    configuration at full size only as a bench record)"""
    _every_half_sweep_on_windows(mode, 3840, 2160, 20, 12, 50000, keep=(0, 3, 5, 6, 8, 11))


@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_reference_default_window_full_size_half_sweeps_on_windows(mode):
    """the reference BINARY's own defaults — box 19 (100 taps), n_best 2 (algorithmparameters.h:25-26), the general-window tap loop of
    pm_core_lut.h — at the bench workload's full size: the six half-sweeps of three iterations, both modes, bit for bit on six of the windows (corners, interior)"""
    _every_half_sweep_on_windows(mode, 6048, 4032, 10, 3, 50000, box=19, n_best=2, converged=0.9, keep=(0, 3, 5, 6, 8, 11))
