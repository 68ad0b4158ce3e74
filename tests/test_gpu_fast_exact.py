"""The DEFAULT ("fast") arithmetic of the HIP library — the mode bench.py times — against the oracle's restatement of that same
arithmetic (oracle/tsar_oracle.c S7), BIT FOR BIT.

tests/test_gpu_fast_mode.py bounds how far the fast mode is from the reference's arithmetic (statistically: fast differs from
strict by rounding, and PatchMatch amplifies rounding into different random walks).  This file closes the other half: that the
fast kernels compute exactly what their specification says — the reference's algorithm with five stated rounding liberties
(reciprocal-multiply perspective divide on the GPU's v_rcp_f32, homography as A - b m^T, (w s) r, clamp to [0, w - 1], row-wise
summation on 8-bit imagery) — and nothing else: no addressing slip, no skipped hypothesis, no mis-ordered accept.  The one
hardware function involved, v_rcp_f32, enters the oracle as a table of its 2^23 mantissa results read from the device through
the C ABI (tsar_selftest_divide, fast form); the test first checks that the table plus exact exponent arithmetic reproduces the
device's reciprocal on operands across the whole normal range."""
import numpy as np
import pytest

import oracle_lib as ol
from tsar_mvs_amd import api, synth

pytestmark = pytest.mark.gpu

FAST8 = ol.FLAGS_FAST_8BIT_IMAGERY           # reciprocal divide, A - b m^T, (w s) r, rows: the production kernels on 8-bit imagery
FASTF = ol.FLAG_FAST_ARITH                   # the same with column order: float imagery, and the column-order fast loop


@pytest.fixture(scope="module")
def rcp_table():
    m = api.Matcher()
    t = ol.rcp_table_from_device(m)
    m.close()
    return t


def _oracle(scene, table, images=None, **kw):
    o = ol.Oracle([im.cpu().numpy() for im in (images or scene.images)], scene.K, scene.R, scene.t, scene.depth_min, scene.depth_max, **kw)
    o.set_rcp_table(table)
    return o


def _random_planes(scene, orc, seed):
    rng = np.random.default_rng(seed)
    h, w = scene.h, scene.w
    planes = np.empty((h, w, 4), np.float32)
    for y in range(h):
        for x in range(w):
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            if n @ orc.view_vector(x, y) > 0:
                n = -n
            n = n.astype(np.float32)
            planes[y, x, :3] = n
            planes[y, x, 3] = orc.getD(n, x, y, rng.uniform(scene.depth_min, scene.depth_max))
    return planes


def _same_state(m, orc):
    planes, cost, bv, rt = m.get_plane()
    assert np.array_equal(cost, orc.c)
    assert np.array_equal(planes.view(np.uint32), orc.norm4.view(np.uint32))
    assert np.array_equal(bv, orc.beview)
    assert np.array_equal(rt.view(np.uint32), orc.ratio.view(np.uint32))


def test_rcp_table_reproduces_the_device_reciprocal(rcp_table):
    assert rcp_table[0] == 1.0 and 0.5 < rcp_table.min() and rcp_table.max() <= 1.0
    # 1 ulp of 1 / x (what the ISA promises), and not simply the correctly rounded reciprocal (else the table would be pointless)
    z = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
    exact = (1.0 / z.astype(np.float64))
    ulp = np.spacing(exact.astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(rcp_table.astype(np.float64) - exact) / ulp) <= 1.0
    assert (rcp_table != (np.float32(1.0) / z)).mean() > 0.01
    # exponent / sign handling of the oracle's rcp_gpu against the device, on operands across the normal range
    rng = np.random.default_rng(5)
    n = 1 << 20
    x = (rng.integers(0, 1 << 23, n, dtype=np.uint32) | (rng.integers(2, 253, n, dtype=np.uint32) << 23) | (rng.integers(0, 2, n, dtype=np.uint32) << 31)).view(np.float32)
    m = api.Matcher()
    dev, _ = m.selftest_divide(np.ones_like(x), np.ones_like(x), x, mode=2)
    m.close()
    bits = x.view(np.uint32)
    t = rcp_table[bits & 0x7fffff].view(np.uint32)
    re = ((t >> 23) & 0xff).astype(np.int64) + 127 - ((bits >> 23) & 0xff).astype(np.int64)
    ok = (re > 0) & (re < 255)
    emu = ((bits & 0x80000000) | (re.clip(0, 255).astype(np.uint32) << 23) | (t & 0x7fffff)).view(np.float32)
    assert ok.mean() > 0.98
    assert np.array_equal(emu[ok].view(np.uint32), dev[ok].view(np.uint32))


# box 11: the hand-scheduled loop (pm_tap_r5.h, row walk); every other box and n_best > 4: the general-window loop (pm_core_lut.h)
@pytest.mark.parametrize("box,n_best,comb", [(11, 1, 1), (11, 2, 1), (11, 4, 1), (11, 5, 1), (11, 1, 0), (7, 1, 1), (19, 2, 1), (9, 3, 0),
                                             ((7, 13), 1, 1), ((19, 5), 2, 1), (25, 1, 1), (1, 1, 1), (12, 1, 1)])
def test_cost_planes_fast_bit_exact(small_scene, rcp_table, box, n_best, comb):
    sc = small_scene
    box, box_v = box if isinstance(box, tuple) else (box, box)
    orc = _oracle(sc, rcp_table, box=box, box_v=box_v, n_best=n_best, cost_comb=comb, flags=FAST8)
    m = api.matcher_from_scene(sc, box=box, box_v=box_v, n_best=n_best, cost_comb=comb)
    for planes in (synth.gt_planes(sc).numpy(), _random_planes(sc, orc, 3)):
        c_ref, bv_ref, rt_ref = orc.pm_cost_planes(planes)
        c, bv, rt = m.pm_cost_planes(planes)
        assert np.array_equal(c, c_ref)
        assert np.array_equal(bv, bv_ref)
        assert np.array_equal(rt.view(np.uint32), rt_ref.view(np.uint32))
    assert not orc.rcp_out_of_range
    m.close()


@pytest.mark.parametrize("box,n_best,flags", [(11, 1, 0), (11, 1, api.FLAG_FIX_DOWN_FAR_SEED | api.FLAG_FIX_RIGHT_FAR_CMP), (19, 2, 0), (7, 3, 0), (12, 1, 0),
                                              (11, 1, api.FLAG_TEX_FILTER_8BIT), (15, 2, api.FLAG_TEX_FILTER_8BIT)])
def test_init_and_iterations_fast_bit_exact(small_scene, rcp_table, box, n_best, flags):
    """random initialisation + three red/black iterations (the sweep kernel: global-load gathers for the first two launches,
    buffer-load gathers from the third on) + lrdiff, in the default arithmetic"""
    sc = small_scene
    orc = _oracle(sc, rcp_table, seed=5, box=box, n_best=n_best, flags=flags | FAST8)
    m = api.matcher_from_scene(sc, seed=5, box=box, n_best=n_best, flags=flags)
    orc.pm_init()
    m.pm_init()
    _same_state(m, orc)
    orc.pm_iterate(3)
    m.pm_iterate(3)
    _same_state(m, orc)
    orc.lrdiff_op()
    orc.getview()
    m.lrdiff()
    m.getview()
    m.compute_disp()
    assert np.array_equal(m.get_result(("confid",))["confid"], orc.confid)
    assert not orc.rcp_out_of_range
    m.close()


def test_cfg1_whole_run_fast_bit_exact(rcp_table):
    """BASELINE configs[0]: 640x480, 4 source views, 8 iterations, in the arithmetic bench.py times"""
    sc = synth.make_scene(640, 480, 4, seed=1234)
    orc = _oracle(sc, rcp_table, seed=2024, flags=FAST8)
    m = api.matcher_from_scene(sc, seed=2024)
    orc.pm_init()
    orc.pm_iterate(8)
    m.pm_init()
    m.pm_iterate(8)
    _same_state(m, orc)
    m.compute_disp()
    ref = orc.compute_disp()
    res = m.get_result(("depth", "normal"))
    assert np.array_equal(res["depth"], ref[..., 3]) and np.array_equal(res["normal"], ref[..., :3])
    gt = sc.gt_depth.numpy()
    assert (np.abs(res["depth"] - gt) / gt < 0.01).mean() > 0.9
    assert not orc.rcp_out_of_range
    m.close()


def test_float_imagery_and_column_order_fast_bit_exact(small_scene, rcp_table, monkeypatch):
    """the fast arithmetic in COLUMN order: images that are not an 8-bit decode (the generic one-tap loop), and the column-order
    variant of the hand-scheduled loop (TSAR_VARIANT=122: what a device whose D16 probe fails would run, minus the D16 loads)"""
    sc = small_scene
    imgs = [im + 0.25 for im in sc.images]
    orc = _oracle(sc, rcp_table, images=imgs, seed=9, flags=FASTF)
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max, seed=9))
    m.set_views(imgs, sc.K, sc.R, sc.t)
    orc.pm_init(); orc.pm_iterate(2)
    m.pm_init(); m.pm_iterate(2)
    _same_state(m, orc)
    m.close()
    monkeypatch.setenv("TSAR_VARIANT", "122")
    orc = _oracle(sc, rcp_table, seed=9, flags=FASTF)
    m = api.matcher_from_scene(sc, seed=9)
    orc.pm_init(); orc.pm_iterate(2)
    m.pm_init(); m.pm_iterate(2)
    _same_state(m, orc)
    m.close()


def test_twenty_views_twelve_iterations_fast_bit_exact(rcp_table):
    """BASELINE configs[4]'s view count and iteration count at reduced image size, default arithmetic"""
    sc = synth.make_scene(160, 96, 20, seed=77)
    orc = _oracle(sc, rcp_table, seed=3, flags=FAST8)
    m = api.matcher_from_scene(sc, seed=3)
    orc.pm_init(); orc.pm_iterate(12)
    m.pm_init(); m.pm_iterate(12)
    _same_state(m, orc)
    m.close()
