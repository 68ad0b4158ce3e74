"""Taps on and beyond the border of a source image, and behind the camera — the addressing edge behind round 2's abort.

What happened (gpurun_out/t_r2c.log, round 2): while the quad-texture base was being moved into an SGPR pair with an UNSIGNED
per-tap offset from entry (1, 1), strict mode still clamped tap positions to the oracle's [-1, w].  A tap with u in [-1, 0) then
has floor(u) = -1, its element index iv * pitch + iu became negative, the unsigned byte offset wrapped to ~4 GiB and the gather
left the allocation: a GPU memory fault, reported as `Fatal Python error: Aborted` inside tsar_pm_cost_planes
(tests/test_golden.py, whose ground-truth planes put border windows half a pixel outside a source view).  Fixed in the next commit:
positions are clamped to [0, w - 1] (the same sample bit for bit: with edge replication both texels of the pair are the edge
texel), and the clamp-free loop keeps a pixel of margin (pm_tap_r5.h).  The production fast path has since moved to range-checked
buffer loads, which would turn the same slip into silent zeros instead of a fault — so this test pins the edge in every form:
strict and fast arithmetic, global-load and buffer-load gathers (the two must agree BIT FOR BIT), the every-pixel kernel and the
sweep kernel, against the CPU oracle run in the same arithmetic (bit for bit in both).

Construction: reference camera at the origin, three source cameras that differ by a pure translation along x, y and z; for a
fronto-parallel plane at depth Z0 the homography is then the shift u = x + f tx / Z0 (v likewise), or a point reflection with
Z = 1 + tz / Z0 < 0.  Z0 is chosen so that the outermost taps of the border pixels' windows land at the wanted offsets."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from tsar_mvs_amd import api

pytestmark = pytest.mark.gpu

W, H, F = 96, 64, 100.0
TX, TY = 0.1, 0.1
# tap offset (pixels) of the window's outermost column / row relative to the border texel: inside [-1, 0), exactly -1, beyond -1
# (the oracle clamps to [-1, w]), a hair below 0, and the same on the far side through the sign
SHIFTS = [-0.5, -0.999, -1.0, -1.5, -3.25, -1e-4, 0.5, 0.999, 1.0, 1.5, 3.25, 1e-4]


def _scene():
    rng = np.random.default_rng(31)
    base = rng.integers(0, 256, size=(H + 16, W + 16)).astype(np.float32)
    imgs = [base[8:8 + H, 8:8 + W].copy(), base[8:8 + H, 9:9 + W].copy(), base[9:9 + H, 8:8 + W].copy(), base[7:7 + H, 8:8 + W].copy()]
    K = np.tile(np.array([[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]], np.float32), (4, 1, 1))
    R = np.tile(np.eye(3, dtype=np.float32), (4, 1, 1))
    t = np.array([[0, 0, 0], [TX, 0, 0], [0, TY, 0], [0.01, 0.02, -1.0]], np.float32)
    return imgs, K, R, t


def _plane_map(z0):
    p = np.zeros((H, W, 4), np.float32)
    p[..., 2] = -1.0
    p[..., 3] = z0                      # n . X + d = 0 with n = (0, 0, -1): the plane Z = z0
    return p


def _cases():
    out = []
    for s in SHIFTS:
        out.append((1, F * TX / s))     # view 1: u = x + s; a negative depth is a legal plane for pm_cost_planes (Z = 1 here)
        out.append((2, F * TY / s))     # view 2: v = y + s
    out += [(3, 0.5), (3, 0.9), (3, 0.999)]     # view 3: Z = 1 - 1 / z0 < 0 (behind the source camera), and barely negative
    return out


def _matcher(flags, buffer_gather, mix_gather=True):
    """buffer_gather: the sweeps' gathers as structured buffer loads (from the third sweep on) or global loads; mix_gather: the
    buffer-load launches read the half-float difference texture (pm_tap_r5.h MIX) or the byte texture.  Read by tsar_create."""
    old = {k: os.environ.get(k) for k in ("TSAR_BUFFER_GATHER", "TSAR_MIX_GATHER")}
    os.environ["TSAR_BUFFER_GATHER"] = "1" if buffer_gather else "0"
    os.environ["TSAR_MIX_GATHER"] = "1" if mix_gather else "0"
    try:
        m = api.Matcher()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v
    imgs, K, R, t = _scene()
    m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=1e-3, depth_max=1e6, seed=3, flags=flags))
    m.set_views(imgs, K, R, t)
    return m


def _sweep_cost(m, planes):
    """cost of `planes` as the SWEEP kernel scores them: every pixel holds MAXCOST and the same plane as all its neighbours, so a
    propagation-only sweep of both colours leaves c[p] = cost(plane at p).  Sweep counter 2: the launcher's buffer-load form
    (when the context allows it)."""
    m.set_plane(planes, np.full((H, W), 2.0, np.float32))
    m.set_sweep_counter(2)
    m.pm_sweep(0, do_prop=True, do_refine=False)
    m.pm_sweep(1, do_prop=True, do_refine=False)
    return m.get_plane()[1]


@pytest.fixture
def block_shape(request):
    """TSAR_BLOCK (read once per context, by tsar_create: the fixture sets it before the matchers are made): the sweep's 128-thread workgroups (what an image this small gets) or the 256-thread shape of
    full-size images"""
    old = os.environ.get("TSAR_BLOCK")
    os.environ["TSAR_BLOCK"] = str(request.param)
    yield request.param
    if old is None:
        os.environ.pop("TSAR_BLOCK")
    else:
        os.environ["TSAR_BLOCK"] = old


@pytest.mark.parametrize("block_shape", [128, 256], indirect=True)
@pytest.mark.parametrize("strict", [True, False])
def test_border_and_behind_camera_taps_all_gather_forms(strict, block_shape):
    imgs, K, R, t = _scene()
    flags = api.FLAG_STRICT_DIV if strict else 0
    mb, mg, mq = _matcher(flags, True), _matcher(flags, False), _matcher(flags, True, mix_gather=False)
    n_border_pixels = 0
    for view, z0 in _cases():
        if z0 < 0 and view != 3:
            continue                    # (negative shifts are produced by the sign of the shift below, not of the depth)
        planes = _plane_map(np.float32(z0))
        orc = ol.Oracle(imgs, K, R, t, 1e-3, 1e6, box=11, n_best=1, seed=3, subset=[view], flags=0 if strict else ol.FLAGS_FAST_8BIT_IMAGERY)
        if not strict:
            orc.set_rcp_table(ol.rcp_table_from_device(mb))     # fast mode: against the oracle's restatement of the fast arithmetic
        c_ref = orc.pm_cost_planes(planes)[0]
        res = {}
        for name, m in (("buffer", mb), ("global", mg), ("bytes", mq)):
            m.set_view_subset([view])
            res[name + "_full"] = m.pm_cost_planes(planes)[0]
            res[name + "_sweep"] = _sweep_cost(m, planes)
        # the three gather forms (buffer loads from the half-float difference texture, buffer loads from the byte texture, global
        # loads), and the two kernels, agree bit for bit in either arithmetic
        for k in ("buffer_sweep", "global_full", "global_sweep", "bytes_sweep"):
            assert np.array_equal(res["buffer_full"], res[k]), (view, z0, k)
        assert np.array_equal(res["buffer_sweep"], c_ref), (view, z0)          # either mode: the oracle run in the same arithmetic
        n_border_pixels += int((c_ref < 2.0).sum())
    assert n_border_pixels > 1000        # the cases score real windows, not MAXCOST everywhere
    mb.close(); mg.close(); mq.close()


@pytest.mark.parametrize("strict", [True, False])
def test_negative_shifts(strict):
    """the same with the cameras mirrored (tx, ty < 0): taps beyond the LEFT and TOP borders (u, v in [-1, 0) and below) — the
    side on which the unsigned offset wrapped"""
    imgs, K, R, t = _scene()
    t = t.copy()
    t[1, 0], t[2, 1] = -TX, -TY
    flags = api.FLAG_STRICT_DIV if strict else 0
    ms = []
    for bg in (True, False):
        m = _matcher(flags, bg)
        m.set_views(imgs, K, R, t)
        ms.append(m)
    for s in SHIFTS:
        if s < 0:
            continue
        for view, z0 in ((1, F * TX / s), (2, F * TY / s)):      # u = x - s, v = y - s
            planes = _plane_map(np.float32(z0))
            orc = ol.Oracle(imgs, K, R, t, 1e-3, 1e6, box=11, n_best=1, seed=3, subset=[view], flags=0 if strict else ol.FLAGS_FAST_8BIT_IMAGERY)
            if not strict:
                orc.set_rcp_table(ol.rcp_table_from_device(ms[0]))
            c_ref = orc.pm_cost_planes(planes)[0]
            got = []
            for m in ms:
                m.set_view_subset([view])
                got += [m.pm_cost_planes(planes)[0], _sweep_cost(m, planes)]
            for g in got[1:]:
                assert np.array_equal(got[0], g), (view, s)
            assert np.array_equal(got[0], c_ref), (view, s)
    for m in ms:
        m.close()


@pytest.mark.parametrize("strict", [True, False])
def test_perspective_divide_at_the_operand_guard(strict):
    """strict mode's short division form is only valid for operands in [2^-20, 2^38] (tsar_device_math.h); the clamp-free tap loop
    proves that from the window's corners (Z >= 2^-18 there), the clamp loop tests it per tap.  Planes whose Z = 1 + tz / z0 sits
    just below, on and above both thresholds (view 3, tz = -1: Z ~ z0 - 1 for z0 = 1 + k 2^-23), and Z of 2^17 .. 2^39 (a plane
    2^-17 .. 2^-39 behind the reference camera): the every-pixel kernel against the oracle, bit for bit in strict mode (the
    oracle divides with IEEE `/`); in fast mode against the oracle's restatement of the fast arithmetic, bit for bit as well (the
    fast homography A - b m^T cancels badly for such planes, so fast and strict costs differ visibly here: extreme planes are
    where the two arithmetics part, and each must still be exactly what it says)."""
    imgs, K, R, t = _scene()
    flags = api.FLAG_STRICT_DIV if strict else 0
    m = _matcher(flags, True)
    m.set_view_subset([3])
    orc = ol.Oracle(imgs, K, R, t, 1e-3, 1e6, box=11, n_best=1, seed=3, subset=[3], flags=0 if strict else ol.FLAGS_FAST_8BIT_IMAGERY)
    if not strict:
        orc.set_rcp_table(ol.rcp_table_from_device(m))
    z0s = [np.float32(1.0) + np.float32(k * 2.0 ** -23) for k in (1, 2, 7, 8, 9, 31, 32, 33, 34, 100, 1000)]
    z0s += [np.float32(-(2.0 ** -e)) for e in (16, 17, 18, 19, 37, 38, 39, 40)]
    scored = 0
    for z0 in z0s:
        planes = _plane_map(z0)
        c_ref = orc.pm_cost_planes(planes)[0]
        c = m.pm_cost_planes(planes)[0]
        assert np.array_equal(c, c_ref), float(z0)
        scored += int((c_ref < 2.0).sum())
    assert scored > 0 and not orc.rcp_out_of_range
    m.close()


# ---- the fast path's clamp-free decision from the window's centre and a bound on its extent (pm_tap_r5.h:52-71) -------------------
def _rot(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return (Rz @ Rx @ Ry).astype(np.float32)


def _scene_rotated():
    """the translations of _scene() plus small rotations, so that the homographies have perspective terms (H6, H7 != 0: the
    bound's c = 5 (|H6| + |H7|) and |Xc| c terms are live) and the window's extent in the source image is not 5 pixels flat"""
    imgs, K, R, t = _scene()
    R = R.copy()
    R[1], R[2], R[3] = _rot(0.04, 0.0, 0.0), _rot(0.0, -0.05, 0.01), _rot(-0.03, 0.02, -0.02)
    t = t.copy()
    t[3] = (0.07, -0.06, 0.02)
    return imgs, K, R, t


def _centre_extent_margins(Hm):
    """the decision of pm_tap_r5.h:52-71 restated in numpy (fp32): per pixel, the four distances by which the window bound clears
    (>= 0) or misses (< 0) the source image's clamp-free region; H is the 3x3 plane homography of one view"""
    f32 = np.float32
    Hm = np.asarray(Hm, f32).ravel()
    ys, xs = np.mgrid[0:H, 0:W]
    xc, yc = xs.astype(f32), ys.astype(f32)
    Xc, Yc, Zc = Hm[0] * xc + Hm[1] * yc + Hm[2], Hm[3] * xc + Hm[4] * yc + Hm[5], Hm[6] * xc + Hm[7] * yc + Hm[8]
    a, b, c = abs(Hm[0]) + abs(Hm[1]), abs(Hm[3]) + abs(Hm[4]), abs(Hm[6]) + abs(Hm[7])
    Zmin = Zc - f32(5) * c
    with np.errstate(all="ignore"):
        r = f32(1) / (Zmin * Zc)
        du, dv = (a * Zc + np.abs(Xc) * c) * f32(5) * r, (b * Zc + np.abs(Yc) * c) * f32(5) * r
        uc, vc = Xc * Zmin * r, Yc * Zmin * r
    ok = Zmin > 0
    return [np.where(ok, m, -np.inf) for m in (uc - du - f32(1.5), vc - dv - f32(1.5), f32(W - 1) - f32(1.5) - (uc + du), f32(H - 1) - f32(1.5) - (vc + dv))]


def test_fast_path_window_bound_walked_across_its_threshold_on_every_border():
    """Fast mode decides per wave, from the window's centre and a rigorous bound on its extent, whether its taps need the clamp;
    a wrong bound would send a border window down range-checked buffer loads that return silent zeros (or, in the global-load
    form, outside the texture).  Planes of several slants are moved in depth so that, view by view, the source windows of the
    border pixels sit from 0 to more than 3 extents inside each of the four borders — the numpy restatement of the decision
    asserts that pixels land within a quarter pixel of the threshold on BOTH sides of it, for each border — and the cost of every
    pixel must be bit-identical across the three gather forms, both kernels, and equal to the oracle's restatement of the fast
    arithmetic (S7), which always clamps."""
    imgs, K, R, t = _scene_rotated()
    ms = {}
    for name, (bg, mix) in (("buffer", (True, True)), ("global", (False, True)), ("bytes", (True, False))):
        m = _matcher(0, bg, mix_gather=mix)
        m.set_views(imgs, K, R, t)
        ms[name] = m
    near = np.zeros((4, 2), np.int64)             # per border: pixels just inside / just outside the decision's threshold
    scored = 0
    normals = [(0.0, 0.0, -1.0), (0.35, -0.2, -1.0), (-0.5, 0.3, -1.0)]
    depths = [40.0, 9.0, 4.0, 2.2, 1.45, 1.1, 0.83, 0.61, 0.47, 0.36, 0.3]
    for view in (1, 2, 3):
        orc = ol.Oracle(imgs, K, R, t, 1e-3, 1e6, box=11, n_best=1, seed=3, subset=[view], flags=ol.FLAGS_FAST_8BIT_IMAGERY)
        orc.set_rcp_table(ol.rcp_table_from_device(ms["buffer"]))
        for nv in normals:
            n = np.array(nv, np.float64)
            n /= np.linalg.norm(n)
            for z0 in depths:
                planes = np.zeros((H, W, 4), np.float32)
                planes[..., :3] = n.astype(np.float32)
                planes[..., 3] = np.float32(-n[2] * z0)             # the plane through (0, 0, z0) on the optical axis
                c_ref = orc.pm_cost_planes(planes)[0]
                got = {}
                for name, m in ms.items():
                    m.set_view_subset([view])
                    got[name + "_full"] = m.pm_cost_planes(planes)[0]
                    got[name + "_sweep"] = _sweep_cost(m, planes)
                for k, v in got.items():
                    assert np.array_equal(v, c_ref), (view, nv, z0, k, int((v != c_ref).sum()))
                scored += int((c_ref < 2.0).sum())
                for b, mg in enumerate(_centre_extent_margins(orc.homography(view, planes[0, 0]))):
                    near[b, 0] += int(((mg >= 0) & (mg < 0.25)).sum())
                    near[b, 1] += int(((mg < 0) & (mg > -0.25)).sum())
        assert not orc.rcp_out_of_range
    assert scored > 100000
    assert (near > 0).all(), near               # every border's threshold approached from both sides
    for m in ms.values():
        m.close()
