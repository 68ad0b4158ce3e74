"""A second, independent restatement of ONE red/black half-sweep — the control flow of rows A7 / A8 / A9 — in numpy float32, written
from the reference's text (gipuma_checkerboard_spatialProp_cu gipuma.cu:846-1050, spatialPropagation_cu :524-566,
gipuma_checkerboard_planeRefinement_cu :1053-1094, planeRefinement_cu :621-676, getRndDispAndUnitVector_cu :582-619, getD_cu :71-86,
getDisparity_cu / getDepthFromPlane3_cu :436-453, getViewVector_cu :97-105, the red/black wrappers :1096-1138) without following
oracle/tsar_oracle.c's code, against the C oracle's orc_pm_sweep.

gipuma.cu cannot be built here, so nothing pins these rows to outputs of the reference; test_oracle_independent_float64.py
restates the matching cost a second time, this file the loop around it: which pixels a launch touches, the eight arms in the
reference's order with their border tests and the two quirks, the strict `<` of the accept test after the depth-range test, that a
pixel's refinement starts from what its propagation left (the reference runs two kernels, the build one), the four uniforms of a
refinement step and what they perturb, the running step widths, the disparity the NEXT step starts from (the accepted hypothesis's own,
not the plane's), `+` in minDelta, the flip to the viewing hemisphere, getD_cu's plane offset.  Shared with the oracle, on purpose: the
score of a plane at a pixel (orc_pm_cost_multiview — restated independently in the float64 file), the uniforms (Philox, S1 — the
build's replacement of clock-seeded cuRAND), reads from the launch-start state (S2) and 1 / sqrtf for rsqrtf (S4).

Every expression below is one IEEE operation per operator in the reference's order (numpy float32 scalars), which is what the oracle
computes when built with -DORC_NO_FMA (every fmaf of its S4 written as multiply-then-add): against that build the restatement must agree
BIT FOR BIT; against the default build (S4's fused operations) to rounding.  A misreading of the text would have to be made twice, in two
languages and two loop structures, to go unseen.

Further down, the same way: rows A10 / A12 (get_disp, compute_disp, compute_disp_final, dptow, update_scale, update_scale_2), the matching
cost itself operation for operation (getCorrespondingPoint_cu :161-171, pmCost :229-298, pmCostMultiview_cu :455-518 with sort_small,
rlCost :300-392, gipuma_getlrdiff :1160-1186, gipuma_getview :1188-1213) and row A13 (gipuma_WMF :1499-1698, gipuma_WMF_Final :1294-1497).
What it found when it was written (end of round 5): the oracle and the kernels evaluated a tap's position as m[1] y + (m[0] x + m[2]) — the
line term hoisted with the constant folded in — where matvecmul4noz (config.h:150-162) forms (m[0] x + m[1] y) + m[2]; strict mode follows
the text since, and test_matching_cost_operation_for_operation fails on the first pixel if it does not."""
import numpy as np
import pytest

import oracle_lib as ol
from tsar_mvs_amd import synth

f32 = np.float32


class _Cam:
    """Camera_cu of the reference view as the oracle derived it (camera.h:9-33): the fields the sweep reads"""

    def __init__(self, cv):
        a = lambda v: np.array(list(v), dtype=np.float32)
        self.K, self.Minv, self.P34, self.C = a(cv.K), a(cv.Minv), a(cv.P34), a(cv.C)
        self.fx, self.alpha, self.f, self.baseline = f32(cv.fx), f32(cv.alpha), f32(cv.f), f32(cv.baseline)
        self.depthMin, self.depthMax = f32(cv.depthMin), f32(cv.depthMax)


def _matvec(m, v):                                       # matvecmul4, config.h:164-176: m[0] v.x + m[1] v.y + m[2] v.z, left to right
    return [m[3 * r] * v[0] + m[3 * r + 1] * v[1] + m[3 * r + 2] * v[2] for r in range(3)]


def _dot(a, b):                                          # dot4, config.h:36-38 (three components)
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _depth_of_plane(cam, n4, x, y):                      # getDisparity_cu -> getDepthFromPlane3_cu, gipuma.cu:436-453
    d = n4[3]
    if d != d:
        return f32(1000)
    den = (n4[0] * (f32(x) - cam.K[2])) + (n4[1] * (f32(y) - cam.K[5])) * cam.alpha + n4[2] * cam.fx
    return -d * cam.fx / den


def _plane_offset(cam, n, x, y, depth):                  # getD_cu, gipuma.cu:71-86
    pt = [depth * f32(x) - cam.P34[0], depth * f32(y) - cam.P34[1], depth - cam.P34[2]]
    return -_dot(n, _matvec(cam.Minv, pt))


def _normalize(v):                                       # normalize_cu, gipuma.cu:88-95 (rsqrtf -> 1 / sqrtf: S4)
    inv = f32(1) / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return [v[0] * inv, v[1] * inv, v[2] * inv]


def _view_vector(cam, x, y):                             # getViewVector_cu :97-105 over get3Dpoint_cu1 :56-66
    pt = [f32(x) - cam.P34[0], f32(y) - cam.P34[1], f32(1) - cam.P34[2]]
    v = _matvec(cam.Minv, pt)
    return _normalize([v[0] - cam.C[0], v[1] - cam.C[1], v[2] - cam.C[2]])


def _between(u, lo, hi):                                 # curand_between :113-116
    return u * (hi - lo) + lo


def _arms(c, x, y, cols, rows, quirks=True):
    """the neighbour each of the eight arms proposes, in the order the reference calls SPATIALPROPAGATION (gipuma.cu:888-1042); c is
    the cost plane, flat.  None = the arm's border test fails."""
    p = y * cols + x
    out = []
    # up_far :889-902
    if y > 2:
        best, at = c[p - 3 * cols], p - 3 * cols
        for i in range(1, 11):
            if y > 2 + 2 * i:
                q = p - 3 * cols - 2 * i * cols
                if c[q] < best:
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    # down_far :905-918 — the running minimum starts from c[up_far] (quirk; out of the image for y <= 2, where the build defines c[down_far])
    if y < rows - 3:
        best = c[p - 3 * cols] if (quirks and y > 2) else c[p + 3 * cols]
        at = p + 3 * cols
        for i in range(1, 11):
            if y < rows - 3 - 2 * i:
                q = p + 3 * cols + 2 * i * cols
                if c[q] < best:
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    # left_far :921-934
    if x > 2:
        best, at = c[p - 3], p - 3
        for i in range(1, 11):
            if x > 2 + 2 * i:
                q = p - 3 - 2 * i
                if c[q] < best:
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    # right_far :937-950 — `if (costMin < c[pointTemp])`: walks to the larger cost (quirk)
    if x < cols - 3:
        best, at = c[p + 3], p + 3
        for i in range(1, 11):
            if x < cols - 3 - 2 * i:
                q = p + 3 + 2 * i
                if (best < c[q]) if quirks else (c[q] < best):
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    # up_near :953-972
    if y > 0:
        near = p - cols
        best, at = c[near], near
        for i in range(3):
            if y > 1 + i and x > i:
                q = near - (1 + i) * cols - i
                if c[q] < best:
                    best, at = c[q], q
            if y > 1 + i and x < cols - 1 - i:
                q = near - (1 + i) * cols + i
                if c[q] < best:
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    # down_near :976-995
    if y < rows - 1:
        near = p + cols
        best, at = c[near], near
        for i in range(3):
            if y < rows - 2 - i and x > i:
                q = near + (1 + i) * cols - i
                if c[q] < best:
                    best, at = c[q], q
            if y < rows - 2 - i and x < cols - 1 - i:
                q = near + (1 + i) * cols + i
                if c[q] < best:
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    # left_near :999-1018
    if x > 0:
        near = p - 1
        best, at = c[near], near
        for i in range(3):
            if x > 1 + i and y > i:
                q = near - (1 + i) - i * cols
                if c[q] < best:
                    best, at = c[q], q
            if x > 1 + i and y < rows - 1 - i:
                q = near - (1 + i) + i * cols
                if c[q] < best:
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    # right_near :1022-1041
    if x < cols - 1:
        near = p + 1
        best, at = c[near], near
        for i in range(3):
            if x < cols - 2 - i and y > i:
                q = near + (1 + i) - i * cols
                if c[q] < best:
                    best, at = c[q], q
            if x < cols - 2 - i and y < rows - 1 - i:
                q = near + (1 + i) + i * cols
                if c[q] < best:
                    best, at = c[q], q
        out.append(at)
    else:
        out.append(None)
    return out


def half_sweep(orc, cam, colour, stream, seed, quirks=True):
    """one launch pair of the reference — gipuma_{black,red}_spatialProp_cu then gipuma_{black,red}_planeRefine_cu — on the pixels with
    (x + y) % 2 == colour (black: threadIdx.x even <=> p.y even, :1096-1105, i.e. x + y even), every read from the state as the launch
    found it (S2).  Returns the new (c, norm4, ratio, beview) planes; orc scores, nothing else."""
    rows, cols = orc.h, orc.w
    c0, n0 = orc.c.copy(), orc.norm4.copy()
    c1, n1, r1, b1 = c0.copy(), n0.copy(), orc.ratio.copy(), orc.beview.copy()
    cf, nf = c0.reshape(-1), n0.reshape(-1, 4)
    min_disp, max_disp = f32(orc.min_disp), f32(orc.max_disp)
    for y in range(rows):
        for x in range(cols):
            if (x + y) % 2 != colour:
                continue
            p = y * cols + x
            cost_now, norm_now = cf[p], nf[p].copy()
            disp_now = _depth_of_plane(cam, norm_now, x, y)
            # ---- propagation :846-1050
            for q in _arms(cf, x, y, cols, rows, quirks):
                if q is None:
                    continue
                norm_before = nf[q]
                disp_before = _depth_of_plane(cam, norm_before, x, y)
                cost_before, beview, ratio = orc.pm_cost_multiview(x, y, norm_before)
                if disp_before >= cam.depthMin and disp_before <= cam.depthMax:
                    if f32(cost_before) < cost_now:
                        disp_now, norm_now, cost_now = disp_before, norm_before.copy(), f32(cost_before)
                        r1[y, x], b1[y, x] = ratio, beview
            # ---- refinement :1053-1094 (its kernel re-reads c, norm4 and recomputes the disparity from the plane: the same values)
            disp_now = _depth_of_plane(cam, norm_now, x, y)
            vv = _view_vector(cam, x, y)
            deltaN = f32(1)
            deltaZ = max_disp / f32(2)
            step = 0
            while deltaZ >= f32(0.01):
                u = ol.rng4(seed, p, stream, step)
                disp = cam.f * cam.baseline / disp_now                       # disparityDepthConversion_cu :44-46
                minDelta = -min(deltaZ, min_disp + disp)
                maxDelta = min(deltaZ, max_disp - disp)
                dz = _between(u[0], minDelta, maxDelta)
                dispOut = min(max(disp + dz, min_disp), max_disp)
                dispOut = cam.f * cam.baseline / dispOut
                nt = [norm_now[0] + _between(u[1], -deltaN, deltaN), norm_now[1] + _between(u[2], -deltaN, deltaN),
                      norm_now[2] + _between(u[3], -deltaN, deltaN)]
                nt = _normalize(nt)
                if _dot(nt, vv) > f32(0):                                    # vecOnHemisphere_cu :106-112
                    nt = [-nt[0], -nt[1], -nt[2]]
                norm_temp = np.array([nt[0], nt[1], nt[2], _plane_offset(cam, nt, x, y, dispOut)], dtype=np.float32)
                cost_t, beview, ratio = orc.pm_cost_multiview(x, y, norm_temp)
                if f32(cost_t) < cost_now:
                    cost_now, disp_now, norm_now = f32(cost_t), dispOut, norm_temp
                    r1[y, x], b1[y, x] = ratio, beview
                deltaN = deltaN / f32(4)
                deltaZ = deltaZ / f32(10)
                step += 1
            c1[y, x], n1[y, x] = cost_now, norm_now
    return c1, n1, r1, b1


def random_init(orc, cam, seed):
    """gipuma_init_cu2, gipuma.cu:678-729, with rndUnitVectorOnHemisphere_cu :134-137 over rndUnitVectorSphereMarsaglia_cu :118-132.  How
    the rejection loop's uniforms are numbered is the build's own (S1: draw 0 = (disparity, x, y, -) of counter step 0; every further
    counter step holds two (x, y) attempts; after 16 steps the pole) — the reference draws from a clock-seeded XORWOW sequence."""
    rows, cols = orc.h, orc.w
    c1, n1 = np.empty((rows, cols), np.float32), np.empty((rows, cols, 4), np.float32)
    mind, maxd = f32(orc.min_disp), f32(orc.max_disp)
    for y in range(rows):
        for x in range(cols):
            p = y * cols + x
            vv = _view_vector(cam, x, y)
            u = ol.rng4(seed, p, 0, 0)
            disp_now = _between(u[0], mind, maxd)
            attempts = [(u[1], u[2])]
            for call in range(1, 16):
                w = ol.rng4(seed, p, 0, call)
                attempts += [(w[0], w[1]), (w[2], w[3])]
            a = b = total = f32(0)                                          # (what the build defines when every attempt is rejected)
            for ua, ub in attempts:
                xa, xb = _between(ua, f32(-1), f32(1)), _between(ub, f32(-1), f32(1))
                t = xa * xa + xb * xb                                       # get_pow2_norm, config.h:30
                if not (t >= f32(1)):
                    a, b, total = xa, xb, t
                    break
            sq = np.sqrt(f32(1) - total)
            n = [f32(2) * a * sq, f32(2) * b * sq, f32(1) - f32(2) * total]
            if _dot(n, vv) > f32(0):
                n = [-n[0], -n[1], -n[2]]
            depth = cam.f * cam.baseline / disp_now
            n4 = np.array([n[0], n[1], n[2], _plane_offset(cam, n, x, y, depth)], dtype=np.float32)
            n1[y, x] = n4
            c1[y, x] = orc.pm_cost_multiview(x, y, n4)[0]                   # (odd boxes: the init window box / 2 is the sweeps' (box - 1) / 2)
    return c1, n1


def _scene_and_oracle(nofma, flags=0, w=40, h=30, views=3, box=7, n_best=1):
    sc = synth.make_scene(w, h, views, seed=5)
    images = [im.cpu().numpy() for im in sc.images]
    orc = ol.Oracle(images, sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, seed=77, box=box, n_best=n_best, flags=flags, nofma=nofma)
    return sc, orc


# flags: the reference's quirks 1 / 2 as written (0), and both fixed (TSAR_FLAG_FIX_*: 3); the scripts' window with the best two of four views
@pytest.mark.parametrize("flags,views,box,n_best", [(0, 3, 7, 1), (3, 3, 7, 1), (0, 5, 11, 2)])
def test_two_iterations_against_the_restatement_bit_for_bit(flags, views, box, n_best):
    sc, orc = _scene_and_oracle(nofma=True, flags=flags, views=views, box=box, n_best=n_best)
    cam = _Cam(orc.camera(0))
    c_init, n_init = random_init(orc, cam, seed=77)
    orc.pm_init()
    assert np.array_equal(orc.norm4.view(np.uint32), n_init.view(np.uint32))
    assert np.array_equal(orc.c.view(np.uint32), c_init.view(np.uint32))
    d0 = np.array([[_depth_of_plane(cam, orc.norm4[y, x], x, y) for x in range(orc.w)] for y in range(orc.h)])
    assert (d0 > cam.depthMin * 0.999).all() and (d0 < cam.depthMax * 1.001).all()      # every initial plane passes through its pixel's ray inside the range
    launch = 0
    changed = 0
    for it in range(2):
        for colour in (0, 1):                   # black then red, gipuma.cu:1744-1751
            c1, n1, r1, b1 = half_sweep(orc, cam, colour, stream=1 + launch, seed=77, quirks=(flags == 0))
            before = orc.norm4.copy()
            orc.pm_sweep(colour)
            launch += 1
            assert np.array_equal(orc.c.view(np.uint32), c1.view(np.uint32))
            assert np.array_equal(orc.norm4.view(np.uint32), n1.view(np.uint32))
            assert np.array_equal(orc.beview, b1)
            assert np.array_equal(orc.ratio.view(np.uint32), r1.view(np.uint32))
            other = (np.add.outer(np.arange(orc.h), np.arange(orc.w)) % 2) != colour
            assert np.array_equal(before[other], orc.norm4[other])                   # the other colour is not touched
            changed += int((before != orc.norm4).any(-1).sum())
    assert changed > orc.h * orc.w // 2          # the sweeps did something


def test_default_build_agrees_to_rounding():
    """the oracle as the parity tests use it (S4's fused operations in the plane / depth helpers): same decisions wherever two costs are
    not within rounding of each other"""
    sc, orc = _scene_and_oracle(nofma=False)
    cam = _Cam(orc.camera(0))
    orc.pm_init()
    c1, n1, r1, b1 = half_sweep(orc, cam, 0, stream=1, seed=77)
    orc.pm_sweep(0)
    same = (np.abs(orc.norm4 - n1).max(-1) < 1e-4) & (np.abs(orc.c - c1) < 1e-5)
    assert same.mean() > 0.99                    # (0.994-0.997 measured: a handful of the 600 pixels accept or reject a refinement step whose gain is within rounding)


# ---- rows A10 / A12: the plane <-> depth kernels and the textureless fill, the same way ----

def test_plane_depth_kernels_and_textureless_fill_bit_for_bit():
    """host fill main.cpp:1479-1490 + gipuma_get_disp gipuma.cu:731-755, gipuma_compute_disp :810-844, gipuma_dptow :1140-1158,
    gipuma_update_scale :1215-1259, gipuma_update_scale_2 :1261-1292 — per-pixel numpy float32 from the text against the oracle built
    without fused operations"""
    sc, orc = _scene_and_oracle(nofma=True, w=36, h=26)
    cv = orc.camera(0)
    cam = _Cam(cv)
    Rorig, RorigInv = np.array(list(cv.Rorig), np.float32), np.array(list(cv.RorigInv), np.float32)
    rows, cols = orc.h, orc.w
    rng = np.random.default_rng(9)
    depth_in = rng.uniform(sc.depth_min, sc.depth_max, (rows, cols)).astype(np.float32)
    nw = rng.normal(size=(rows, cols, 3)).astype(np.float32)
    nw /= np.linalg.norm(nw, axis=-1, keepdims=True).astype(np.float32)
    fb = cam.f * cam.baseline

    # get_disp: lines->depth was filled with f b / depth (main.cpp:1488), lines->norm4 with the world normal; the kernel rotates the
    # normal by R_orig and takes the offset of the plane through the pixel's ray at f b / lines->depth
    orc.load_planes(depth_in, nw)
    want = np.empty((rows, cols, 4), np.float32)
    held = np.empty((rows, cols), np.float32)
    for y in range(rows):
        for x in range(cols):
            n = _matvec(Rorig, nw[y, x])
            held[y, x] = fb / depth_in[y, x]
            want[y, x] = [n[0], n[1], n[2], _plane_offset(cam, n, x, y, fb / held[y, x])]
    assert np.array_equal(orc.norm4.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(orc.depth.view(np.uint32), held.view(np.uint32))
    assert (orc.c == 1.0).all()                                               # main.cpp:1490

    # compute_disp: (R_orig_inv n, depth of the plane at the pixel), depth 0 where the cost is MAXCOST
    orc.c[3, 4] = 2.0
    out = orc.compute_disp()
    for y in range(rows):
        for x in range(cols):
            n4 = orc.norm4[y, x]
            o = _matvec(RorigInv, n4)
            d = _depth_of_plane(cam, n4, x, y) if orc.c[y, x] != f32(2) else f32(0)
            assert np.array_equal(np.array([o[0], o[1], o[2], d], np.float32).view(np.uint32), out[y, x].view(np.uint32)), (x, y)
    assert out[3, 4, 3] == 0.0 and np.abs(out[..., 3][orc.c != 2.0] / depth_in[orc.c != 2.0] - 1.0).max() < 1e-4      # the round trip returns the depths put in

    # update_scale: a pixel of a region flagged -1 takes the region's plane, turned to face the camera (all four components negated), cost 0,
    # scale 1; EVERY pixel's lines->depth becomes f b / (depth of its plane).  update_scale_2: fakedepth of the flagged regions' planes
    labels = (np.arange(rows)[:, None] // 9 * 4 + np.arange(cols)[None] // 9).astype(np.int32)
    n_regions = int(labels.max()) + 1
    text = np.where(np.arange(n_regions) % 3 == 0, -1.0, 1.0).astype(np.float32)
    planes = np.empty((n_regions, 4), np.float32)
    for r in range(n_regions):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        n = n.astype(np.float32)
        ys, xs = np.nonzero(labels == r)
        planes[r] = [n[0], n[1], n[2], _plane_offset(cam, n, int(xs[0]), int(ys[0]), f32(0.5 * (sc.depth_min + sc.depth_max)))]
    before = orc.norm4.copy()
    c_before = orc.c.copy()
    orc.set_regions(labels, text)
    orc.set_region_planes(planes)
    orc.fake_depth()
    fake = orc.fakedepth.copy()
    orc.update_scale()
    flipped = 0
    for y in range(rows):
        for x in range(cols):
            r = labels[y, x]
            n4 = before[y, x]
            if text[r] == -1:
                vv = _view_vector(cam, x, y)
                n4 = planes[r].copy()
                if n4[0] * vv[0] + n4[1] * vv[1] + n4[2] * vv[2] > f32(0):
                    n4 = -n4
                    flipped += 1
                assert orc.c[y, x] == 0.0 and orc.scale[y, x] == 1.0
                assert np.array_equal(np.array([_depth_of_plane(cam, n4, x, y)], np.float32).view(np.uint32), fake[y, x:x + 1].view(np.uint32))
            else:
                assert orc.c[y, x] == c_before[y, x]
            assert np.array_equal(orc.norm4[y, x].view(np.uint32), n4.view(np.uint32)), (x, y)
            assert np.array_equal(np.array([fb / _depth_of_plane(cam, n4, x, y)], np.float32).view(np.uint32), orc.depth[y, x:x + 1].view(np.uint32))
    assert flipped > 0

    # dptow: the offset of the plane through the pixel's ray at f b / lines->depth, normal kept
    orc.depth[...] = (fb / depth_in).astype(np.float32)
    kept = orc.norm4.copy()
    orc.depth_to_plane()
    for y in range(rows):
        for x in range(cols):
            w = _plane_offset(cam, kept[y, x, :3], x, y, fb / orc.depth[y, x])
            assert np.array_equal(orc.norm4[y, x].view(np.uint32), np.array([*kept[y, x, :3], w], np.float32).view(np.uint32)), (x, y)


# ---- rows A2 / A3 / A4: the matching cost itself in float32, operation for operation ----

def _warp(H, px, py):                                    # getCorrespondingPoint_cu :161-171: matvecmul4noz then vecdiv4 by the third component
    X = H[0] * f32(px) + H[1] * f32(py) + H[2]
    Y = H[3] * f32(px) + H[4] * f32(py) + H[5]
    Z = H[6] * f32(px) + H[7] * f32(py) + H[8]
    return X / Z, Y / Z


def pm_cost_f32(orc, img_ptrs, view, x, y, n4, hrad, vrad):
    """pmCost, gipuma.cu:229-298.  Shared with the oracle: the 3x3 homography (held to the reference's own macros by
    tests/test_reference_macros_golden.py), the bilinear fetch (S3: what replaces tex2D) and expf (S4's polynomial)."""
    import ctypes as C
    L = orc.L
    H = orc.homography(view, n4).reshape(-1)
    w, h = orc.w, orc.h
    ref = orc.images[0]
    tex = lambda v, u_, v_: f32(L.orc_bilinear(img_ptrs[v], w, h, C.c_float(u_), C.c_float(v_)))
    cen = ref[y, x]
    s_r = s_rr = s_s = s_ss = s_rs = wsum = f32(0)
    for i in range(-hrad, hrad + 1, 2):
        for j in range(-vrad, vrad + 1, 2):
            plx, ply = x + i, y + j
            r = ref[min(max(ply, 0), h - 1), min(max(plx, 0), w - 1)]          # tex2D at a texel centre, clamp addressing
            u_, v_ = _warp(H, plx, ply)
            s = tex(view, u_, v_)
            sd = np.sqrt(f32(i * i + j * j))
            cd = abs(r - cen)
            wt = f32(L.orc_expf(C.c_float(-sd / (f32(2) * f32(5) * f32(5)) - cd / (f32(2) * f32(3) * f32(3)))))      # (this build's expf: the polynomial's own fused operations are S4 too)
            s_r = s_r + wt * r
            s_rr = s_rr + wt * r * r
            s_s = s_s + wt * s
            s_ss = s_ss + wt * s * s
            s_rs = s_rs + wt * r * s
            wsum = wsum + wt
    inv = f32(1) / wsum
    s_r, s_rr, s_s, s_ss, s_rs = s_r * inv, s_rr * inv, s_s * inv, s_ss * inv, s_rs * inv
    var_r = s_rr - s_r * s_r
    var_s = s_ss - s_s * s_s
    if var_r < f32(1e-5) or var_s < f32(1e-5):
        return f32(2)
    covar = s_rs - s_r * s_s
    return max(f32(0), min(f32(2), f32(1) - covar / np.sqrt(var_r * var_s)))


def multiview_f32(costs, subset, n_best):
    """pmCostMultiview_cu, gipuma.cu:455-518 (COMB_BEST_N), from the per-view costs in subset order: cost, beview, ratio"""
    orig = [c if c < f32(2) else f32(2) for c in costs]
    valid = sum(1 for c in costs if c < f32(2))
    srt = list(orig)
    for i in range(1, len(srt)):                        # sort_small :425-434
        tmp, j = srt[i], i
        while j >= 1 and tmp < srt[j - 1]:
            srt[j] = srt[j - 1]
            j -= 1
        srt[j] = tmp
    nb = min(valid, n_best)
    if nb <= 0:
        return f32(2), -1, f32(0)
    cost = f32(0)
    for i in range(nb):
        cost = cost + srt[i]
    cost = cost / f32(nb)
    beview = -1
    for i, c in enumerate(orig):
        if srt[0] == c:
            beview = subset[i]                          # the LAST view attaining the minimum
    return cost, beview, srt[0] / srt[1] if len(srt) > 1 else f32(0)


@pytest.mark.parametrize("box,n_best", [(11, 1), (7, 2), (5, 3)])
def test_matching_cost_operation_for_operation(box, n_best):
    import ctypes as C
    sc, orc = _scene_and_oracle(nofma=True, w=48, h=36, views=4, box=box, n_best=n_best)
    ptrs = [im.ctypes.data_as(C.c_void_p) for im in orc.images]
    orc.L.orc_bilinear.restype = C.c_float
    orc.pm_init()
    orc.pm_iterate(1)                                                       # planes from random to nearly right
    planes = orc.norm4.copy()
    rng = np.random.default_rng(4)
    planes[::2, ::3] = planes[rng.integers(0, orc.h, planes[::2, ::3].shape[:2]), rng.integers(0, orc.w, planes[::2, ::3].shape[:2])]     # and some foreign ones
    rad = (box - 1) // 2
    subset = [1, 2, 3, 4]
    hit_low = hit_valid = 0
    for y in list(range(0, orc.h, 5)) + [orc.h - 1]:
        for x in list(range(0, orc.w, 7)) + [orc.w - 1]:                   # borders included: clamped taps
            per_view = []
            for v in subset:
                c = pm_cost_f32(orc, ptrs, v, x, y, planes[y, x], rad, rad)
                got = f32(orc.pm_cost(v, x, y, planes[y, x]))
                assert np.array([c], np.float32).view(np.uint32)[0] == np.array([got], np.float32).view(np.uint32)[0], (x, y, v, c, got)
                per_view.append(c)
                hit_low += c == f32(2)
                hit_valid += c < f32(2)
            want = multiview_f32(per_view, subset, n_best)
            cost, bv, rt = orc.pm_cost_multiview(x, y, planes[y, x])
            assert f32(cost) == want[0] and bv == want[1], (x, y, cost, bv, want)
            if want[1] >= 0:
                assert f32(rt) == want[2]
    assert hit_valid > 100


def rl_cost_f32(orc, img_ptrs, view, x, y, n4, hrad, vrad, rcp_table=None):
    """rlCost, gipuma.cu:300-392: `r` is the source image, `l` the reference image; same shared primitives as pm_cost_f32"""
    import ctypes as C
    L = orc.L
    H = orc.homography(view, n4).reshape(-1)
    w, h = orc.w, orc.h
    src = orc.images[view]
    tex = lambda v, u_, v_: f32(L.orc_bilinear(img_ptrs[v], w, h, C.c_float(u_), C.c_float(v_)))
    det = H[0] * H[4] * H[8] + H[1] * H[5] * H[6] + H[2] * H[3] * H[7] - H[2] * H[4] * H[6] - H[1] * H[3] * H[8] - H[0] * H[5] * H[7]
    V = [H[4] * H[8] - H[5] * H[7], H[1] * H[8] - H[2] * H[7], H[1] * H[5] - H[2] * H[4],
         H[3] * H[8] - H[5] * H[6], H[0] * H[8] - H[2] * H[6], H[0] * H[5] - H[2] * H[3],
         H[3] * H[7] - H[4] * H[6], H[0] * H[7] - H[1] * H[6], H[0] * H[4] - H[1] * H[3]]
    V = [V[0] / det, -V[1] / det, V[2] / det, -V[3] / det, V[4] / det, -V[5] / det, V[6] / det, -V[7] / det, V[8] / det]
    pcx, pcy = _warp(H, x, y)
    cen = tex(view, pcx, pcy)
    s_r = s_rr = s_s = s_ss = s_rs = wsum = f32(0)
    for i in range(-hrad, hrad + 1, 2):
        for j in range(-vrad, vrad + 1, 2):
            plx, ply = int(np.trunc(pcx + f32(i))), int(np.trunc(pcy + f32(j)))       # make_int2(pt_c.x + i, pt_c.y + j): toward zero
            r = src[min(max(ply, 0), h - 1), min(max(plx, 0), w - 1)]
            u_, v_ = _warp(V, plx, ply)
            if rcp_table is not None:             # the default arithmetic's one liberty here (oracle S7 (1)): the quotient as a product with rcp(Z)
                Xq = V[0] * f32(plx) + V[1] * f32(ply) + V[2]
                Yq = V[3] * f32(plx) + V[4] * f32(ply) + V[5]
                rz = _rcp(rcp_table, V[6] * f32(plx) + V[7] * f32(ply) + V[8])
                u_, v_ = Xq * rz, Yq * rz
            s = tex(0, u_, v_)
            sd = np.sqrt(f32(i * i + j * j))
            cd = abs(r - cen)
            wt = f32(L.orc_expf(C.c_float(-sd / (f32(2) * f32(5) * f32(5)) - cd / (f32(2) * f32(3) * f32(3)))))
            s_r = s_r + wt * r
            s_rr = s_rr + wt * r * r
            s_s = s_s + wt * s
            s_ss = s_ss + wt * s * s
            s_rs = s_rs + wt * r * s
            wsum = wsum + wt
    inv = f32(1) / wsum
    s_r, s_rr, s_s, s_ss, s_rs = s_r * inv, s_rr * inv, s_s * inv, s_ss * inv, s_rs * inv
    var_r = s_rr - s_r * s_r
    var_s = s_ss - s_s * s_s
    if var_r < f32(1e-5) or var_s < f32(1e-5):
        return f32(2)
    covar = s_rs - s_r * s_s
    return max(f32(0), min(f32(2), f32(1) - covar / np.sqrt(var_r * var_s)))


@pytest.mark.parametrize("fast", [False, True])
def test_lrdiff_operation_for_operation(fast):
    """gipuma_getlrdiff :1160-1186 over rlCost, and gipuma_getview :1188-1213, after an iteration: every pixel, bit for bit — in the
    reference's arithmetic and in the default one, where rlCost takes ONE liberty (the reciprocal of S7 (1))"""
    import ctypes as C
    sc, orc = _scene_and_oracle(nofma=True, w=44, h=32, views=4, box=7, n_best=1, flags=ol.FLAGS_FAST_8BIT_IMAGERY if fast else 0)
    table = None
    if fast:
        table = _rcp_table_correctly_rounded()
        orc.set_rcp_table(table)
    ptrs = [im.ctypes.data_as(C.c_void_p) for im in orc.images]
    orc.L.orc_bilinear.restype = C.c_float
    orc.L.orc_expf.restype = C.c_float
    cam = _Cam(orc.camera(0))
    orc.pm_init()
    orc.pm_iterate(1)
    planes, cost, bv = orc.norm4.copy(), orc.c.copy(), orc.beview.copy()
    orc.lrdiff_op()
    orc.getview()
    checked = 0
    for y in range(orc.h):
        for x in range(orc.w):
            if not (1 <= bv[y, x] <= 4):
                continue                                                    # (no accepted hypothesis yet: the build leaves lrdiff as it is)
            rc = rl_cost_f32(orc, ptrs, int(bv[y, x]), x, y, planes[y, x], 3, 3, rcp_table=table)
            d = abs(cost[y, x] - rc)
            d = f32(1) if d > f32(1) else d
            assert np.array([d], np.float32).view(np.uint32)[0] == orc.lrdiff[y, x:x + 1].view(np.uint32)[0], (x, y, d, orc.lrdiff[y, x])
            conf = ((f32(2) - cost[y, x]) / f32(2) + (f32(1) - d)) / f32(2)
            assert conf == orc.confid[y, x]
            depth = cam.f * cam.baseline / _depth_of_plane(cam, planes[y, x], x, y)
            assert depth == orc.depth[y, x]
            checked += 1
    assert checked > 0.8 * orc.h * orc.w


def test_compute_disp_final_operation_for_operation():
    """gipuma_compute_disp_final gipuma.cu:757-808: the multi-scale merge (take the coarser level's plane where the disparities differ by more
    than 6 on textured regions, or always on regions flagged -1), the clamp of the plane into the depth range, the output conversion"""
    sc, orc = _scene_and_oracle(nofma=True, w=36, h=26)
    cv = orc.camera(0)
    cam = _Cam(cv)
    RorigInv = np.array(list(cv.RorigInv), np.float32)
    rows, cols = orc.h, orc.w
    orc.pm_init()
    orc.pm_iterate(1)
    planes, cost = orc.norm4.copy(), orc.c.copy()
    cost[2, 3] = 2.0
    orc.c[2, 3] = 2.0
    rng = np.random.default_rng(12)
    resize4 = planes[rng.integers(0, rows, (rows, cols)), rng.integers(0, cols, (rows, cols))].copy()      # foreign planes: many out of range here
    resize4[::3] = planes[::3] * np.float32(1.0002)                                                          # and near copies
    text = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), (rows, cols))
    out = orc.compute_disp_final(resize4, text)
    fb = cam.f * cam.baseline
    taken = clamped = 0
    for y in range(rows):
        for x in range(cols):
            n4 = planes[y, x].copy()
            disp_now = fb / _depth_of_plane(cam, n4, x, y)
            disp_org = fb / _depth_of_plane(cam, resize4[y, x], x, y)
            if (abs(disp_now - disp_org) > f32(6) and text[y, x] == 1) or text[y, x] == -1:
                n4 = resize4[y, x].copy()
                taken += 1
            disp = _depth_of_plane(cam, n4, x, y)
            if disp > cam.depthMax:
                n4[3] = _plane_offset(cam, n4, x, y, cam.depthMax)
                clamped += 1
            if disp < cam.depthMin:
                n4[3] = _plane_offset(cam, n4, x, y, cam.depthMin)
                clamped += 1
            depth = _depth_of_plane(cam, n4, x, y)
            o = _matvec(RorigInv, n4)
            want = np.array([o[0], o[1], o[2], depth if cost[y, x] != f32(2) else f32(0)], np.float32)
            assert np.array_equal(want.view(np.uint32), out[y, x].view(np.uint32)), (x, y, want, out[y, x])
            assert np.array([depth], np.float32).view(np.uint32)[0] == orc.depth[y, x:x + 1].view(np.uint32)[0]
    assert taken > 100 and clamped > 20


# ---- row A13: the weighted median filters, from the text ----

def _bubble(vals, *carried):
    """the reference's sort, gipuma.cu:1564-1611 / :1369-1416, on ONE of its four lists: `num` outer passes whose inner loop reaches index
    num (the zero-initialised slot behind the list takes part); strict `>`; the carried arrays swap with the values"""
    num = len(vals) - 1
    for i in range(num):
        for j in range(num - i):
            if vals[j] > vals[j + 1]:
                vals[j], vals[j + 1] = vals[j + 1], vals[j]
                for c in carried:
                    c[j], c[j + 1] = c[j + 1], c[j]


def _stable(vals, *carried):
    """the same order without the quadratic loop: a strict-`>` bubble sort is a stable sort (asserted against _bubble on a sample)"""
    order = sorted(range(len(vals)), key=lambda k: vals[k])
    out = [[vals[k] for k in order]] + [[c[k] for k in order] for c in carried]
    return out


def _wmf_pixel(orc, L, cam, scale_in, depth_in, n_in, x, y, radius, gap, sdiv, literal):
    """taps, sort, weighted medians and the plane through the median-depth pixel of gipuma_WMF :1538-1676 / gipuma_WMF_Final :1333-1473.
    Returns (num, norm_mid or None)."""
    import ctypes as C
    rows, cols = orc.h, orc.w
    img = orc.images[0]
    w_, d_, n_, x_, y_, z_ = [], [], [], [], [], []
    for i in range(-radius, radius + 1, gap):
        for j in range(-radius, radius + 1, gap):
            px, py = x + i, y + j
            if 0 <= px < cols and 0 <= py < rows and scale_in[py, px] == 1:
                cen, refp = img[y, x], img[py, px]
                color_dist = abs(refp - cen)
                spatial_dist = np.sqrt(f32(i * i + j * j)) / f32(sdiv)
                wt = f32(L.orc_expf(C.c_float(-spatial_dist / (f32(2) * f32(2))))) * f32(L.orc_expf(C.c_float(-color_dist / (f32(3) * f32(3)))))
                w_.append(wt); d_.append(depth_in[py, px]); n_.append(py * cols + px)
                x_.append(n_in[py, px, 0]); y_.append(n_in[py, px, 1]); z_.append(n_in[py, px, 2])
    num = len(w_)
    if num == 0:
        return 0, None
    z0 = f32(0)
    lists = [(d_ + [z0], w_ + [z0], n_ + [0]), (x_ + [z0], list(w_) + [z0]), (y_ + [z0], list(w_) + [z0]), (z_ + [z0], list(w_) + [z0])]
    if literal:
        for l in lists:
            _bubble(*l)
        (d, w, n), (xs, w1), (ys, w2), (zs, w3) = lists
    else:
        (d, w, n), (xs, w1), (ys, w2), (zs, w3) = [_stable(*l) for l in lists]
    wSum = f32(0)
    for i in range(num):
        wSum = wSum + w[i]
    half = wSum / f32(2)

    def median(vals, wts):
        acc = f32(0)
        for i in range(num):
            acc = acc + wts[i]
            if acc >= half:
                return vals[i]
        return vals[num - 1]          # (the reference leaves norm_mid uninitialised here; the build defines the last sorted element)
    nm = [median(xs, w1), median(ys, w2), median(zs, w3)]
    acc = f32(0)
    for i in range(num):
        acc = acc + w[i]
        if acc >= half:
            weimid = n[i]
            disp_mid = cam.f * cam.baseline / depth_in.reshape(-1)[weimid]
            nrm = np.float64(np.sqrt(nm[0] * nm[0] + nm[1] * nm[1] + nm[2] * nm[2]))       # `double xyzsqr = sqrtf(...)`
            nm = [f32(np.float64(nm[0]) / nrm), f32(np.float64(nm[1]) / nrm), f32(np.float64(nm[2]) / nrm)]
            return num, np.array([nm[0], nm[1], nm[2], _plane_offset(cam, nm, weimid % cols, weimid // cols, disp_mid)], np.float32)
    return num, None


def test_weighted_median_filters_from_the_text(small_scene):
    import ctypes as C
    sc = small_scene                                      # 96 x 64, 3 source views
    images = [im.numpy() for im in sc.images]
    orc = ol.Oracle(images, sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, seed=5, box=7, n_best=1, nofma=True)
    L = orc.L
    L.orc_expf.restype = C.c_float
    cam = _Cam(orc.camera(0))
    rows, cols = orc.h, orc.w
    fb = cam.f * cam.baseline
    orc.pm_init()
    orc.pm_iterate(2)
    orc.getview()                                         # lines->depth = f b / depth of the plane
    rng = np.random.default_rng(21)
    orc.scale[...] = (rng.random((rows, cols)) < 0.85).astype(np.float32)
    sample = {(int(a), int(b)) for a, b in zip(rng.integers(0, cols, 12), rng.integers(0, rows, 12))}
    # ---- gipuma_WMF :1499-1698, the last two of its four passes (tap grids of radius 20 and 10)
    for it in (2, 3):
        po, repo = 2 ** it, 2 ** (3 - it)
        radius, gap, ths = 80 // po, 16 // po, 24 // po
        scale_in, depth_in, n_in = orc.scale.copy(), orc.depth.copy(), orc.norm4.copy()
        orc.wmf_detect(it)
        flagged = 0
        for y in range(rows):
            for x in range(cols):
                num, nm = _wmf_pixel(orc, L, cam, scale_in, depth_in, n_in, x, y, radius, gap, repo, literal=False)
                if (x, y) in sample:
                    num2, nm2 = _wmf_pixel(orc, L, cam, scale_in, depth_in, n_in, x, y, radius, gap, repo, literal=True)
                    assert num2 == num and (nm is None) == (nm2 is None) and (nm is None or np.array_equal(nm.view(np.uint32), nm2.view(np.uint32)))
                want = f32(0)
                if num > 0 and nm is not None:
                    disp_now = fb / _depth_of_plane(cam, nm, x, y)
                    disp_org = fb / _depth_of_plane(cam, n_in[y, x], x, y)
                    want = f32(0) if abs(disp_now - disp_org) > f32(ths) else f32(1)       # DEPTH_THS_MIN / MAX are 0 (:38-39): never true
                assert orc.scale[y, x] == want, (it, x, y, num, want)
                flagged += want == 0
        assert 0 < flagged < rows * cols
    # ---- gipuma_WMF_Final :1294-1497: unreliable pixels of textured regions are refilled where enough reliable taps surround them
    orc.set_regions(np.zeros((rows, cols), np.int32), np.array([1.0], np.float32))
    filled = 0
    for it in (0, 1):
        po = 2 ** it
        radius, gap, ths = 5 * po, po, 32 // po
        scale_in, depth_in, n_in = orc.scale.copy(), orc.depth.copy(), orc.norm4.copy()
        orc.wmf_fill(it)
        for y in range(rows):
            for x in range(cols):
                if scale_in[y, x] != 0:
                    assert np.array_equal(orc.norm4[y, x].view(np.uint32), n_in[y, x].view(np.uint32)) and orc.scale[y, x] == scale_in[y, x]
                    continue
                num, nm = _wmf_pixel(orc, L, cam, scale_in, depth_in, n_in, x, y, radius, gap, po, literal=False)
                if num < ths or nm is None:
                    assert np.array_equal(orc.norm4[y, x].view(np.uint32), n_in[y, x].view(np.uint32)) and orc.scale[y, x] == 0, (it, x, y, num)
                    continue
                assert np.array_equal(orc.norm4[y, x].view(np.uint32), nm.view(np.uint32)), (it, x, y)
                disp = fb / _depth_of_plane(cam, nm, x, y)
                if disp <= f32(orc.min_disp) or disp >= f32(orc.max_disp):
                    assert orc.scale[y, x] == 0 and orc.depth[y, x] == f32(orc.min_disp)
                else:
                    assert orc.scale[y, x] == 1 and orc.depth[y, x] == disp
                    filled += 1
    assert filled > 10


# ---- the default ("fast") arithmetic: the reference's text with the seven liberties of oracle S7 applied, and nothing else ----

def _rcp_table_correctly_rounded():
    """a stand-in for the device's v_rcp_f32 table (the GPU tests read the real one): the correctly rounded reciprocal of every
    mantissa of [1, 2) — any table serves a CPU test of WHERE the reciprocal enters"""
    m = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
    return (np.float32(1) / m).astype(np.float32)


def _rcp(table, x):                                      # S7 (1): table on the mantissa, exponent and sign exact
    bits = np.array([x], np.float32).view(np.uint32)[0]
    e, m = int((bits >> 23) & 0xFF), int(bits & 0x7FFFFF)
    r = np.array([table[m]], np.float32).view(np.uint32)[0]
    re = int((r >> 23) & 0xFF) + 127 - e
    assert 0 < re < 255 and 0 < e < 255
    out = (int(bits) & 0x80000000) | (re << 23) | (int(r) & 0x7FFFFF)
    return np.array([out], np.uint32).view(np.float32)[0]


def _bilinear_differences(img, u, v):
    """S7 (4) + (6): clamp to the image (the same sample as tex2D's clamp addressing), the blend (t00 + ax d1) + ay (d2 + ax d3)"""
    h, w = img.shape
    u = min(max(u, f32(-1)), f32(w))
    v = min(max(v, f32(-1)), f32(h))
    fu, fv = np.floor(u), np.floor(v)
    ax, ay = u - fu, v - fv
    x0, y0 = int(fu), int(fv)
    t = lambda xx, yy: img[min(max(yy, 0), h - 1), min(max(xx, 0), w - 1)]
    t00, t10, t01, t11 = t(x0, y0), t(x0 + 1, y0), t(x0, y0 + 1), t(x0 + 1, y0 + 1)
    d1, d2 = t10 - t00, t01 - t00
    d3 = (t11 - t01) - d1
    return ay * (ax * d3 + d2) + (ax * d1 + t00)


def pm_cost_fast_f32(orc, cv0, cvv, table, view, x, y, n4, hrad, vrad, rows=True):
    """pm_cost_f32 above with the seven liberties of oracle S7 and nothing else changed"""
    import ctypes as C
    L = orc.L
    w, h = orc.w, orc.h
    ref, src = orc.images[0], orc.images[view]
    Kinv = np.array(list(cv0.Kinv), np.float32)
    A, b = np.array(list(cvv.A), np.float32), np.array(list(cvv.b), np.float32)
    inv_d = _rcp(table, n4[3])                                                        # (2): H = A - b m^T, m = K_ref^-T n * rcp(d)
    m = [((n4[0] * Kinv[c] + n4[1] * Kinv[3 + c]) + n4[2] * Kinv[6 + c]) * inv_d for c in range(3)]
    H = [(-b[r]) * m[c] + A[3 * r + c] for r in range(3) for c in range(3)]
    cen = ref[y, x]
    s_r = s_rr = wsum = f32(0)
    wts = {}
    for i in range(-hrad, hrad + 1, 2):                                               # the reference terms: the text's order
        for j in range(-vrad, vrad + 1, 2):
            r = ref[min(max(y + j, 0), h - 1), min(max(x + i, 0), w - 1)]
            sd = np.sqrt(f32(i * i + j * j))
            wt = f32(L.orc_expf(C.c_float(-sd / (f32(2) * f32(5) * f32(5)) - abs(r - cen) / (f32(2) * f32(3) * f32(3)))))
            wts[(i, j)] = (wt, r)
            s_r = s_r + wt * r
            s_rr = s_rr + wt * r * r
            wsum = wsum + wt
    s_s = s_ss = s_rs = f32(0)
    # (5): on 8-bit imagery the source sums run row by row (x fastest); float imagery keeps the text's columns
    lines = [(a, b_) for a in range(-(vrad if rows else hrad), (vrad if rows else hrad) + 1, 2) for b_ in range(-(hrad if rows else vrad), (hrad if rows else vrad) + 1, 2)]
    for a, b_ in lines:
        i, j = (b_, a) if rows else (a, b_)
        xi, yj = f32(x + i), f32(y + j)
        # (7): the line term with the constant folded in, then the coordinate that runs along the line
        if rows:
            X, Y, Z = H[0] * xi + (H[1] * yj + H[2]), H[3] * xi + (H[4] * yj + H[5]), H[6] * xi + (H[7] * yj + H[8])
        else:
            X, Y, Z = H[1] * yj + (H[0] * xi + H[2]), H[4] * yj + (H[3] * xi + H[5]), H[7] * yj + (H[6] * xi + H[8])
        rz = _rcp(table, Z)                                                           # (1)
        s = _bilinear_differences(src, X * rz, Y * rz)                                # (4), (6)
        wt, r = wts[(i, j)]
        ws = wt * s
        s_s = s_s + ws
        s_ss = s_ss + ws * s
        s_rs = s_rs + ws * r                                                          # (3): (w s) r
    inv = f32(1) / wsum
    s_r, s_rr, s_s, s_ss, s_rs = s_r * inv, s_rr * inv, s_s * inv, s_ss * inv, s_rs * inv
    var_r = s_rr - s_r * s_r
    var_s = s_ss - s_s * s_s
    if var_r < f32(1e-5) or var_s < f32(1e-5):
        return f32(2)
    return max(f32(0), min(f32(2), f32(1) - (s_rs - s_r * s_s) / np.sqrt(var_r * var_s)))


@pytest.mark.parametrize("rows", [True, False])
def test_fast_arithmetic_is_the_text_plus_its_seven_liberties(rows):
    """oracle S7 says the default arithmetic of the library is the reference's algorithm with seven rounding-level liberties.  Here the
    float32 restatement of the text takes exactly those seven and must then equal the oracle's fast cost bit for bit (unfused build,
    8-bit imagery's row order): the list is complete."""
    import ctypes as C
    sc = synth.make_scene(48, 36, 4, seed=5)
    images = [np.rint(im.cpu().numpy()).astype(np.float32) for im in sc.images]       # 8-bit imagery: exact integer texels
    orc = ol.Oracle(images, sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, seed=77, box=11, n_best=1, flags=ol.FLAGS_FAST_8BIT_IMAGERY if rows else ol.FLAG_FAST_ARITH, nofma=True)
    table = _rcp_table_correctly_rounded()
    orc.set_rcp_table(table)
    orc.L.orc_expf.restype = C.c_float
    orc.pm_init()
    orc.pm_iterate(1)
    planes = orc.norm4.copy()
    cv0 = orc.camera(0)
    checked = 0
    for y in list(range(0, orc.h, 5)) + [orc.h - 1]:
        for x in list(range(0, orc.w, 7)) + [orc.w - 1]:
            for v in (1, 2, 3, 4):
                c = pm_cost_fast_f32(orc, cv0, orc.camera(v), table, v, x, y, planes[y, x], 5, 5, rows=rows)
                got = f32(orc.pm_cost(v, x, y, planes[y, x]))
                assert np.array([c], np.float32).view(np.uint32)[0] == np.array([got], np.float32).view(np.uint32)[0], (x, y, v, c, got)
                checked += c < f32(2)
    assert checked > 100 and not orc.rcp_out_of_range


def test_even_box_init_scores_on_box_over_two():
    """gipuma_init_cu2 takes box / 2 as its window radius (gipuma.cu:693-694), the sweeps (box - 1) / 2 (:858-859): with an even box the
    initial costs come from a larger window than every later one"""
    import ctypes as C
    sc, orc = _scene_and_oracle(nofma=True, w=30, h=24, views=3, box=8, n_best=1)
    ptrs = [im.ctypes.data_as(C.c_void_p) for im in orc.images]
    orc.L.orc_bilinear.restype = C.c_float
    orc.L.orc_expf.restype = C.c_float
    cam = _Cam(orc.camera(0))
    _, n_init = random_init(orc, cam, seed=77)          # (its costs use the sweeps' radius: not compared here)
    orc.pm_init()
    assert np.array_equal(orc.norm4.view(np.uint32), n_init.view(np.uint32))
    differ = 0
    for y in range(0, orc.h, 3):
        for x in range(0, orc.w, 3):
            big = multiview_f32([pm_cost_f32(orc, ptrs, v, x, y, n_init[y, x], 4, 4) for v in (1, 2, 3)], [1, 2, 3], 1)[0]
            small = multiview_f32([pm_cost_f32(orc, ptrs, v, x, y, n_init[y, x], 3, 3) for v in (1, 2, 3)], [1, 2, 3], 1)[0]
            assert orc.c[y, x] == big, (x, y, orc.c[y, x], big, small)
            assert f32(orc.pm_cost_multiview(x, y, n_init[y, x])[0]) == small
            differ += big != small
    assert differ > 20
