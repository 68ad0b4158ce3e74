"""ctypes binding of the CPU oracle (oracle/libtsar_oracle.so).  Test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libtsar_oracle.so")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


ORACLE_NOFMA_SO = os.path.join(ORACLE_DIR, "libtsar_oracle_nofma.so")


def nofma_lib():
    """tsar_oracle.c built with every fmaf() as multiply-then-add (-DORC_NO_FMA): the form that can be compared bit for bit with
    the reference's own config.h macros compiled without contraction (tests/test_reference_macros_golden.py)"""
    src = os.path.join(ORACLE_DIR, "tsar_oracle.c")
    if not os.path.exists(ORACLE_NOFMA_SO) or os.path.getmtime(src) > os.path.getmtime(ORACLE_NOFMA_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libtsar_oracle_nofma.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(ORACLE_NOFMA_SO)
    for n in ("orc_homography_arrays", "orc_mat3mul", "orc_mat3vec"):
        getattr(L, n).restype = None
    return L


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
    stale = (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libtsar_oracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_lib = None


class _Partial:
    """a CDLL whose missing symbols bind to nothing: libtsar_oracle_nofma.so holds tsar_oracle.c only (no SLIC / texture / fusion)"""

    class _Sink:
        restype = None
        argtypes = None

    def __init__(self, L):
        object.__setattr__(self, "_L", L)

    def __getattr__(self, name):
        try:
            return getattr(self._L, name)
        except AttributeError:
            return _Partial._Sink()


def _bind(L):
    """prototypes of the oracle's entry points on a loaded library (the default build or the -DORC_NO_FMA one)"""
    real = L
    L = _Partial(real)
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int, C.c_int]
    L.orc_destroy.argtypes = [C.c_void_p]
    for name in ("c", "norm4", "ratio", "depth", "scale", "lrdiff", "confid", "fakedepth"):
        fn = getattr(L, "orc_plane_" + name)
        fn.restype = _f32p
        fn.argtypes = [C.c_void_p]
    L.orc_plane_beview.restype = _i32p
    L.orc_plane_beview.argtypes = [C.c_void_p]
    L.orc_expf.restype = C.c_float
    L.orc_expf.argtypes = [C.c_float]
    L.orc_bilinear.restype = C.c_float
    L.orc_bilinear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.orc_bilinear_q8.restype = C.c_float
    L.orc_bilinear_q8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.orc_pm_cost.restype = C.c_float
    L.orc_pm_cost.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_pm_cost_multiview.restype = C.c_float
    L.orc_pm_cost_multiview.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_getD.restype = C.c_float
    L.orc_getD.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.orc_depth_from_plane.restype = C.c_float
    L.orc_depth_from_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.orc_min_disp.restype = C.c_float
    L.orc_max_disp.restype = C.c_float
    L.orc_min_disp.argtypes = [C.c_void_p]
    L.orc_max_disp.argtypes = [C.c_void_p]
    L.orc_camera_ptr.restype = C.c_void_p
    L.orc_camera_ptr.argtypes = [C.c_void_p, C.c_int]
    L.orc_set_params.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64]
    L.orc_derive_cameras.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float]
    L.orc_rng4.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    for n in ("orc_set_image", "orc_set_subset", "orc_pm_init", "orc_pm_sweep", "orc_pm_sweep_rects", "orc_pm_iterate", "orc_pm_iterate_final", "orc_pm_cost_planes",
              "orc_select_candidates", "orc_homography", "orc_view_vector", "orc_load_planes", "orc_compute_disp",
              "orc_depth_to_plane", "orc_compute_disp_final", "orc_lrdiff", "orc_getview", "orc_fake_depth",
              "orc_update_scale", "orc_set_regions", "orc_set_region_planes", "orc_set_launch", "orc_wmf_detect", "orc_wmf_fill",
              "orc_ransac_regions", "orc_slic", "orc_rgb2lab", "orc_philox_raw"):
        getattr(L, n).restype = None
    L.orc_region_planes.restype = _f32p
    L.orc_region_planes.argtypes = [C.c_void_p]
    L.orc_ransac_points.restype = C.c_int
    L.orc_ransac_points.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.orc_pow_third.restype = C.c_float
    L.orc_pow_third.argtypes = [C.c_float]
    L.orc_pow_third_check.restype = None
    L.orc_pow_third_check.argtypes = [C.c_void_p]
    L.orc_slic_distance.restype = C.c_float
    L.orc_slic_distance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float]
    L.orc_slic_blocks_per_spixel.restype = C.c_int
    L.orc_slic_blocks_per_spixel.argtypes = [C.c_int]
    for n, at in (("orc_slic_convert", [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
                  ("orc_slic_init_centers", [C.c_void_p, C.c_void_p] + [C.c_int] * 5),
                  ("orc_slic_find_association", [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_float]),
                  ("orc_slic_partials", [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5),
                  ("orc_slic_finalize", [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
                  ("orc_slic_update_centers", [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5),
                  ("orc_slic_connectivity", [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
                  ("orc_slic_from_lab", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
                  ("orc_homography_arrays", [C.c_void_p] * 6), ("orc_mat3mul", [C.c_void_p] * 3), ("orc_mat3vec", [C.c_void_p] * 3)):
        getattr(L, n).restype = None
        getattr(L, n).argtypes = at
    L.orc_slic.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_refine_steps.restype = C.c_int
    L.orc_refine_steps.argtypes = [C.c_void_p]
    L.orc_set_rcp_table.restype = None
    L.orc_set_rcp_table.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_rcp_out_of_range.restype = C.c_int
    L.orc_rcp_out_of_range.argtypes = [C.c_void_p]
    L.orc_rcp_gpu.restype = C.c_float
    L.orc_rcp_gpu.argtypes = [C.c_void_p, C.c_float]
    return real


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = _bind(C.CDLL(ORACLE_SO))
    return _lib


_lib_nofma = None


def lib_nofma():
    """the whole of tsar_oracle.c with every fmaf() as multiply-then-add, bound like lib(): PatchMatch only"""
    global _lib_nofma
    if _lib_nofma is None:
        nofma_lib()                                   # (builds it when stale)
        _lib_nofma = _bind(C.CDLL(ORACLE_NOFMA_SO))
    return _lib_nofma


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CameraView(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("Kinv", C.c_float * 9), ("R", C.c_float * 9), ("t", C.c_float * 3),
                ("Minv", C.c_float * 9), ("P34", C.c_float * 3), ("C", C.c_float * 3),
                ("Rorig", C.c_float * 9), ("RorigInv", C.c_float * 9),
                ("fx", C.c_float), ("fy", C.c_float), ("f", C.c_float), ("alpha", C.c_float),
                ("baseline", C.c_float), ("depthMin", C.c_float), ("depthMax", C.c_float),
                ("A", C.c_float * 9), ("b", C.c_float * 3)]


# oracle/tsar_oracle.c S7: the arithmetic of the HIP library's default mode (needs the device's v_rcp_f32 table, set_rcp_table)
FLAG_FAST_ARITH, FLAG_ROW_ORDER = 128, 256
FLAGS_FAST_8BIT_IMAGERY = FLAG_FAST_ARITH | FLAG_ROW_ORDER      # what the production kernels for 8-bit imagery compute


class Oracle:
    """Host-side mirror of the matcher state used by the parity tests."""

    def __init__(self, images, K, R, t, depth_min, depth_max, box=11, n_best=1, cost_comb=1, flags=0, seed=2024,
                 cam_scale=1.0, subset=None, box_v=None, nofma=False):
        L = lib_nofma() if nofma else lib()
        self.L = L
        self.images = [np.ascontiguousarray(np.asarray(im, dtype=np.float32)) for im in images]
        self.h, self.w = self.images[0].shape
        self.n_views = len(self.images)
        self.s = C.c_void_p(L.orc_create(self.w, self.h))
        for v, im in enumerate(self.images):
            L.orc_set_image(self.s, C.c_int(v), _p(im))
        K = np.ascontiguousarray(K, dtype=np.float32)
        R = np.ascontiguousarray(R, dtype=np.float32)
        t = np.ascontiguousarray(t, dtype=np.float32)
        L.orc_derive_cameras(self.s, self.n_views, _p(K), _p(R), _p(t), C.c_float(cam_scale), C.c_float(depth_min), C.c_float(depth_max))
        L.orc_set_params(self.s, box, box if box_v is None else box_v, n_best, cost_comb, flags, seed)
        if subset is None:
            subset = list(range(1, self.n_views))
        sub = np.asarray(subset, dtype=np.int32)
        L.orc_set_subset(self.s, len(sub), _p(sub))
        self.n_sel = len(sub)

    def __del__(self):
        try:
            self.L.orc_destroy(self.s)
        except Exception:
            pass

    # ---- raw plane views (numpy arrays aliasing the oracle's memory) ----
    def _plane(self, name, shape, dtype=np.float32):
        ptr = getattr(self.L, "orc_plane_" + name)(self.s)
        return np.ctypeslib.as_array(ptr, shape=shape)

    @property
    def c(self):
        return self._plane("c", (self.h, self.w))

    @property
    def norm4(self):
        return self._plane("norm4", (self.h, self.w, 4))

    @property
    def ratio(self):
        return self._plane("ratio", (self.h, self.w))

    @property
    def beview(self):
        return self._plane("beview", (self.h, self.w))

    @property
    def depth(self):
        return self._plane("depth", (self.h, self.w))

    @property
    def scale(self):
        return self._plane("scale", (self.h, self.w))

    @property
    def lrdiff(self):
        return self._plane("lrdiff", (self.h, self.w))

    @property
    def confid(self):
        return self._plane("confid", (self.h, self.w))

    @property
    def fakedepth(self):
        return self._plane("fakedepth", (self.h, self.w))

    def camera(self, view) -> CameraView:
        return CameraView.from_address(self.L.orc_camera_ptr(self.s, view))

    @property
    def min_disp(self):
        return self.L.orc_min_disp(self.s)

    @property
    def max_disp(self):
        return self.L.orc_max_disp(self.s)

    # ---- operators ----
    def pm_cost(self, view, x, y, n4):
        n4 = np.ascontiguousarray(n4, dtype=np.float32)
        return float(self.L.orc_pm_cost(self.s, view, x, y, _p(n4)))

    def pm_cost_multiview(self, x, y, n4):
        n4 = np.ascontiguousarray(n4, dtype=np.float32)
        bv = C.c_int(0)
        rt = C.c_float(0)
        c = self.L.orc_pm_cost_multiview(self.s, x, y, _p(n4), C.byref(bv), C.byref(rt))
        return float(c), bv.value, rt.value

    def pm_cost_planes(self, planes):
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        cost = np.empty((self.h, self.w), np.float32)
        bv = np.empty((self.h, self.w), np.int32)
        rt = np.empty((self.h, self.w), np.float32)
        self.L.orc_pm_cost_planes(self.s, _p(planes), _p(cost), _p(bv), _p(rt))
        return cost, bv, rt

    def homography(self, view, n4):
        n4 = np.ascontiguousarray(n4, dtype=np.float32)
        H = np.empty(9, np.float32)
        self.L.orc_homography(self.s, C.c_int(view), _p(n4), _p(H))
        return H.reshape(3, 3)

    def getD(self, n, x, y, depth):
        n = np.ascontiguousarray(n, dtype=np.float32)
        return float(self.L.orc_getD(self.s, _p(n), x, y, C.c_float(depth)))

    def depth_from_plane(self, n4, x, y):
        n4 = np.ascontiguousarray(n4, dtype=np.float32)
        return float(self.L.orc_depth_from_plane(self.s, _p(n4), x, y))

    def view_vector(self, x, y):
        v = np.empty(3, np.float32)
        self.L.orc_view_vector(self.s, C.c_int(x), C.c_int(y), _p(v))
        return v

    def select_candidates(self, c, x, y):
        c = np.ascontiguousarray(c, dtype=np.float32)
        out = np.empty(8, np.int32)
        self.L.orc_select_candidates(self.s, _p(c), C.c_int(x), C.c_int(y), _p(out))
        return out

    def set_subset(self, subset):
        sub = np.asarray(subset, dtype=np.int32)
        self.L.orc_set_subset(self.s, len(sub), _p(sub))
        self.n_sel = len(sub)

    def set_rcp_table(self, table):
        """the device's v_rcp_f32 results for the 2^23 mantissas of [1, 2) (rcp_table_from_device): needed by FLAG_FAST_ARITH"""
        table = np.ascontiguousarray(table, np.float32)
        assert table.size == 1 << 23
        self._rcp_table = table              # borrowed by the C side
        self.L.orc_set_rcp_table(self.s, _p(table))

    @property
    def rcp_out_of_range(self):
        return bool(self.L.orc_rcp_out_of_range(self.s))

    def pm_init(self):
        self.L.orc_pm_init(self.s)

    def pm_sweep(self, colour, do_prop=1, do_refine=1):
        self.L.orc_pm_sweep(self.s, C.c_int(colour), C.c_int(do_prop), C.c_int(do_refine))

    def pm_sweep_rects(self, colour, rects, do_prop=1, do_refine=1):
        """the same launch restricted to the pixels inside the rectangles [(x0, y0, x1, y1), ...] (reads see the whole image)"""
        r = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
        self.L.orc_pm_sweep_rects(self.s, C.c_int(colour), C.c_int(do_prop), C.c_int(do_refine), C.c_int(len(r)), _p(r))

    def pm_iterate(self, iters):
        self.L.orc_pm_iterate(self.s, C.c_int(iters))

    def pm_iterate_final(self, iters, text):
        """the kernels' `final == true` mode (reference gipuma.cu:856,1063,559-562,669-672); text [h][w] = lines->text"""
        tx = np.ascontiguousarray(text, np.float32)
        self.L.orc_pm_iterate_final(self.s, C.c_int(iters), _p(tx))

    def set_launch(self, n):
        self.L.orc_set_launch(self.s, C.c_int(n))

    def refine_steps(self):
        return int(self.L.orc_refine_steps(self.s))

    def load_planes(self, depth, normal_world):
        d = np.ascontiguousarray(depth, np.float32)
        n = np.ascontiguousarray(normal_world, np.float32)
        self.L.orc_load_planes(self.s, _p(d), _p(n))

    def compute_disp(self):
        out = np.empty((self.h, self.w, 4), np.float32)
        self.L.orc_compute_disp(self.s, _p(out))
        return out

    def compute_disp_final(self, resize4, text):
        r = np.ascontiguousarray(resize4, np.float32)
        tx = np.ascontiguousarray(text, np.float32)
        out = np.empty((self.h, self.w, 4), np.float32)
        self.L.orc_compute_disp_final(self.s, _p(r), _p(tx), _p(out))
        return out

    def depth_to_plane(self):
        self.L.orc_depth_to_plane(self.s)

    def lrdiff_op(self):
        self.L.orc_lrdiff(self.s)

    def getview(self):
        self.L.orc_getview(self.s)

    def set_regions(self, labels, text, size=None):
        lb = np.ascontiguousarray(labels, np.int32)
        tx = np.ascontiguousarray(text, np.float32)
        sz = np.ascontiguousarray(size, np.float32) if size is not None else None
        self.L.orc_set_regions(self.s, _p(lb), C.c_int(len(tx)), _p(tx), _p(sz) if sz is not None else None)
        self.n_regions = len(tx)

    def set_region_planes(self, planes):
        pl = np.ascontiguousarray(planes, np.float32)
        self.L.orc_set_region_planes(self.s, _p(pl))

    def fake_depth(self):
        self.L.orc_fake_depth(self.s)

    def update_scale(self):
        self.L.orc_update_scale(self.s)

    def wmf_detect(self, it):
        self.L.orc_wmf_detect(self.s, C.c_int(it))

    def wmf_fill(self, it):
        self.L.orc_wmf_fill(self.s, C.c_int(it))

    def ransac_regions(self):
        ratio = np.zeros(self.n_regions, np.float32)
        self.L.orc_ransac_regions(self.s, _p(ratio))
        planes = np.ctypeslib.as_array(self.L.orc_region_planes(self.s), shape=(self.n_regions, 4)).copy()
        return planes, ratio


def rng4(seed, pixel, stream, step):
    u = np.empty(4, np.float32)
    lib().orc_rng4(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(stream), C.c_uint32(step), _p(u))
    return u


def expf(x):
    return float(lib().orc_expf(C.c_float(x)))


def ransac_points(pts, region_size, seed=0, region=0, flags=0):
    pts = np.ascontiguousarray(pts, np.float32)
    plane = np.empty(4, np.float32)
    best = lib().orc_ransac_points(_p(pts), len(pts), C.c_float(region_size), C.c_uint64(seed), C.c_uint32(region), C.c_uint32(flags), _p(plane))
    return plane, int(best)


def pow_third(x):
    """x^(1.0f / 3.0f) as rgb2CIELab's pow() is restated (oracle/tsar_oracle_slic.c): correctly rounded to fp32"""
    return float(lib().orc_pow_third(C.c_float(x)))


def pow_third_check():
    """(evaluations, mismatches against powl rounded to fp32, evaluations where this host's powf differs from that) over every
    argument an 8-bit colour can hand to rgb2CIELab's pow()"""
    out = np.zeros(3, np.int64)
    lib().orc_pow_third_check(_p(out))
    return tuple(int(v) for v in out)


# the reference's spixel_info (gSLICr_spixel_info.h:11-17) = the oracle's `spixel`
SPIXEL_DTYPE = np.dtype([("center", np.float32, 2), ("color", np.float32, 4), ("id", np.int32), ("n", np.int32)])


def slic_convert(bgra, color_space=0):
    px = np.ascontiguousarray(bgra, np.uint8).reshape(-1, 4)
    out = np.zeros((px.shape[0], 4), np.float32)
    lib().orc_slic_convert(_p(px), _p(out), px.shape[0], color_space)
    return out


def slic_init_centers(lab, mw, mh, S):
    lab = np.ascontiguousarray(lab, np.float32)
    h, w = lab.shape[:2]
    out = np.zeros(mw * mh, SPIXEL_DTYPE)
    lib().orc_slic_init_centers(_p(lab), _p(out), w, h, mw, mh, S)
    return out


def slic_distance(pix, x, y, centre, weight, norm_xy):
    pix = np.ascontiguousarray(pix, np.float32)
    c = np.ascontiguousarray(centre)
    return float(lib().orc_slic_distance(_p(pix), int(x), int(y), _p(c), C.c_float(weight), C.c_float(norm_xy)))


def slic_find_association(lab, centres, mw, mh, S, weight, labels_before=None):
    lab = np.ascontiguousarray(lab, np.float32)
    h, w = lab.shape[:2]
    labels = np.zeros((h, w), np.int32) if labels_before is None else np.ascontiguousarray(labels_before, np.int32).copy()
    lib().orc_slic_find_association(_p(lab), _p(np.ascontiguousarray(centres)), _p(labels), w, h, mw, mh, S, C.c_float(weight))
    return labels


def slic_partials(lab, labels, S):
    """what Update_Cluster_Center_device leaves in accum_map: blocks_per_spixel(S) records per superpixel"""
    lab = np.ascontiguousarray(lab, np.float32)
    h, w = lab.shape[:2]
    mw, mh = w // S, h // S
    nblk = lib().orc_slic_blocks_per_spixel(S)
    out = np.zeros((mw * mh, nblk), SPIXEL_DTYPE)
    lib().orc_slic_partials(_p(lab), _p(np.ascontiguousarray(labels, np.int32)), _p(out), w, h, mw, mh, S)
    return out


def slic_finalize(accum):
    accum = np.ascontiguousarray(accum)
    out = np.zeros(accum.shape[0], SPIXEL_DTYPE)
    lib().orc_slic_finalize(_p(accum), _p(out), accum.shape[0], accum.shape[1])
    return out


def slic_update_centers(lab, labels, S):
    lab = np.ascontiguousarray(lab, np.float32)
    h, w = lab.shape[:2]
    out = np.zeros((w // S) * (h // S), SPIXEL_DTYPE)
    lib().orc_slic_update_centers(_p(lab), _p(np.ascontiguousarray(labels, np.int32)), _p(out), w, h, w // S, h // S, S)
    out["id"] = np.arange(out.size)
    return out


def slic_connectivity(labels):
    labels = np.ascontiguousarray(labels, np.int32)
    out = np.zeros_like(labels)
    lib().orc_slic_connectivity(_p(labels), _p(out), labels.shape[1], labels.shape[0])
    return out


def slic_from_lab(lab, S=20, iters=5, weight=5.0, connectivity=0):
    lab = np.ascontiguousarray(lab, np.float32)
    h, w = lab.shape[:2]
    labels = np.zeros((h, w), np.int32)
    centres = np.zeros((w // S) * (h // S), SPIXEL_DTYPE)
    lib().orc_slic_from_lab(_p(lab), w, h, S, iters, C.c_float(weight), connectivity, _p(labels), _p(centres))
    return labels, centres


def rgb2lab(bgra):
    px = np.ascontiguousarray(bgra, np.uint8)
    out = np.empty(4, np.float32)
    lib().orc_rgb2lab(_p(px), _p(out))
    return out


def slic(bgra, spixel_size=20, iters=5, weight=5.0, connectivity=0, color_space=0, want_centers=False):
    img = np.ascontiguousarray(bgra, np.uint8)
    h, w = img.shape[:2]
    labels = np.empty((h, w), np.int32)
    lab = np.empty((h, w, 4), np.float32)
    mw, mh = w // spixel_size, h // spixel_size
    centers = np.empty((mw * mh, 8), np.float32)
    lib().orc_slic(_p(img), w, h, spixel_size, iters, C.c_float(weight), connectivity, color_space, _p(labels), _p(lab), _p(centers))
    return (labels, lab, centers) if want_centers else labels


# ---- weak-texture detection (oracle/tsar_oracle_texture.c) ---------------------------------------------
def weak_texture(gray_u8, connect="true", close_lines=False):
    """CPU restatement of texture(): -> dict(labels4, labels, text, size, cenx, ceny, count, edge).
    close_lines: run the deterministic Hough boundary closing (orc_hough_close) between the first labelling and the
    border fix, where the reference calls cv::HoughLinesP (main.cpp:385-435)."""
    L = lib()
    g = np.ascontiguousarray(gray_u8, np.uint8)
    h, w = g.shape
    w2, h2 = w // 2, h // 2
    w4, h4 = w2 // 2, h2 // 2
    d2 = np.empty((h2, w2), np.uint8)
    d4 = np.empty((h4, w4), np.uint8)
    L.orc_pyrdown(_p(g), w, h, _p(d2))
    L.orc_pyrdown(_p(d2), w2, h2, _p(d4))
    edge = np.empty((h4, w4), np.uint8)
    L.orc_roberts_threshold(_p(d4), w4, h4, _p(edge))
    fn = L.orc_connect_true if connect == "true" else L.orc_connect_literal
    fn.restype = C.c_int
    segments = 0
    if close_lines:
        lab0 = np.empty((h4, w4), np.int32)
        n0 = L.orc_connect_true(_p(edge), w4, h4, _p(lab0), None, 0)
        L.orc_hough_close.restype = C.c_int
        segments = L.orc_hough_close(_p(edge), _p(lab0), n0, w4, h4)
    L.orc_border_fix(_p(edge), w4, h4)
    lab4 = np.empty((h4, w4), np.int32)
    n = fn(_p(edge), w4, h4, _p(lab4), None, 0)
    text = np.empty(n, np.float32); size = np.empty(n, np.float32)
    cenx = np.empty(n, np.int32); ceny = np.empty(n, np.int32); count = np.empty(n, np.int32)
    L.orc_region_stats(_p(lab4), w4, h4, n, _p(text), _p(size), _p(cenx), _p(ceny), _p(count))
    labels = np.empty((h, w), np.int32)
    L.orc_upsample_labels(_p(lab4), w4, h4, w, h, _p(labels))
    return dict(labels4=lab4, labels=labels, text=text, size=size, cenx=cenx, ceny=ceny, count=count, edge=edge, down4=d4, segments=segments)


# ---- fusion (oracle/tsar_oracle_fusion.c) ----------------------------------------------------------------
class FusCam(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("R", C.c_float * 9), ("t", C.c_float * 3)]


def fuse(depths, normals, grays, K, R, t, pairs, num_consistent=1, reproj_error=2.0, depth_diff=0.01, angle_deg=15.0, used_list=1):
    L = lib()
    n = len(depths)
    h, w = depths[0].shape
    d = [np.ascontiguousarray(a, np.float32) for a in depths]
    nr = [np.ascontiguousarray(a, np.float32) for a in normals]
    g = [np.ascontiguousarray(a, np.float32) for a in grays]
    mk = lambda seq: (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p) for a in seq])
    cams = (FusCam * n)()
    K = np.asarray(K, np.float32).reshape(n, 9); R = np.asarray(R, np.float32).reshape(n, 9); t = np.asarray(t, np.float32).reshape(n, 3)
    for i in range(n):
        cams[i].K[:] = K[i].tolist(); cams[i].R[:] = R[i].tolist(); cams[i].t[:] = t[i].tolist()
    lists = [list(pairs[v]) for v in range(n)]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(x) for x in lists])
    idx = np.asarray([s for x in lists for s in x] or [0], np.int32)
    cap = n * h * w
    out = np.empty((cap, 9), np.float32)
    cos_angle = np.float32(np.cos(np.float64(np.float32(angle_deg)) * 3.14159265358979323846 / 180.0))
    L.orc_fuse.restype = C.c_int
    cnt = L.orc_fuse(n, w, h, cams, mk(d), mk(nr), mk(g), _p(off), _p(idx), num_consistent, C.c_float(reproj_error), C.c_float(depth_diff),
                     C.c_float(cos_angle), used_list, _p(out), cap)
    return out[:cnt].copy()


_rcp_table_cache = {}


def rcp_table_from_device(matcher):
    """v_rcp_f32 of every fp32 in [1, 2), read from the GPU through the C ABI's self-test entry (fast form: u = X * rcp(Z) with
    X = 1).  2^23 results, 32 MB; cached per process."""
    if "t" not in _rcp_table_cache:
        z = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
        one = np.ones_like(z)
        u, _ = matcher.selftest_divide(one, one, z, mode=2)
        _rcp_table_cache["t"] = u
    return _rcp_table_cache["t"]
