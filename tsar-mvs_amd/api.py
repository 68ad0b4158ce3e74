"""ctypes binding of libtsar_hip.so (include/tsar.h) and a thin `Matcher` object mirroring the reference's
operator sequence (`firstcuda` / `sliccuda` / `fakecuda` / `fillcuda`, reference gipuma.h:2-6, called from
runGipuma, reference main.cpp:1493-1783).

There is no CPU fallback: if the HIP library is missing or no GPU is present the calls raise.
Buffers may be numpy arrays (host) or torch CUDA tensors (device, zero-copy via data_ptr()).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSAR_LIB") or os.path.join(_HERE, "libtsar_hip.so")     # TSAR_LIB: A/B builds only

TSAR_OK = 0
TSAR_ERR_INVALID, TSAR_ERR_HIP, TSAR_ERR_STATE, TSAR_ERR_NOMEM = -1, -2, -3, -4
MEM_HOST, MEM_DEVICE = 0, 1
COMB_ALL, COMB_BEST_N, COMB_ANGLE, COMB_GOOD = 0, 1, 2, 3
FLAG_FIX_DOWN_FAR_SEED, FLAG_FIX_RIGHT_FAR_CMP, FLAG_STRICT_DIV = 1, 2, 4
FLAG_NO_LINE_CLOSING = 16
FLAG_TEX_FILTER_8BIT = 32
FLAG_FIX_INIT_RADIUS = 64
MAXCOST = 2.0
MAX_VIEWS = 64


class TsarError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tsar error {code}: {msg}")
        self.code = code


class Camera(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("R", C.c_float * 9), ("t", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("box_hsize", C.c_int32), ("box_vsize", C.c_int32), ("n_best", C.c_int32), ("cost_comb", C.c_int32),
                ("depth_min", C.c_float), ("depth_max", C.c_float), ("cam_scale", C.c_float), ("flags", C.c_uint32),
                ("seed", C.c_uint64)]


class SlicSettings(C.Structure):
    _fields_ = [("spixel_size", C.c_int32), ("no_iters", C.c_int32), ("coh_weight", C.c_float),
                ("do_enforce_connectivity", C.c_int32), ("color_space", C.c_int32)]


class FusionParams(C.Structure):
    _fields_ = [("num_consistent", C.c_int32), ("reproj_error", C.c_float), ("depth_diff", C.c_float), ("angle_deg", C.c_float),
                ("used_list", C.c_int32)]


class KernelTiming(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int32), ("total_ms", C.c_float)]


# the reference's spixel_info (gSLICr_spixel_info.h:11-17), the record of tsar_selftest_slic_stage's centre arrays
SPIXEL_DTYPE = np.dtype([("center", np.float32, 2), ("color", np.float32, 4), ("id", np.int32), ("n", np.int32)])

# every symbol include/tsar.h declares (tests check that the library exports exactly these)
ABI_SYMBOLS = [
    "tsar_create", "tsar_destroy", "tsar_last_error", "tsar_version", "tsar_get_stream", "tsar_synchronize",
    "tsar_default_params", "tsar_set_params", "tsar_set_views", "tsar_set_views_u8", "tsar_set_view_subset",
    "tsar_pm_init", "tsar_pm_iterate", "tsar_pm_iterate_final", "tsar_pm_sweep", "tsar_set_sweep_counter", "tsar_pm_cost_planes", "tsar_set_plane", "tsar_get_plane",
    "tsar_load_planes", "tsar_compute_disp", "tsar_compute_disp_final", "tsar_depth_to_plane", "tsar_get_result",
    "tsar_set_reliable_mask", "tsar_get_reliable_mask", "tsar_lrdiff", "tsar_getview", "tsar_wmf", "tsar_set_regions", "tsar_detect_weak_texture", "tsar_ransac_regions",
    "tsar_set_region_planes", "tsar_fake_depth", "tsar_fill_textureless",
    "tsar_default_slic_settings", "tsar_slic", "tsar_default_fusion_params", "tsar_fuse", "tsar_fuse_ctx",
    "tsar_host_alloc", "tsar_host_free", "tsar_device_alloc", "tsar_device_free", "tsar_device_write", "tsar_peer_copy", "tsar_enable_kernel_timing", "tsar_reset_kernel_timing", "tsar_get_kernel_timing",
    "tsar_selftest_divide", "tsar_selftest_divide_random", "tsar_selftest_sqrt", "tsar_selftest_sweep_census", "tsar_selftest_sweep_repeat", "tsar_selftest_slic_stage",
]

_lib = None


def load_library(path: str = LIB_PATH):
    """Load libtsar_hip.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C tsar-mvs_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    L.tsar_last_error.restype = C.c_char_p
    L.tsar_last_error.argtypes = [C.c_void_p]
    L.tsar_version.restype = C.c_char_p
    L.tsar_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.tsar_destroy.argtypes = [C.c_void_p]
    L.tsar_default_params.restype = None
    L.tsar_default_params.argtypes = [C.POINTER(Params)]
    L.tsar_default_slic_settings.restype = None
    L.tsar_default_slic_settings.argtypes = [C.POINTER(SlicSettings)]
    L.tsar_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    L.tsar_set_views.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(Camera)]
    L.tsar_set_views_u8.argtypes = L.tsar_set_views.argtypes
    L.tsar_set_view_subset.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.tsar_pm_init.argtypes = [C.c_void_p]
    L.tsar_pm_iterate.argtypes = [C.c_void_p, C.c_int]
    L.tsar_pm_iterate_final.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.tsar_pm_sweep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.tsar_set_sweep_counter.argtypes = [C.c_void_p, C.c_int]
    L.tsar_pm_cost_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsar_set_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_get_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_load_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_compute_disp.argtypes = [C.c_void_p]
    L.tsar_compute_disp_final.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_depth_to_plane.argtypes = [C.c_void_p]
    L.tsar_get_result.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_set_reliable_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_get_reliable_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_host_alloc.restype = C.c_void_p
    L.tsar_host_alloc.argtypes = [C.c_size_t]
    L.tsar_host_free.restype = None
    L.tsar_host_free.argtypes = [C.c_void_p]
    L.tsar_device_alloc.restype = C.c_void_p
    L.tsar_device_alloc.argtypes = [C.c_int, C.c_size_t]
    L.tsar_device_free.restype = None
    L.tsar_device_free.argtypes = [C.c_int, C.c_void_p]
    L.tsar_device_write.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.tsar_peer_copy.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.tsar_lrdiff.argtypes = [C.c_void_p]
    L.tsar_getview.argtypes = [C.c_void_p]
    L.tsar_wmf.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.tsar_set_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_detect_weak_texture.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_ransac_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsar_set_region_planes.argtypes = [C.c_void_p, C.c_void_p]
    L.tsar_fake_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_fill_textureless.argtypes = [C.c_void_p]
    L.tsar_slic.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(SlicSettings), C.c_void_p, C.c_int]
    L.tsar_default_fusion_params.restype = None
    L.tsar_default_fusion_params.argtypes = [C.POINTER(FusionParams)]
    L.tsar_fuse.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Camera), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int,
                            C.c_void_p, C.c_void_p, C.POINTER(FusionParams), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.tsar_fuse_ctx.argtypes = [C.c_void_p] + list(L.tsar_fuse.argtypes[1:])
    L.tsar_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.tsar_synchronize.argtypes = [C.c_void_p]
    L.tsar_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    L.tsar_reset_kernel_timing.argtypes = [C.c_void_p]
    L.tsar_get_kernel_timing.argtypes = [C.c_void_p, C.POINTER(KernelTiming), C.c_int, C.POINTER(C.c_int)]
    L.tsar_selftest_divide.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    L.tsar_selftest_sweep_census.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.tsar_selftest_sweep_repeat.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    L.tsar_selftest_sqrt.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_uint64)]
    L.tsar_selftest_divide_random.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.tsar_selftest_slic_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SlicSettings), C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def default_params(**kw) -> Params:
    p = Params()
    load_library().tsar_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def _ptr(a):
    """(pointer, mem kind) of a numpy array or torch tensor; None -> (NULL, host)."""
    if a is None:
        return None, MEM_HOST
    if _is_torch(a):
        assert a.is_contiguous(), "tensor must be contiguous"
        if a.is_cuda:
            # include/tsar.h: device buffers must be complete before the call — the context's stream is not ordered
            # against torch's.  Whatever produced (or still reads) this tensor ran on torch's current stream.
            import torch
            torch.cuda.current_stream(a.device).synchronize()
        return C.c_void_p(a.data_ptr()), (MEM_DEVICE if a.is_cuda else MEM_HOST)
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p), MEM_HOST


class Matcher:
    """One context = one GPU.  Mirrors the order in which runGipuma drives the GPU operators."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        self._ctx = C.c_void_p()
        rc = self.L.tsar_create(device, C.byref(self._ctx))
        if rc != TSAR_OK:
            raise TsarError(rc, "tsar_create failed (is a HIP device visible?)")
        self.w = self.h = self.n_views = 0
        self._keep = []

    def close(self):
        if self._ctx:
            self.L.tsar_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != TSAR_OK:
            raise TsarError(rc, self.L.tsar_last_error(self._ctx).decode())

    # ---- inputs ----
    def set_params(self, params: Params):
        self._chk(self.L.tsar_set_params(self._ctx, C.byref(params)))
        self.params = params

    def set_views(self, images, K, R, t, u8: bool = False):
        """images: list of [h, w] float32 arrays/tensors (view 0 = reference); K, R, t: per-view
        intrinsics and world->camera extrinsics as in cams/%08d_cam.txt.  u8: the images are uint8 (the 8-bit decode itself)
        and go through tsar_set_views_u8, which widens them on the device."""
        n = len(images)
        first = images[0]
        h, w = int(first.shape[0]), int(first.shape[1])
        ptrs = (C.c_void_p * n)()
        kinds = set()
        for i, im in enumerate(images):
            assert tuple(im.shape) == (h, w)
            if not _is_torch(im):
                im = np.ascontiguousarray(im, dtype=np.uint8 if u8 else np.float32)
                self._keep.append(im)
            elif u8:
                assert str(im.dtype) == "torch.uint8" and im.is_contiguous()
            p, kind = _ptr(im)
            ptrs[i] = p
            kinds.add(kind)
        assert len(kinds) == 1, "all images must live in the same memory space"
        cams = (Camera * n)()
        K = np.asarray(K, np.float32).reshape(n, 9)
        R = np.asarray(R, np.float32).reshape(n, 9)
        t = np.asarray(t, np.float32).reshape(n, 3)
        for i in range(n):
            cams[i].K[:] = K[i].tolist()
            cams[i].R[:] = R[i].tolist()
            cams[i].t[:] = t[i].tolist()
        self._chk((self.L.tsar_set_views_u8 if u8 else self.L.tsar_set_views)(self._ctx, n, w, h, ptrs, kinds.pop(), cams))
        self._keep.clear()
        self.w, self.h, self.n_views = w, h, n

    def set_view_subset(self, idx):
        a = np.ascontiguousarray(idx, np.int32)
        self._chk(self.L.tsar_set_view_subset(self._ctx, len(a), a.ctypes.data_as(C.c_void_p)))

    # ---- PatchMatch ----
    def pm_init(self):
        self._chk(self.L.tsar_pm_init(self._ctx))

    def pm_iterate(self, iters: int):
        self._chk(self.L.tsar_pm_iterate(self._ctx, iters))

    def pm_iterate_final(self, iters: int, text):
        """the kernels' `final == true` mode; text [h, w] = lines->text (-1: pixel is left untouched)"""
        t = text if _is_torch(text) else np.ascontiguousarray(text, np.float32)
        p, kind = _ptr(t)
        self._chk(self.L.tsar_pm_iterate_final(self._ctx, iters, p, kind))

    def pm_sweep(self, colour: int, do_prop: bool = True, do_refine: bool = True):
        self._chk(self.L.tsar_pm_sweep(self._ctx, colour, int(do_prop), int(do_refine)))

    def set_sweep_counter(self, n: int):
        self._chk(self.L.tsar_set_sweep_counter(self._ctx, n))

    def pm_cost_planes(self, planes):
        planes = np.ascontiguousarray(planes, np.float32)
        cost = np.empty((self.h, self.w), np.float32)
        bv = np.empty((self.h, self.w), np.int32)
        rt = np.empty((self.h, self.w), np.float32)
        self._chk(self.L.tsar_pm_cost_planes(self._ctx, _ptr(planes)[0], MEM_HOST, _ptr(cost)[0], _ptr(bv)[0], _ptr(rt)[0]))
        return cost, bv, rt

    def set_plane(self, planes, cost):
        planes = np.ascontiguousarray(planes, np.float32)
        cost = np.ascontiguousarray(cost, np.float32)
        self._chk(self.L.tsar_set_plane(self._ctx, _ptr(planes)[0], _ptr(cost)[0], MEM_HOST))

    def get_plane(self):
        planes = np.empty((self.h, self.w, 4), np.float32)
        cost = np.empty((self.h, self.w), np.float32)
        bv = np.empty((self.h, self.w), np.int32)
        rt = np.empty((self.h, self.w), np.float32)
        self._chk(self.L.tsar_get_plane(self._ctx, _ptr(planes)[0], _ptr(cost)[0], _ptr(bv)[0], _ptr(rt)[0], MEM_HOST))
        return planes, cost, bv, rt

    # ---- plane <-> depth ----
    def load_planes(self, depth, normal_world):
        d, kd = _ptr(depth if _is_torch(depth) else np.ascontiguousarray(depth, np.float32))
        depth_keep = depth if _is_torch(depth) else np.ascontiguousarray(depth, np.float32)
        normal_keep = normal_world if _is_torch(normal_world) else np.ascontiguousarray(normal_world, np.float32)
        d, kd = _ptr(depth_keep)
        n, kn = _ptr(normal_keep)
        assert kd == kn
        self._chk(self.L.tsar_load_planes(self._ctx, d, n, kd))

    def compute_disp(self):
        self._chk(self.L.tsar_compute_disp(self._ctx))

    def compute_disp_final(self, resize_planes, text):
        r = np.ascontiguousarray(resize_planes, np.float32)
        t = np.ascontiguousarray(text, np.float32)
        self._chk(self.L.tsar_compute_disp_final(self._ctx, _ptr(r)[0], _ptr(t)[0], MEM_HOST))

    def depth_to_plane(self):
        self._chk(self.L.tsar_depth_to_plane(self._ctx))

    def get_result(self, want=("depth", "normal", "cost", "confid"), pinned=False, out=None):
        """pinned=True: the result arrays are page-locked (tsar_host_alloc), so the D2H copies run at PCIe rate.
        out: dict of caller-owned arrays to fill instead (e.g. page-locked ones allocated once and reused per view)."""
        shapes = {"depth": (self.h, self.w), "normal": (self.h, self.w, 3), "cost": (self.h, self.w), "confid": (self.h, self.w)}
        empty = pinned_empty if pinned else np.empty
        res = {}
        for k in ("depth", "normal", "cost", "confid"):
            if out is not None and k in out:
                a = out[k]
                assert a.dtype == np.float32 and tuple(a.shape) == shapes[k] and a.flags["C_CONTIGUOUS"]
                res[k] = a
            elif out is None and k in want:
                res[k] = empty(shapes[k], np.float32)
        self._chk(self.L.tsar_get_result(self._ctx, _ptr(res.get("depth"))[0], _ptr(res.get("normal"))[0], _ptr(res.get("cost"))[0],
                                         _ptr(res.get("confid"))[0], MEM_HOST))
        return res

    def get_result_device(self, depth=None, normal=None, cost=None, confid=None):
        """Write results into caller-provided torch CUDA tensors."""
        self._chk(self.L.tsar_get_result(self._ctx, _ptr(depth)[0], _ptr(normal)[0], _ptr(cost)[0], _ptr(confid)[0], MEM_DEVICE))

    # ---- TSAR refinement ----
    def set_reliable_mask(self, scale):
        s = np.ascontiguousarray(scale, np.float32)
        self._chk(self.L.tsar_set_reliable_mask(self._ctx, _ptr(s)[0], MEM_HOST))

    def get_reliable_mask(self):
        s = np.empty((self.h, self.w), np.float32)
        self._chk(self.L.tsar_get_reliable_mask(self._ctx, _ptr(s)[0], MEM_HOST))
        return s

    def lrdiff(self):
        self._chk(self.L.tsar_lrdiff(self._ctx))

    # ---- self-tests ----
    def selftest_divide(self, X, Y, Z, ieee: bool = False, mode: int | None = None):
        """(u, v) = (X / Z, Y / Z) as strict mode's tap loops compute them (mode 0); as the device's IEEE division does (ieee /
        mode 1); as the fast mode computes them, X * v_rcp_f32(Z) (mode 2)."""
        X, Y, Z = (np.ascontiguousarray(a, dtype=np.float32) for a in (X, Y, Z))
        assert X.shape == Y.shape == Z.shape
        u, v = np.empty_like(X), np.empty_like(X)
        self._chk(self.L.tsar_selftest_divide(self._ctx, _ptr(X)[0], _ptr(Y)[0], _ptr(Z)[0], X.size, _ptr(u)[0], _ptr(v)[0], int(ieee) if mode is None else mode))
        return u, v

    def selftest_sqrt(self, mode: int, seed: int = 0) -> int:
        """mismatches of the cost tail's square root (sqrt_rsq_exact) against sqrtf on the device; mode 0: all 2^24 mantissa / parity cases"""
        bad = C.c_uint64(0)
        self._chk(self.L.tsar_selftest_sqrt(self._ctx, mode, seed, C.byref(bad)))
        return bad.value

    def selftest_sweep_census(self, colour: int) -> dict:
        out = (C.c_uint64 * 8)()
        self._chk(self.L.tsar_selftest_sweep_census(self._ctx, colour, out))
        keys = ("waves", "arms_any_lane", "max_lane_survivors", "max_lane_survivors_nodup", "lane_survivors", "lane_survivors_nodup", "pixels", "arms_present")
        return dict(zip(keys, [int(v) for v in out]))

    def selftest_divide_random(self, log2_triples: int, seed: int, mode: int, guarded: bool = True):
        """(quotients differing from IEEE division, triples outside the operand guard) over 2^log2_triples device-generated triples."""
        bad, out = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.L.tsar_selftest_divide_random(self._ctx, log2_triples, seed, mode, int(guarded), C.byref(bad), C.byref(out)))
        return bad.value, out.value

    # one stage of tsar_slic on host arrays (include/tsar.h tsar_selftest_slic_stage); centres are SPIXEL_DTYPE records
    def _slic_stage(self, stage, w, h, mw, mh, settings, in0, in1, inout):
        p1 = _ptr(in1)[0] if in1 is not None else None
        self._chk(self.L.tsar_selftest_slic_stage(self._ctx, stage, w, h, mw, mh, C.byref(settings), _ptr(in0)[0], p1, _ptr(inout)[0]))
        return inout

    def slic_convert(self, bgra, color_space=0):
        img = np.ascontiguousarray(bgra, np.uint8).reshape(-1, 4)
        return self._slic_stage(0, img.shape[0], 1, 0, 0, SlicSettings(20, 0, 5.0, 0, color_space), img, None, np.zeros((img.shape[0], 4), np.float32))

    def slic_init_centers(self, lab, mw, mh, S):
        lab = np.ascontiguousarray(lab, np.float32)
        h, w = lab.shape[:2]
        return self._slic_stage(1, w, h, mw, mh, SlicSettings(S, 0, 5.0, 0, 0), lab, None, np.zeros(mw * mh, SPIXEL_DTYPE))

    def slic_find_association(self, lab, centres, mw, mh, S, weight, labels_before=None):
        lab = np.ascontiguousarray(lab, np.float32)
        h, w = lab.shape[:2]
        labels = np.zeros((h, w), np.int32) if labels_before is None else np.ascontiguousarray(labels_before, np.int32).copy()
        return self._slic_stage(2, w, h, mw, mh, SlicSettings(S, 0, weight, 0, 0), lab, np.ascontiguousarray(centres), labels)

    def slic_update_centers(self, lab, labels, S):
        lab = np.ascontiguousarray(lab, np.float32)
        h, w = lab.shape[:2]
        return self._slic_stage(3, w, h, w // S, h // S, SlicSettings(S, 0, 5.0, 0, 0), lab, np.ascontiguousarray(labels, np.int32), np.zeros((w // S) * (h // S), SPIXEL_DTYPE))

    def slic_connectivity(self, labels):
        labels = np.ascontiguousarray(labels, np.int32)
        h, w = labels.shape
        return self._slic_stage(4, w, h, 0, 0, SlicSettings(20, 0, 5.0, 0, 0), labels, None, np.zeros((h, w), np.int32))

    def getview(self):
        self._chk(self.L.tsar_getview(self._ctx))

    def wmf(self, iters: int, final_pass: bool):
        self._chk(self.L.tsar_wmf(self._ctx, iters, int(final_pass)))

    def set_regions(self, labels, region_text, region_size=None):
        tx = np.ascontiguousarray(region_text, np.float32)
        sz = np.ascontiguousarray(region_size, np.float32) if region_size is not None else None
        if _is_torch(labels) and labels.is_cuda:
            # labels on the device; the (small) region tables travel to the device the same way
            import torch
            assert labels.dtype == torch.int32
            txd = torch.from_numpy(tx).to(labels.device)
            szd = torch.from_numpy(sz).to(labels.device) if sz is not None else None
            self._chk(self.L.tsar_set_regions(self._ctx, _ptr(labels)[0], len(tx), _ptr(txd)[0], _ptr(szd)[0], MEM_DEVICE))
        else:
            lb = np.ascontiguousarray(labels, np.int32)
            self._chk(self.L.tsar_set_regions(self._ctx, _ptr(lb)[0], len(tx), _ptr(tx)[0], _ptr(sz)[0], MEM_HOST))
        self.n_regions = len(tx)

    def detect_weak_texture(self, cap: int = 1 << 16, want_labels: bool = True):
        """-> (labels [h, w] int32, region_text [n], region_size [n]); also installs them as the regions.
        want_labels=False: labels_out = NULL, the call tsar_gipuma makes (the labels stay on the device for the next operators;
        no [h, w] int32 D2H copy) -> labels is None"""
        labels = np.empty((self.h, self.w), np.int32) if want_labels else None
        text = np.empty(cap, np.float32)
        size = np.empty(cap, np.float32)
        n = C.c_int(0)
        self._chk(self.L.tsar_detect_weak_texture(self._ctx, _ptr(labels)[0] if want_labels else None, MEM_HOST, C.byref(n), _ptr(text)[0], _ptr(size)[0], cap))
        self.n_regions = n.value
        k = min(n.value, cap)
        return labels, text[:k].copy(), size[:k].copy()

    def ransac_regions(self):
        planes = np.empty((self.n_regions, 4), np.float32)
        ratio = np.empty((self.n_regions,), np.float32)
        self._chk(self.L.tsar_ransac_regions(self._ctx, _ptr(planes)[0], _ptr(ratio)[0]))
        return planes, ratio

    def set_region_planes(self, planes):
        pl = np.ascontiguousarray(planes, np.float32)
        self._chk(self.L.tsar_set_region_planes(self._ctx, _ptr(pl)[0]))

    def fake_depth(self):
        out = np.empty((self.h, self.w), np.float32)
        self._chk(self.L.tsar_fake_depth(self._ctx, _ptr(out)[0], MEM_HOST))
        return out

    def fill_textureless(self):
        self._chk(self.L.tsar_fill_textureless(self._ctx))

    def slic(self, bgra, settings: SlicSettings | None = None):
        img = np.ascontiguousarray(bgra, np.uint8)
        h, w = img.shape[:2]
        if settings is None:
            settings = SlicSettings()
            self.L.tsar_default_slic_settings(C.byref(settings))
        labels = np.empty((h, w), np.int32)
        self._chk(self.L.tsar_slic(self._ctx, _ptr(img)[0], w, h, C.byref(settings), _ptr(labels)[0], MEM_HOST))
        return labels

    # ---- measurement ----
    @property
    def stream(self) -> int:
        s = C.c_void_p()
        self._chk(self.L.tsar_get_stream(self._ctx, C.byref(s)))
        return s.value or 0

    def synchronize(self):
        self._chk(self.L.tsar_synchronize(self._ctx))

    def enable_kernel_timing(self, on: bool = True):
        self._chk(self.L.tsar_enable_kernel_timing(self._ctx, int(on)))

    def reset_kernel_timing(self):
        self._chk(self.L.tsar_reset_kernel_timing(self._ctx))

    def kernel_timing(self):
        buf = (KernelTiming * 32)()
        n = C.c_int(0)
        self._chk(self.L.tsar_get_kernel_timing(self._ctx, buf, 32, C.byref(n)))
        return {buf[i].name.decode(): (buf[i].launches, buf[i].total_ms) for i in range(min(n.value, 32))}


def matcher_from_scene(scene, box=11, n_best=1, cost_comb=COMB_BEST_N, flags=0, seed=2024, device=0, subset=None, box_v=None) -> Matcher:
    """Convenience used by tests/bench: a Matcher loaded with a tsar_mvs_amd.synth.Scene."""
    m = Matcher(device)
    m.set_params(default_params(box_hsize=box, box_vsize=box if box_v is None else box_v, n_best=n_best, cost_comb=cost_comb, depth_min=scene.depth_min,
                                depth_max=scene.depth_max, flags=flags, seed=seed))
    m.set_views(scene.images, scene.K, scene.R, scene.t)
    if subset is not None:
        m.set_view_subset(subset)
    return m


def fuse(depths, normals, grays, K, R, t, pairs, params: FusionParams | None = None, cap: int | None = None, device: int = 0, matcher=None):
    """Fuse per-view depth [h, w] / world-normal [h, w, 3] maps into a point cloud (tsar_fuse; with `matcher`, tsar_fuse_ctx on that
    context: its stream, temporaries from its scratch arena).
    pairs: {view: [source views]} or list of lists.  Returns an [n, 9] float32 array:
    x y z, nx ny nz, gray, number of agreeing views, reference view."""
    L = load_library()
    n = len(depths)
    h, w = int(depths[0].shape[0]), int(depths[0].shape[1])
    if params is None:
        params = FusionParams()
        L.tsar_default_fusion_params(C.byref(params))
    keep, kinds = [], set()

    def ptrs(seq, shape):
        arr = (C.c_void_p * n)()
        for i, a in enumerate(seq):
            if not _is_torch(a):
                a = np.ascontiguousarray(a, np.float32)
                keep.append(a)
            assert tuple(a.shape) == shape
            p, kind = _ptr(a)
            arr[i] = p
            kinds.add(kind)
        return arr
    pd, pn, pg = ptrs(depths, (h, w)), ptrs(normals, (h, w, 3)), ptrs(grays, (h, w))
    assert len(kinds) == 1
    cams = (Camera * n)()
    K = np.asarray(K, np.float32).reshape(n, 9); R = np.asarray(R, np.float32).reshape(n, 9); t = np.asarray(t, np.float32).reshape(n, 3)
    for i in range(n):
        cams[i].K[:] = K[i].tolist(); cams[i].R[:] = R[i].tolist(); cams[i].t[:] = t[i].tolist()
    lists = [list(pairs[v]) for v in range(n)]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(x) for x in lists])
    idx = np.asarray([s for x in lists for s in x] or [0], np.int32)
    if cap is None:
        cap = n * h * w
    out = np.empty((cap, 9), np.float32)
    cnt = C.c_int64(0)
    tail = (n, w, h, cams, pd, pn, pg, kinds.pop(), off.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), C.byref(params),
            out.ctypes.data_as(C.c_void_p), cap, C.byref(cnt))
    rc = L.tsar_fuse_ctx(matcher._ctx, *tail) if matcher is not None else L.tsar_fuse(device, *tail)
    if rc != TSAR_OK:
        raise TsarError(rc, L.tsar_last_error(matcher._ctx).decode() if matcher is not None else "tsar_fuse failed")
    return out[: min(cnt.value, cap)].copy()


def pinned_empty(shape, dtype=np.float32):
    """numpy array over page-locked host memory from tsar_host_alloc; the memory is released when the last array
    (or view of it) referring to it is collected"""
    import weakref
    L = load_library()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = L.tsar_host_alloc(n)
    if not p:
        raise MemoryError(f"tsar_host_alloc({n}) failed")
    buf = (C.c_char * n).from_address(p)
    weakref.finalize(buf, L.tsar_host_free, p)      # every numpy view keeps `buf` alive through its base chain
    return np.frombuffer(buf, dtype=dt).reshape(shape)
