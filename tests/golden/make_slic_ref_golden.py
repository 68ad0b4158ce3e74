"""Writes tests/golden/slic_ref.npz and tests/golden/mat_ref.npz: outputs of the REFERENCE ITSELF, compiled in the build container.

Runs only where /root/reference exists (the build container):

    make -C oracle ref                      # oracle/_ref/libslic_ref.so, libmat_ref.so, libmat_ref_fma.so  (oracle/Makefile)
    python tests/golden/make_slic_ref_golden.py

What speaks in the fixtures is the reference's own text, compiled from where it lies by g++ with the reference's own
-DCOMPILE_WITHOUT_CUDA switch — no stub header, no stand-in type (oracle/ref_harness/*.cpp only loop its per-pixel functions over
arrays / expand its macros):
  * gSLICr_Lib/engines/gSLICr_seg_engine_shared.h:7-204: rgb2xyz, rgb2CIELab, cvt_img_space_shared, init_cluster_centers_shared,
    compute_slic_distance, find_center_association_shared, finalize_reduction_result_shared, supress_local_lable;
  * config.h:60-240: outer_product, matdivide, matmatsub2, matmul_cu, matvecmul, composed as getHomography_cu does (gipuma.cu:207-224).
Inputs are synthetic (seeded below).  The one stage of gSLICr no host compiler reaches, Update_Cluster_Center_device
(gSLICr_seg_engine_GPU.cu:260-357), is taken from the oracle's restatement (orc_slic_partials) wherever a multi-iteration run needs
it — the arrays it produced are stored as INPUTS of the reference's finalize, so the fixture says exactly which numbers are whose.
The fixtures are data (inputs + the reference's outputs); nothing of the reference's text is stored.
Also writes profiles/r05/slic_reference_pin.json: the census over all 2^24 colours and the label-share figure DESIGN.md quotes.
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
SP = ol.SPIXEL_DTYPE


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def synth_bgra(w, h, seed, cell=(40, 30)):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (127 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)
    img[..., 1] = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
    img[..., 2] = (rng.integers(0, 30, size=(h, w)) + 100 * ((xx // cell[0] + yy // cell[1]) % 2)).astype(np.uint8)
    img[..., 3] = 255
    return img


def lab_newton_cube_root(img):
    """rgb2CIELab with rounds 1-4's stand-in for pow(): a Newton CUBE root in fp32 (kept only to quantify what replacing it bought)"""
    f32 = np.float32
    px = img.reshape(-1, 4)
    _b, _g, _r = (px[:, k].astype(f32) * f32(0.0039216) for k in range(3))
    x = (_r * f32(0.412453) + _g * f32(0.357580)) + _b * f32(0.180423)
    y = (_r * f32(0.212671) + _g * f32(0.715160)) + _b * f32(0.072169)
    z = (_r * f32(0.019334) + _g * f32(0.119193)) + _b * f32(0.950227)

    def f(v):
        c = (np.maximum(v, f32(1e-6)).view(np.uint32) // np.uint32(3) + np.uint32(0x2a5137a0)).view(f32)
        for _ in range(4):
            c = ((c + c) + v / (c * c)) * f32(0.333333343)
        return np.where(v > f32(0.008856), c, (f32(903.3) * v + f32(16)) / f32(116))
    fx, fy, fz = f(x / f32(0.950456)), f(y), f(z / f32(1.088754))
    out = np.zeros((px.shape[0], 4), f32)
    out[:, 0], out[:, 1], out[:, 2] = f32(116) * fy - f32(16), f32(500) * (fx - fy), f32(200) * (fy - fz)
    return out.reshape(img.shape[:-1] + (4,))


class Ref:
    def __init__(self):
        self.L = C.CDLL(os.path.join(REF, "libslic_ref.so"))
        assert self.L.ref_spixel_bytes() == SP.itemsize
        self.L.ref_slic_distance.restype = C.c_float
        self.L.ref_slic_distance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float]

    def cvt(self, bgra, space):
        img = np.ascontiguousarray(bgra, np.uint8)
        h, w = img.shape[:2]
        out = np.zeros((h, w, 4), np.float32)      # the reference leaves .w unwritten for XYZ / CIELAB: stays 0 here
        self.L.ref_cvt_img_space(p(img), p(out), w, h, space)
        return out

    def lab(self, px):
        px = np.ascontiguousarray(px, np.uint8).reshape(-1, 4)
        out = np.zeros((px.shape[0], 4), np.float32)
        self.L.ref_rgb2lab(p(px), p(out), px.shape[0])
        return out

    def xyz(self, px):
        px = np.ascontiguousarray(px, np.uint8).reshape(-1, 4)
        out = np.zeros((px.shape[0], 4), np.float32)
        self.L.ref_rgb2xyz(p(px), p(out), px.shape[0])
        return out

    def init(self, lab, mw, mh, S):
        h, w = lab.shape[:2]
        out = np.zeros(mw * mh, SP)
        self.L.ref_init_cluster_centers(p(lab), p(out), mw, mh, w, h, S)
        return out

    def assoc(self, lab, centres, mw, mh, S, weight, labels_before=None):
        h, w = lab.shape[:2]
        labels = np.zeros((h, w), np.int32) if labels_before is None else labels_before.copy()
        self.L.ref_find_center_association(p(lab), p(centres), p(labels), mw, mh, w, h, S, C.c_float(weight), C.c_float(1.0 / S), C.c_float(15.0 / (1.7321 * 128)))
        return labels

    def finalize(self, accum, centres_before):
        out = centres_before.copy()                 # finalize leaves .id as it was
        mw_mh, nblk = accum.shape
        self.L.ref_finalize_reduction_result(p(np.ascontiguousarray(accum)), p(out), mw_mh, 1, nblk)
        return out

    def supress(self, labels):
        out = np.zeros_like(labels)
        self.L.ref_supress_local_lable(p(labels), p(out), labels.shape[1], labels.shape[0])
        return out

    def distance(self, pix, x, y, centre, weight, norm_xy):
        return self.L.ref_slic_distance(p(pix), int(x), int(y), p(centre), weight, norm_xy, 0.0)


def hybrid_run(ref, lab, S, iters, weight, out, tag):
    """Perform_Segmentation (gSLICr_seg_engine.cpp:30-44) with every stage the reference's own function except the block sums"""
    h, w = lab.shape[:2]
    mw, mh = w // S, h // S
    centres = ref.init(lab, mw, mh, S)
    labels = ref.assoc(lab, centres, mw, mh, S, weight)
    out[tag + "_centres_init"] = centres
    out[tag + "_labels_init"] = labels
    for it in range(iters):
        accum = ol.slic_partials(lab, labels, S)                 # the oracle's restatement of Update_Cluster_Center_device
        centres = ref.finalize(accum, centres)
        labels = ref.assoc(lab, centres, mw, mh, S, weight, labels)
        if it == 0:
            out[tag + "_accum_it0"] = accum
            out[tag + "_centres_it0"] = centres
            out[tag + "_labels_it0"] = labels
    out[tag + "_centres_final"] = centres
    out[tag + "_labels_final"] = labels
    once = ref.supress(labels)
    out[tag + "_labels_connected"] = ref.supress(once)
    return labels


def main():
    ref = Ref()
    rng = np.random.default_rng(20251005)
    out = {}
    # ---- colours: every colour with channels <= 6 (the epsilon branch and its edge), the grey ramp, primaries, then random ----
    dark = np.array([(b, g, r, 0) for b in range(7) for g in range(7) for r in range(7)], np.uint8)
    grey = np.array([(v, v, v, 0) for v in range(256)], np.uint8)
    prim = np.array([(255, 0, 0, 0), (0, 255, 0, 0), (0, 0, 255, 0), (255, 255, 0, 0), (0, 255, 255, 0), (255, 0, 255, 0), (255, 255, 255, 255)], np.uint8)
    rnd = rng.integers(0, 256, size=(65536 - len(dark) - len(grey) - len(prim), 4), dtype=np.uint8)
    colours = np.concatenate([dark, grey, prim, rnd])
    out["colours"] = colours
    out["colours_lab"] = ref.lab(colours)[:, :3].copy()
    out["colours_xyz"] = ref.xyz(colours[:4096])[:, :3].copy()
    # all 2^24 colours: rgb2xyz is exact arithmetic -> a hash pins every one of them
    i = np.arange(1 << 24, dtype=np.uint32)
    allc = np.zeros((1 << 24, 4), np.uint8)
    allc[:, 0] = i & 255
    allc[:, 1] = (i >> 8) & 255
    allc[:, 2] = (i >> 16) & 255
    xyz_all = ref.xyz(allc)[:, :3]
    out["xyz_all_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(xyz_all).tobytes()).digest(), np.uint8)
    lab_all = ref.lab(allc)[:, :3]
    lab_orc = ol.slic_convert(allc, 0)[:, :3]
    d = np.abs(lab_all - lab_orc)
    census = {"colours": 1 << 24, "lab_components_differing": int((d > 0).sum()), "lab_components": int(d.size),
              "lab_colours_differing": int((d > 0).any(1).sum()), "lab_max_abs_diff": float(d.max())}
    n_eval, bad, libm = ol.pow_third_check()
    census.update({"pow_evaluations": n_eval, "pow_third_not_correctly_rounded": bad, "host_powf_not_correctly_rounded": libm})
    out["lab_all_census"] = np.array([census["lab_components_differing"], census["lab_colours_differing"]], np.int64)
    out["lab_all_max_abs_diff"] = np.float32(census["lab_max_abs_diff"])
    del xyz_all, lab_all, lab_orc, d, allc
    # ---- case A: 80 x 60, S = 20 (the reference's setting, main.cpp:608-615), CIELAB, 5 iterations ----
    bgra_a = synth_bgra(80, 60, 7, cell=(24, 18))
    lab_a = ref.cvt(bgra_a, 0)
    out["A_bgra"], out["A_lab"] = bgra_a, lab_a
    hybrid_run(ref, lab_a, 20, 5, 5.0, out, "A")
    # ---- case B: init_cluster_centers_shared's edge branch (shared.h:83-84): a map one column / row larger than the engine's ----
    lab_b = np.ascontiguousarray(lab_a[:50, :70])
    out["B_lab"] = lab_b
    out["B_centres_init"] = ref.init(lab_b, 4, 3, 20)
    # ---- case C: 132 x 100, S = 12 (2 blocks per window line, 6 per superpixel), XYZ, weight 3, 3 iterations ----
    bgra_c = synth_bgra(132, 100, 11, cell=(30, 22))
    lab_c = ref.cvt(bgra_c, 1)
    out["C_bgra"], out["C_lab"] = bgra_c, lab_c
    hybrid_run(ref, lab_c, 12, 3, 3.0, out, "C")
    # ---- supress_local_lable on a noisy label image (5 x 5 majority flips) ----
    yy, xx = np.mgrid[0:60, 0:80]
    noisy = ((yy // 20) * 4 + xx // 20).astype(np.int32)
    flip = rng.random((60, 80)) < 0.35
    noisy[flip] = rng.integers(0, 12, size=int(flip.sum()))
    iso = rng.integers(3, 57, size=(40, 2))
    for y, x in iso:                                    # isolated pixels inside flat areas: the >= 16 rule fires
        noisy[y - 2:y + 3, min(x, 77) - 2:min(x, 77) + 3] = noisy[y - 2, min(x, 77) - 2]
        noisy[y, min(x, 77)] = 11 - noisy[y, min(x, 77)]
    out["supress_in"] = noisy
    out["supress_out"] = ref.supress(noisy)
    # ---- compute_slic_distance on 4096 (pixel, centre) pairs drawn from case A ----
    cen = out["A_centres_it0"]
    k = 4096
    ys, xs, cs = rng.integers(0, 60, k), rng.integers(0, 80, k), rng.integers(0, cen.size, k)
    out["dist_x"], out["dist_y"], out["dist_c"] = xs.astype(np.int32), ys.astype(np.int32), cs.astype(np.int32)
    out["dist_ref"] = np.array([ref.distance(np.ascontiguousarray(lab_a[y, x]), x, y, cen[c:c + 1], 5.0, 1.0 / 20) for y, x, c in zip(ys, xs, cs)], np.float32)
    np.savez_compressed(os.path.join(HERE, "slic_ref.npz"), **out)

    # ---- the figure DESIGN.md quotes: labels of the 1512 x 1008 scene of tests/test_gpu_parity.py when the reference-compiled
    # rgb2CIELab replaces the restatement's (same oracle pipeline from the converted image on) ----
    rng2 = np.random.default_rng(20)
    h, w = 1008, 1512
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (127 + 100 * np.sin(xx / 37.0) * np.cos(yy / 53.0)).astype(np.uint8)
    img[..., 1] = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
    img[..., 2] = (rng2.integers(0, 30, size=(h, w)) + 100 * ((xx // 90 + yy // 70) % 2)).astype(np.uint8)
    img[..., 3] = 255
    lab_ref = ref.cvt(img, 0)
    lab_orc = ol.slic_convert(img, 0).reshape(h, w, 4)
    l_ref, _ = ol.slic_from_lab(lab_ref, 20, 5, 5.0, 1)
    l_orc, _ = ol.slic_from_lab(lab_orc, 20, 5, 5.0, 1)
    census["scene_1512x1008"] = {"lab_components_differing": int((lab_ref[..., :3] != lab_orc[..., :3]).sum()), "lab_components": int(lab_ref[..., :3].size),
                                 "labels_differing": int((l_ref != l_orc).sum()), "labels": int(l_ref.size)}
    lab_old = lab_newton_cube_root(img)
    l_old, _ = ol.slic_from_lab(lab_old, 20, 5, 5.0, 1)
    census["scene_1512x1008_with_the_newton_cube_root_of_rounds_1_to_4"] = {
        "lab_components_differing": int((lab_ref[..., :3] != lab_old[..., :3]).sum()), "labels_differing": int((l_ref != l_old).sum())}
    os.makedirs(os.path.join(ROOT, "profiles", "r05"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r05", "slic_reference_pin.json"), "w") as f:
        json.dump(census, f, indent=1)
    print(json.dumps(census, indent=1))

    # ---- config.h macros: getHomography_cu's chain and matmul_cu / matvecmul on 512 cases each ----
    m0 = C.CDLL(os.path.join(REF, "libmat_ref.so"))
    m1 = C.CDLL(os.path.join(REF, "libmat_ref_fma.so"))
    n = 512
    Kinv, K2, R, t, n4 = (np.zeros((n, s), np.float32) for s in (9, 9, 9, 3, 4))
    H0, H1 = np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32)
    A, B, V = rng.normal(size=(n, 9)).astype(np.float32), rng.normal(size=(n, 9)).astype(np.float32), rng.normal(size=(n, 3)).astype(np.float32)
    AB0, AB1, AV0 = np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32), np.zeros((n, 3), np.float32)
    for i in range(n):
        f = rng.uniform(500, 4000)
        skew = rng.uniform(-2, 2) if i % 4 == 0 else 0.0
        K = np.array([f, skew, rng.uniform(300, 3000), 0, f * rng.uniform(0.98, 1.02), rng.uniform(300, 2000), 0, 0, 1])
        K1 = np.array([f * rng.uniform(0.9, 1.1), 0, rng.uniform(300, 3000), 0, f, rng.uniform(300, 2000), 0, 0, 1])
        Kinv[i] = np.linalg.inv(K1.reshape(3, 3)).ravel()
        K2[i] = K
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        R[i] = q.ravel()
        t[i] = rng.normal(size=3)
        nn = rng.normal(size=3)
        n4[i] = [*(nn / np.linalg.norm(nn)), rng.uniform(0.5, 30) * (1 if i % 2 else -1)]
        for lib, H in ((m0, H0), (m1, H1)):
            lib.ref_homography(p(Kinv[i]), p(K2[i]), p(R[i]), p(t[i]), p(n4[i]), C.c_float(n4[i, 3]), p(H[i]))
        m0.ref_matmul(p(A[i]), p(B[i]), p(AB0[i]))
        m1.ref_matmul(p(A[i]), p(B[i]), p(AB1[i]))
        m0.ref_matvecmul(p(A[i]), p(V[i]), p(AV0[i]))
    np.savez_compressed(os.path.join(HERE, "mat_ref.npz"), Kinv=Kinv, K2=K2, R=R, t=t, n4=n4, H_nocontract=H0, H_gcc_contract=H1,
                        A=A, B=B, V=V, AB_nocontract=AB0, AB_gcc_contract=AB1, AV_nocontract=AV0)
    print("wrote", os.path.join(HERE, "slic_ref.npz"), os.path.getsize(os.path.join(HERE, "slic_ref.npz")), "bytes;",
          os.path.join(HERE, "mat_ref.npz"), os.path.getsize(os.path.join(HERE, "mat_ref.npz")), "bytes")


if __name__ == "__main__":
    main()
