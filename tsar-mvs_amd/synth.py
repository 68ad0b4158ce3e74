"""Synthetic multi-view scenes: analytic surfaces ray-cast into mutually consistent
8-bit gray views with ground-truth depth/normal.  The ETH3D / Middlebury / Tanks-and-Temples data the
reference's scripts point at (reference scripts/courtyard.sh:7-16) is not available offline, so every
test and bench.py use these scenes at the same image sizes and view counts.

torch is used only as an array library here (CPU in tests, the GPU in bench.py); nothing in this
file is on the matcher's data path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch


@dataclass
class Scene:
    w: int
    h: int
    images: list            # n_views x float32 [h, w] tensors, integral values 0..255 (like an 8-bit decode)
    K: np.ndarray           # [n_views, 3, 3] float32
    R: np.ndarray           # [n_views, 3, 3] world->camera
    t: np.ndarray           # [n_views, 3]
    depth_min: float
    depth_max: float
    gt_depth: torch.Tensor  # [h, w] depth of the reference view (view 0)
    gt_normal: torch.Tensor  # [h, w, 3] unit normal in reference-camera coordinates, facing the camera
    gt_prim: torch.Tensor   # [h, w] int32 primitive id (0 back plane, 1 slanted plane, 2 sphere)
    textured: torch.Tensor  # [h, w] bool: False inside textureless patches
    meta: dict = field(default_factory=dict)


def _look_at(C: np.ndarray, target: np.ndarray) -> np.ndarray:
    """world->camera rotation with +z towards the target, +y down."""
    z = target - C
    z = z / np.linalg.norm(z)
    up = np.array([0.0, -1.0, 0.0])
    x = np.cross(-up, z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], 0)


def make_cameras(w: int, h: int, n_src: int, seed: int = 42, step: float = 0.03):
    """Reference camera (view 0) in the middle of an arc, sources alternating left/right of it
    with a baseline/depth ratio of ~`step` per position."""
    rng = np.random.default_rng(seed)
    f = 0.56 * w
    K = np.array([[f, 0, (w - 1) / 2.0], [0, f, (h - 1) / 2.0], [0, 0, 1]], dtype=np.float64)
    dist = 5.0
    angles = [0.0]
    for i in range(n_src):
        k = i // 2 + 1
        angles.append(step * k * (1 if i % 2 == 0 else -1))
    Ks, Rs, ts = [], [], []
    for i, a in enumerate(angles):
        elev = 0.0 if i == 0 else float(rng.uniform(-0.02, 0.02))
        C = np.array([dist * math.sin(a), -dist * math.sin(elev) - 0.3, -dist * math.cos(a)])
        tgt = np.array([0.0, 0.0, 0.0]) if i == 0 else rng.uniform(-0.05, 0.05, 3)
        R = _look_at(C, tgt)
        Ks.append(K)
        Rs.append(R)
        ts.append(-R @ C)
    return (np.stack(Ks).astype(np.float32), np.stack(Rs).astype(np.float32), np.stack(ts).astype(np.float32))


class _Texture:
    """Band-limited albedo: a sum of sinusoids of world position over several octaves."""

    def __init__(self, seed: int, base_freq: float, octaves: int = 5, per_octave: int = 5):
        rng = np.random.default_rng(seed)
        fs, ph, am = [], [], []
        for o in range(octaves):
            for _ in range(per_octave):
                d = rng.normal(size=3)
                d /= np.linalg.norm(d)
                fs.append(d * base_freq * (2.0 ** o))
                ph.append(rng.uniform(0, 2 * math.pi))
                am.append(0.75 ** o)
        self.f = np.array(fs)
        self.p = np.array(ph)
        self.a = np.array(am) / np.sqrt(np.sum(np.square(am)) / 2.0)

    def __call__(self, X: torch.Tensor, prim_shift: torch.Tensor) -> torch.Tensor:
        acc = torch.zeros_like(X[..., 0])
        for f, p, a in zip(self.f, self.p, self.a):
            arg = X[..., 0] * float(f[0]) + X[..., 1] * float(f[1]) + X[..., 2] * float(f[2]) + float(p)
            acc += float(a) * torch.sin(arg + prim_shift)
        return acc


def _render(K, R, t, w, h, tex: _Texture, device, textureless: bool, want_gt: bool, tile_rows: int = 512, flat_cell: float = 0.9):
    """Ray-cast one view.  Returns (gray u8-valued float image, depth, normal_cam, prim, textured)."""
    Kinv = np.linalg.inv(K.astype(np.float64))
    Rt = R.astype(np.float64).T
    C = -Rt @ t.astype(np.float64)
    img = torch.empty((h, w), dtype=torch.float32, device=device)
    depth = torch.empty((h, w), dtype=torch.float32, device=device) if want_gt else None
    normal = torch.empty((h, w, 3), dtype=torch.float32, device=device) if want_gt else None
    prim = torch.empty((h, w), dtype=torch.int32, device=device) if want_gt else None
    textured = torch.empty((h, w), dtype=torch.bool, device=device) if want_gt else None
    # primitives (world): back plane, slanted plane (bounded), sphere
    planes = [
        (np.array([0.05, 0.02, -1.0]) / np.linalg.norm([0.05, 0.02, -1.0]), -1.6),   # n.X = d ; behind the origin
        (np.array([0.55, 0.10, -1.0]) / np.linalg.norm([0.55, 0.10, -1.0]), 0.15),
    ]
    sph_c, sph_r = np.array([-0.9, 0.35, -0.2]), 0.75
    xs = torch.arange(w, dtype=torch.float64, device=device)
    for y0 in range(0, h, tile_rows):
        y1 = min(h, y0 + tile_rows)
        ys = torch.arange(y0, y1, dtype=torch.float64, device=device)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        dc = torch.stack([Kinv[0, 0] * gx + Kinv[0, 1] * gy + Kinv[0, 2], Kinv[1, 0] * gx + Kinv[1, 1] * gy + Kinv[1, 2],
                          torch.ones_like(gx)], -1)                 # camera-frame ray with z = 1: t == depth
        dw = torch.stack([sum(float(Rt[r, c]) * dc[..., c] for c in range(3)) for r in range(3)], -1)
        best_t = torch.full_like(gx, 1e30)
        best_p = torch.zeros_like(gx, dtype=torch.int32)
        nrm = torch.zeros_like(dw)
        for pi, (n, d) in enumerate(planes):
            denom = sum(float(n[c]) * dw[..., c] for c in range(3))
            tt = (d - float(n @ C)) / denom
            X = torch.stack([float(C[c]) + tt * dw[..., c] for c in range(3)], -1)
            ok = (tt > 0.1) & (tt < best_t)
            if pi == 1:   # bounded patch
                ok &= (X[..., 0].abs() < 1.3) & ((X[..., 1] + 0.2).abs() < 0.9)
            best_t = torch.where(ok, tt, best_t)
            best_p = torch.where(ok, torch.full_like(best_p, pi), best_p)
            for c in range(3):
                nrm[..., c] = torch.where(ok, torch.full_like(tt, float(n[c])), nrm[..., c])
        oc = [float(C[c] - sph_c[c]) for c in range(3)]
        a = sum(dw[..., c] ** 2 for c in range(3))
        b = 2 * sum(oc[c] * dw[..., c] for c in range(3))
        cc = sum(o * o for o in oc) - sph_r ** 2
        disc = b * b - 4 * a * cc
        ts_ = (-b - torch.sqrt(disc.clamp_min(0))) / (2 * a)
        ok = (disc > 0) & (ts_ > 0.1) & (ts_ < best_t)
        best_t = torch.where(ok, ts_, best_t)
        best_p = torch.where(ok, torch.full_like(best_p, 2), best_p)
        X = torch.stack([float(C[c]) + best_t * dw[..., c] for c in range(3)], -1)
        for c in range(3):
            nrm[..., c] = torch.where(ok, (X[..., c] - float(sph_c[c])) / sph_r, nrm[..., c])
        shift = best_p.to(torch.float64) * 1.7
        val = tex(X, shift)
        tx = torch.ones_like(best_p, dtype=torch.bool)
        if textureless:
            # constant-albedo patches: squares of a coarse world-space checker on the planes
            cell = (torch.floor(X[..., 0] / flat_cell) + torch.floor(X[..., 1] / flat_cell)).to(torch.int64)
            tx = ~((cell % 3 == 0) & (best_p < 2))
            val = torch.where(tx, val, torch.full_like(val, 0.35) + 0.1 * best_p.to(val.dtype))
        g = torch.clamp(127.5 + 52.0 * val, 0, 255)
        img[y0:y1] = torch.round(g).to(torch.float32)
        if want_gt:
            depth[y0:y1] = best_t.to(torch.float32)
            # world normal -> camera frame, facing the camera (n . viewdir < 0)
            ncam = torch.stack([sum(float(R[r, c]) * nrm[..., c] for c in range(3)) for r in range(3)], -1)
            flip = (sum(ncam[..., c] * dc[..., c] for c in range(3)) > 0)
            ncam = torch.where(flip[..., None], -ncam, ncam)
            normal[y0:y1] = ncam.to(torch.float32)
            prim[y0:y1] = best_p
            textured[y0:y1] = tx
    return img, depth, normal, prim, textured


def make_scene(w: int, h: int, n_src: int, device="cpu", seed: int = 1234, cam_seed: int = 42,
               textureless: bool = False, step: float = 0.03, tex_scale: float = 1.0, flat_cell: float = 0.9, all_gt: bool = False) -> Scene:
    """`n_src` source views + 1 reference view of the analytic scene at w x h.
    The texture's finest wavelength is ~4 pixels at every resolution (tex_scale rescales it)."""
    K, R, t = make_cameras(w, h, n_src, cam_seed, step)
    f = float(K[0, 0, 0])
    # pixel footprint at distance 5 is 5/f world units; finest octave (2^4 * base) ~ 4 px wavelength
    base = 2 * math.pi / (4.0 * 5.0 / f) / 16.0 * tex_scale
    tex = _Texture(seed, base)
    images = []
    gt = None
    gt_all = []
    for v in range(n_src + 1):
        out = _render(K[v], R[v], t[v], w, h, tex, device, textureless, want_gt=(v == 0 or all_gt), flat_cell=flat_cell)
        images.append(out[0])
        if v == 0:
            gt = out[1:]
        if all_gt:
            gt_all.append((out[1], out[2]))   # depth, camera-frame normal of every view (fusion tests)
    dmin = float(gt[0].min()) * 0.8
    dmax = float(gt[0].max()) * 1.25
    return Scene(w, h, images, K, R, t, dmin, dmax, gt[0], gt[1], gt[2], gt[3],
                 meta={"seed": seed, "cam_seed": cam_seed, "step": step, "textureless": textureless, "gt_all": gt_all})


def gt_planes(scene: Scene) -> torch.Tensor:
    """Ground-truth (n, d) per pixel in reference-camera coordinates, n.X + d = 0 (the matcher's
    plane parametrisation, reference linestate.h:12)."""
    K = scene.K[0].astype(np.float64)
    h, w = scene.h, scene.w
    dev = scene.gt_depth.device
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64, device=dev), torch.arange(w, dtype=torch.float64, device=dev), indexing="ij")
    Z = scene.gt_depth.to(torch.float64)
    X = (xs - K[0, 2]) / K[0, 0] * Z
    Y = (ys - K[1, 2]) / K[1, 1] * Z
    n = scene.gt_normal.to(torch.float64)
    d = -(n[..., 0] * X + n[..., 1] * Y + n[..., 2] * Z)
    return torch.cat([n, d[..., None]], -1).to(torch.float32)
