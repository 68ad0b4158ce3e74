"""Can the fuser tell the default ("fast") arithmetic from the strict (reference) arithmetic?

Matches an 8-view synthetic scene twice with the C++ host (`tsar_gipuma --all --fuse`, the fuser's gate of the reference's scripts:
x/1.sh:20-30 — num_consistent 1, reproj_error 2 px, depth_diff 0.01, angle 15 deg, used_list 1), once in each mode, same seed, and
compares the two fused clouds (APD/APD_TSAR.ply): point counts, symmetric nearest-neighbour distance relative to depth, and each
cloud's distance to the analytic surface the views were rendered from.  The per-pixel figures of bench.py's `config.tolerance`
(normals within 1 degree: 67 %) say how far the two maps are apart; this says whether that survives the consumer's gate.

    python tools/fused_cloud_fast_vs_strict.py [--size 2016x1344] [--views 8] [--out profiles/r05/fused_cloud_fast_vs_strict.json]

tests/test_gpu_fused_cloud.py runs `compare()` and asserts on its figures.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")
FUSE_GATE = ["--num_consistent=1", "--reproj_error=2", "--depth_diff=0.01", "--angle=15", "--used_list=1"]      # x/1.sh:20-30


def read_cloud(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([ln for ln in head.decode().splitlines() if ln.startswith("element vertex")][0].split()[-1])
    assert len(body) == n * 27
    return np.frombuffer(body, dtype=np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("c", "u1", 3)]))


def surface_distance(P):
    """distance of world points to the analytic scene of tsar_mvs_amd/synth.py (_render: back plane, bounded slanted plane, sphere)"""
    P = P.astype(np.float64)
    n0 = np.array([0.05, 0.02, -1.0]); n0 /= np.linalg.norm(n0)
    n1 = np.array([0.55, 0.10, -1.0]); n1 /= np.linalg.norm(n1)
    d_back = np.abs(P @ n0 - (-1.6))
    d_slant = np.abs(P @ n1 - 0.15)
    inside = (np.abs(P[:, 0]) < 1.3 + 0.05) & (np.abs(P[:, 1] + 0.2) < 0.9 + 0.05)
    d_slant = np.where(inside, d_slant, np.inf)
    d_sph = np.abs(np.linalg.norm(P - np.array([-0.9, 0.35, -0.2]), axis=1) - 0.75)
    return np.minimum(np.minimum(d_back, d_slant), d_sph)


def run_mode(root, strict, iterations, seed, log):
    cmd = [CLI, "--all", "--gpus=1", "--fuse", *FUSE_GATE, "-mslp_folder", root, "-images_folder", root + "images/",
           f"--iterations={iterations}", "--blocksize=11", "--n_best=1", f"--seed={seed}"] + (["--strict"] if strict else [])
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True)
    log.append({"cmd": " ".join(cmd[1:]), "seconds": round(time.time() - t0, 2), "tail": out.stdout.strip().splitlines()[-1:] })
    if out.returncode != 0:
        raise RuntimeError(out.stdout + out.stderr)
    return read_cloud(root + "APD/APD_TSAR.ply")


def nn_stats(A, B, depth_a):
    """for every point of A the distance to its nearest point of B, relative to A's depth"""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(B).query(A, k=1, workers=-1)
    r = d / depth_a
    return {"p50": float(np.percentile(r, 50)), "p90": float(np.percentile(r, 90)), "p99": float(np.percentile(r, 99)), "max": float(r.max()),
            "within_1e-3_of_depth": float((r <= 1e-3).mean()), "within_3e-3_of_depth": float((r <= 3e-3).mean())}


def err_stats(P, depth):
    e = surface_distance(P) / depth
    return {"p50": float(np.percentile(e, 50)), "p90": float(np.percentile(e, 90)), "p99": float(np.percentile(e, 99)), "mean": float(e.mean()),
            "within_1e-3_of_depth": float((e <= 1e-3).mean()), "within_1e-2_of_depth": float((e <= 1e-2).mean())}


def compare(w=2016, h=1344, n_views=8, iterations=8, seed=3, workdir=None, control=True):
    import torch
    from tsar_mvs_amd import io as tio
    from tsar_mvs_amd import synth
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    sc = synth.make_scene(w, h, n_views - 1, device=dev, seed=77)
    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="tsar_cloud_")
    rec = {"scene": f"{n_views} views {w}x{h}, synthetic (tsar_mvs_amd/synth.py seed 77), every view paired with all others", "iterations": iterations,
           "matcher": "box 11, n_best 1", "fuser_gate": " ".join(FUSE_GATE) + "  (reference x/1.sh:20-30)", "runs": []}
    try:
        roots = {}
        for mode in ("fast", "strict") + (("strict_other_seed",) if control else ()):
            roots[mode] = os.path.join(workdir, mode) + "/"
            tio.export_scene(sc, roots[mode])
        R0, t0 = sc.R[0].astype(np.float64), sc.t[0].astype(np.float64)
        depth_of = lambda P: np.maximum((P.astype(np.float64) @ R0.T + t0)[:, 2], 1e-6)       # noqa: E731  depth in view 0 (all cameras look at the origin from ~5)
        clouds = {"fast": run_mode(roots["fast"], False, iterations, seed, rec["runs"]), "strict": run_mode(roots["strict"], True, iterations, seed, rec["runs"])}
        if control:   # what the reference does to itself at every launch (clock-seeded RNG, gipuma.cu:700,1077): strict vs strict under another seed
            clouds["strict_other_seed"] = run_mode(roots["strict_other_seed"], True, iterations, seed + 1000, rec["runs"])
        P = {k: np.ascontiguousarray(v["p"]) for k, v in clouds.items()}
        D = {k: depth_of(v) for k, v in P.items()}
        rec["points"] = {k: int(v.shape[0]) for k, v in P.items()}
        rec["point_count_ratio_fast_over_strict"] = P["fast"].shape[0] / P["strict"].shape[0]
        rec["strict_to_nearest_fast"] = nn_stats(P["strict"], P["fast"], D["strict"])
        rec["fast_to_nearest_strict"] = nn_stats(P["fast"], P["strict"], D["fast"])
        rec["error_against_analytic_surface"] = {k: err_stats(P[k], D[k]) for k in P}
        nf, ns = clouds["fast"]["n"].astype(np.float64), clouds["strict"]["n"].astype(np.float64)
        from scipy.spatial import cKDTree
        _, j = cKDTree(P["fast"]).query(P["strict"], k=1, workers=-1)
        ang = np.degrees(np.arccos(np.clip(np.abs((ns * nf[j]).sum(1)), 0, 1)))
        rec["normal_angle_strict_vs_nearest_fast_deg"] = {"p50": float(np.percentile(ang, 50)), "p90": float(np.percentile(ang, 90)), "p99": float(np.percentile(ang, 99)),
                                                          "within_15_deg": float((ang <= 15).mean())}
        if control:
            rec["control_point_count_ratio"] = P["strict_other_seed"].shape[0] / P["strict"].shape[0]
            rec["control_strict_to_nearest_strict_other_seed"] = nn_stats(P["strict"], P["strict_other_seed"], D["strict"])
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="2016x1344")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--iterations", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    rec = compare(w, h, a.views, a.iterations)
    txt = json.dumps(rec, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
