"""getHomography_cu's arithmetic (gipuma.cu:207-224) against THE REFERENCE'S OWN MACROS: tests/golden/mat_ref.npz holds what
config.h:60-240 (outer_product, matdivide, matmatsub2, matmul_cu, matvecmul — a header of #defines over float arrays, compilable
as it stands) computes when expanded by g++ in the build container (oracle/ref_harness/mat_ref.cpp, oracle/Makefile `ref`,
tests/golden/make_slic_ref_golden.py).  gipuma.cu itself stays unbuildable; this pins the one thing rounds 1-2 misread by
re-reading only: which operands meet in which order, and that t n^T is DIVIDED by d element by element.
The oracle places fused multiply-adds explicitly (tsar_oracle.c S4: mul, fma, fma per dot product — nvcc's default -fmad=true
shape); a host g++ has no say on where nvcc fuses.  So: the oracle built with every fmaf() as multiply-then-add must equal the
reference's macros compiled without contraction BIT FOR BIT, and the shipped oracle may differ from that only by the fusing."""
import os

import numpy as np

import oracle_lib as ol

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mat_ref.npz"))


def _run(L, name, *arrays, n_out=9):
    n = arrays[0].shape[0]
    out = np.zeros((n, n_out), np.float32)
    for i in range(n):
        getattr(L, name)(*[ol._p(np.ascontiguousarray(a[i])) for a in arrays], ol._p(out[i]))
    return out


def test_unfused_oracle_reproduces_the_reference_macros_bit_for_bit():
    L = ol.nofma_lib()
    H = _run(L, "orc_homography_arrays", G["Kinv"], G["K2"], G["R"], G["t"], G["n4"])
    assert np.array_equal(H.view(np.uint32), G["H_nocontract"].view(np.uint32))
    AB = _run(L, "orc_mat3mul", G["A"], G["B"])
    assert np.array_equal(AB.view(np.uint32), G["AB_nocontract"].view(np.uint32))
    AV = _run(L, "orc_mat3vec", G["A"], G["V"], n_out=3)
    assert np.array_equal(AV.view(np.uint32), G["AV_nocontract"].view(np.uint32))


def test_multiplying_by_the_reciprocal_of_d_would_be_caught():
    """the round-1 misreading (t n^T * (1 / d) instead of matdivide) differs from the fixture on a visible share of the cases"""
    Kinv, K2, R, t, n4 = (G[k] for k in ("Kinv", "K2", "R", "t", "n4"))
    n = Kinv.shape[0]
    bad = 0
    for i in range(n):
        inv = np.float32(1.0) / n4[i, 3]
        M = (R[i].reshape(3, 3) - (np.outer(t[i], n4[i, :3]).astype(np.float32) * inv)).astype(np.float32)
        M_ok = (R[i].reshape(3, 3) - (np.outer(t[i], n4[i, :3]).astype(np.float32) / n4[i, 3])).astype(np.float32)
        bad += not np.array_equal(M, M_ok)
    assert bad > n // 8          # measured 121 of 512: most one-ulp differences of t n^T / d vanish in R minus it, these do not


def test_shipped_oracle_differs_from_the_macros_only_by_the_fusing():
    L = ol.lib()
    H = _run(L, "orc_homography_arrays", G["Kinv"], G["K2"], G["R"], G["t"], G["n4"])
    ref = G["H_nocontract"]
    # one fused chain per dot product saves two roundings of terms up to |a b| + |c d| + |e f|: a few ulps of the largest term
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
    assert np.max(np.abs(H - ref) / scale) < 4e-6
    # g++'s own contraction (-ffp-contract=fast -mfma) fuses other products than the oracle's nvcc-shaped chain: recorded, not equal
    assert not np.array_equal(H.view(np.uint32), G["H_gcc_contract"].view(np.uint32))
    assert np.max(np.abs(G["H_gcc_contract"] - ref) / scale) < 4e-6
