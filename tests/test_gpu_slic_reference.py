"""The HIP SLIC kernels (tsar-mvs_amd/csrc/slic_kernels.hip) against THE REFERENCE ITSELF, through the C ABI
(tsar_selftest_slic_stage, tsar_slic): tests/golden/slic_ref.npz holds outputs of the reference's own
gSLICr_seg_engine_shared.h:7-204 compiled on a host (tests/golden/make_slic_ref_golden.py; the CPU twin of this file is
tests/test_slic_reference_golden.py).  Labels, centres and every float the reference forms from IEEE operations: exact.
rgb2CIELab's pow(): the kernel computes the correctly rounded power; the reference compiled against glibc is within an ulp of it
on 0.07 % of the evaluations (bound and share below, measured over all 2^24 colours in profiles/r05/slic_reference_pin.json)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol
from tsar_mvs_amd import api

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slic_ref.npz")
LAB_MAX_ABS = 6.2e-5          # see tests/test_slic_reference_golden.py
LAB_MAX_SHARE = 0.002


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def m():
    mm = api.Matcher()
    yield mm
    mm.close()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same_centres(a, b, colour_channels=3):
    return (np.array_equal(bits(a["center"]), bits(b["center"])) and np.array_equal(bits(a["color"][:, :colour_channels]), bits(b["color"][:, :colour_channels]))
            and np.array_equal(a["id"], b["id"]) and np.array_equal(a["n"], b["n"]))


def test_colour_conversion_every_8bit_colour(g, m):
    """all 2^24 colours through slic_cvt_kernel: XYZ equals the reference's (hash), CIELAB equals the oracle's correctly rounded
    restatement bit for bit (the fp64 sequence of pow_third is the same IEEE operations on both sides) and is within the stated
    bound of the reference-compiled values on the fixture's 65 536 colours"""
    i = np.arange(1 << 24, dtype=np.uint32)
    allc = np.zeros((1 << 24, 4), np.uint8)
    allc[:, 0], allc[:, 1], allc[:, 2] = i & 255, (i >> 8) & 255, (i >> 16) & 255
    xyz = m.slic_convert(allc, 1)
    assert hashlib.sha256(np.ascontiguousarray(xyz[:, :3]).tobytes()).digest() == g["xyz_all_sha256"].tobytes()
    lab = m.slic_convert(allc, 0)
    assert np.array_equal(bits(lab), bits(ol.slic_convert(allc, 0))), "pow_third on the device differs from the enumerated CPU sequence"
    rgb = m.slic_convert(allc[:65536], 2)
    assert np.array_equal(rgb[:, :3], allc[:65536, :3].astype(np.float32))
    got = m.slic_convert(g["colours"], 0)[:, :3]
    ref = g["colours_lab"]
    dark = g["colours"][:, :3].max(axis=1) <= 2
    assert np.array_equal(bits(got[dark]), bits(ref[dark]))
    d = np.abs(got - ref)
    assert d.max() <= LAB_MAX_ABS and (d > 0).mean() <= LAB_MAX_SHARE, (d.max(), (d > 0).mean())


def test_init_cluster_centers_exact_including_the_edge_branch(g, m):
    assert same_centres(m.slic_init_centers(g["A_lab"], 4, 3, 20), g["A_centres_init"], 4)
    assert same_centres(m.slic_init_centers(g["B_lab"], 4, 3, 20), g["B_centres_init"], 4)


@pytest.mark.parametrize("tag,S,weight", [("A", 20, 5.0), ("C", 12, 3.0)])
def test_find_center_association_labels_exact(g, m, tag, S, weight):
    lab = g[tag + "_lab"]
    h, w = lab.shape[:2]
    l0 = m.slic_find_association(lab, g[tag + "_centres_init"], w // S, h // S, S, weight)
    assert np.array_equal(l0, g[tag + "_labels_init"])
    l1 = m.slic_find_association(lab, g[tag + "_centres_it0"], w // S, h // S, S, weight, l0)
    assert np.array_equal(l1, g[tag + "_labels_it0"])


@pytest.mark.parametrize("tag,S", [("A", 20), ("C", 12)])
def test_centre_update_equals_the_reference_finalize_of_the_block_sums(g, m, tag, S):
    """slic_update_kernel fuses Update_Cluster_Center_device with finalize_reduction_result_shared; the fixture's centres are the
    reference's finalize applied to the oracle's block sums of the same labels"""
    got = m.slic_update_centers(g[tag + "_lab"], g[tag + "_labels_init"], S)
    assert same_centres(got, g[tag + "_centres_it0"], 3)


def test_supress_local_lable_exact(g, m):
    assert np.array_equal(m.slic_connectivity(g["supress_in"]), g["supress_out"])


@pytest.mark.parametrize("tag,S,iters,weight", [("A", 20, 5, 5.0), ("C", 12, 3, 3.0)])
def test_whole_segmentation_stage_by_stage_from_the_reference_converted_image(g, m, tag, S, iters, weight):
    lab = g[tag + "_lab"]
    h, w = lab.shape[:2]
    mw, mh = w // S, h // S
    centres = m.slic_init_centers(lab, mw, mh, S)
    labels = m.slic_find_association(lab, centres, mw, mh, S, weight)
    for _ in range(iters):
        centres = m.slic_update_centers(lab, labels, S)
        labels = m.slic_find_association(lab, centres, mw, mh, S, weight, labels)
    assert np.array_equal(labels, g[tag + "_labels_final"])
    assert same_centres(centres, g[tag + "_centres_final"], 3)
    assert np.array_equal(m.slic_connectivity(m.slic_connectivity(labels)), g[tag + "_labels_connected"])


def test_tsar_slic_from_bgra_gives_the_reference_pipeline_labels(g, m):
    """the product entry point on the fixture's images: the ulp-level Lab differences move no label"""
    assert np.array_equal(m.slic(g["A_bgra"], api.SlicSettings(20, 5, 5.0, 0, 0)), g["A_labels_final"])
    assert np.array_equal(m.slic(g["A_bgra"], api.SlicSettings(20, 5, 5.0, 1, 0)), g["A_labels_connected"])
    assert np.array_equal(m.slic(g["C_bgra"], api.SlicSettings(12, 3, 3.0, 0, 1)), g["C_labels_final"])


def test_stage_hook_rejects_bad_arguments(m):
    st = api.SlicSettings(20, 0, 5.0, 0, 0)
    lab = np.zeros((60, 80, 4), np.float32)
    out = np.zeros(12, api.SPIXEL_DTYPE)
    L = m.L
    assert L.tsar_selftest_slic_stage(m._ctx, 9, 80, 60, 4, 3, C.byref(st), lab.ctypes.data, None, out.ctypes.data) == api.TSAR_ERR_INVALID
    assert L.tsar_selftest_slic_stage(m._ctx, 2, 80, 60, 4, 3, C.byref(st), lab.ctypes.data, None, out.ctypes.data) == api.TSAR_ERR_INVALID      # no centres
    assert L.tsar_selftest_slic_stage(m._ctx, 3, 80, 60, 5, 3, C.byref(st), lab.ctypes.data, lab.ctypes.data, out.ctypes.data) == api.TSAR_ERR_INVALID
    assert L.tsar_selftest_slic_stage(m._ctx, 1, 80, 60, 4, 3, C.byref(st), None, None, out.ctypes.data) == api.TSAR_ERR_INVALID
