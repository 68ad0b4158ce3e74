"""Row A15 (gSLICr) of the oracle against THE REFERENCE ITSELF: tests/golden/slic_ref.npz holds outputs of the reference's own
gSLICr_seg_engine_shared.h:7-204, compiled in the build container from the sources where they lie (oracle/Makefile `ref`,
oracle/ref_harness/slic_ref.cpp, tests/golden/make_slic_ref_golden.py).  Integer / label outputs and every float the reference
computes from IEEE operations must be reproduced EXACTLY; rgb2CIELab goes through pow(), a libm call, for which the bound and the
share that differs are stated below.  The -m gpu twin (tests/test_gpu_slic_reference.py) holds the HIP kernels to the same file."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slic_ref.npz")

# rgb2CIELab vs the reference compiled against glibc 2.35 (whose powf is within 1 ulp, not correctly rounded): the restatement's
# pow is the correctly rounded one, so the two differ only where glibc's is off by an ulp — measured over all 2^24 colours
# (profiles/r05/slic_reference_pin.json): 56 642 of 50 331 648 components (0.11 %), max 5.73e-5.  One ulp of fx or fz near 1 is
# 6e-8; it enters a = 500 (fx - fy) and b = 200 (fy - fz), whose own rounding (values up to ~128: ulp 7.6e-6) adds at most 2 ulps.
LAB_MAX_ABS = 6.2e-5
LAB_MAX_SHARE = 0.002


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same_centres(a, b, colour_channels=3):
    return (np.array_equal(bits(a["center"]), bits(b["center"])) and np.array_equal(bits(a["color"][:, :colour_channels]), bits(b["color"][:, :colour_channels]))
            and np.array_equal(a["id"], b["id"]) and np.array_equal(a["n"], b["n"]))


def test_rgb2xyz_every_colour_exact(g):
    xyz = ol.slic_convert(g["colours"][:4096], 1)[:, :3]
    assert np.array_equal(bits(xyz), bits(g["colours_xyz"]))
    i = np.arange(1 << 24, dtype=np.uint32)
    allc = np.zeros((1 << 24, 4), np.uint8)
    allc[:, 0], allc[:, 1], allc[:, 2] = i & 255, (i >> 8) & 255, (i >> 16) & 255
    got = np.ascontiguousarray(ol.slic_convert(allc, 1)[:, :3])
    assert hashlib.sha256(got.tobytes()).digest() == g["xyz_all_sha256"].tobytes(), "rgb2xyz differs from the reference on some of the 2^24 colours"


def test_rgb2cielab_linear_branch_exact_and_pow_branch_within_an_ulp(g):
    col, ref = g["colours"], g["colours_lab"]
    got = ol.slic_convert(col, 0)[:, :3]
    dark = col[:, :3].max(axis=1) <= 2              # x, y, z <= 0.0079 < epsilon: all three through (kappa x + 16) / 116 (shared.h:42-46)
    assert dark.sum() >= 27
    assert np.array_equal(bits(got[dark]), bits(ref[dark])), "the linear branch of rgb2CIELab is IEEE arithmetic: must be exact"
    d = np.abs(got - ref)
    assert d.max() <= LAB_MAX_ABS, d.max()
    assert (d > 0).mean() <= LAB_MAX_SHARE, (d > 0).mean()
    assert np.array_equal(bits(got[:, 0][col[:, :3].max(axis=1) == 0]), bits(ref[:, 0][col[:, :3].max(axis=1) == 0]))
    # L depends on fy alone: where L agrees exactly and a / b do not, the disagreement is fx / fz's pow — sanity of the reading above
    assert (d[:, 0] > 0).sum() <= (d[:, 1] > 0).sum() + (d[:, 2] > 0).sum()


def test_pow_third_is_the_correctly_rounded_power_on_every_reachable_argument():
    """oracle/tsar_oracle_slic.c orc_pow_third against powl (64-bit mantissa) rounded to fp32, on every argument an 8-bit colour can
    hand to rgb2CIELab's pow(): the enumeration the kernel's comment cites"""
    n, bad, libm = ol.pow_third_check()
    assert n == 50329213 and bad == 0
    assert 0 < libm < n // 500, "this host's powf should be close to, and not identical with, the correctly rounded power"
    for x in (0.008857, 0.1, 0.5, 1.0, 1.0888):
        want = np.float32(np.power(np.longdouble(np.float32(x)), np.longdouble(np.float32(1.0) / np.float32(3.0))))
        assert np.float32(ol.pow_third(np.float32(x))) == want


def test_init_cluster_centers_exact_including_the_edge_branch(g):
    a = ol.slic_init_centers(g["A_lab"], 4, 3, 20)
    assert same_centres(a, g["A_centres_init"], 4)
    b = ol.slic_init_centers(g["B_lab"], 4, 3, 20)      # 70 x 50 with a 4 x 3 map: column 3 / row 2 take (x S + w) / 2 (shared.h:83-84)
    assert same_centres(b, g["B_centres_init"], 4)
    assert b["center"][3, 0] == 65.0 and b["center"][8, 1] == 45.0


def test_slic_distance_exact(g):
    lab, cen = g["A_lab"], g["A_centres_it0"]
    got = np.array([ol.slic_distance(lab[y, x], x, y, cen[c:c + 1], 5.0, np.float32(1.0) / np.float32(20)) for x, y, c in zip(g["dist_x"], g["dist_y"], g["dist_c"])], np.float32)
    assert np.array_equal(bits(got), bits(g["dist_ref"]))


@pytest.mark.parametrize("tag,S,weight", [("A", 20, 5.0), ("C", 12, 3.0)])
def test_find_center_association_labels_exact(g, tag, S, weight):
    lab = g[tag + "_lab"]
    h, w = lab.shape[:2]
    mw, mh = w // S, h // S
    l0 = ol.slic_find_association(lab, g[tag + "_centres_init"], mw, mh, S, weight)
    assert np.array_equal(l0, g[tag + "_labels_init"])
    l1 = ol.slic_find_association(lab, g[tag + "_centres_it0"], mw, mh, S, weight, l0)
    assert np.array_equal(l1, g[tag + "_labels_it0"])


@pytest.mark.parametrize("tag", ["A", "C"])
def test_finalize_reduction_result_exact(g, tag):
    got = ol.slic_finalize(g[tag + "_accum_it0"])
    ref = g[tag + "_centres_it0"]
    got["id"] = ref["id"]                                # finalize leaves .id untouched (shared.h:151-173)
    assert same_centres(got, ref, 4)


def test_supress_local_lable_exact(g):
    got = ol.slic_connectivity(g["supress_in"])
    assert np.array_equal(got, g["supress_out"])
    assert (got != g["supress_in"]).sum() >= 20, "the fixture must exercise the >= 16 rule"


@pytest.mark.parametrize("tag,S,iters,weight", [("A", 20, 5, 5.0), ("C", 12, 3, 3.0)])
def test_whole_segmentation_from_the_reference_converted_image(g, tag, S, iters, weight):
    """Perform_Segmentation from the reference's own converted image: every stage of the fixture's run is the reference's function
    except the block sums of Update_Cluster_Center_device (the oracle's, the one stage no host compiler reaches)"""
    labels, centres = ol.slic_from_lab(g[tag + "_lab"], S, iters, weight, 0)
    assert np.array_equal(labels, g[tag + "_labels_final"])
    assert same_centres(centres, g[tag + "_centres_final"], 3)
    connected, _ = ol.slic_from_lab(g[tag + "_lab"], S, iters, weight, 1)
    assert np.array_equal(connected, g[tag + "_labels_connected"])


def test_whole_segmentation_from_bgra_agrees_with_the_reference_pipeline(g):
    """the same run entered through the restatement's own rgb2CIELab: the 0.1 % of Lab components that differ by an ulp move no label"""
    labels = ol.slic(g["A_bgra"], 20, 5, 5.0, 0, 0)
    assert np.array_equal(labels, g["A_labels_final"])
    lab = ol.slic_convert(g["C_bgra"], 1).reshape(g["C_lab"].shape)
    assert np.array_equal(bits(lab[..., :3]), bits(g["C_lab"][..., :3]))
