"""The image bytes the matcher starts from: host/tsar_jpeg.h against libjpeg-turbo (the one inside Pillow, in this image).

The reference reads its views with OpenCV's imread(..., IMREAD_GRAYSCALE) (main.cpp:1302; IMREAD_COLOR with -color_processing,
:1304) and its scenes hold JPEGs.  OpenCV hands a JPEG to libjpeg with out_color_space = JCS_GRAYSCALE: the luminance component
as decoded.  Pillow's draft("L") asks its libjpeg-turbo for exactly that output, and its RGB decode is libjpeg's JCS_RGB with
fancy upsampling, so Pillow is the arbiter here: every sample must be identical.  OpenCV itself is not in the image (parity with
it: unpinned, argued from the shared library).  No GPU: `tsar_gipuma --decode-image=` runs the tool's own reader.
"""
import os
import subprocess

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from tsar_mvs_amd import io as tio  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")


@pytest.fixture(scope="module", autouse=True)
def _cli():
    if not os.path.exists(CLI):
        import __graft_entry__ as ge
        ge.build()


def ours(path, out, colour=False):
    r = subprocess.run([CLI] + (["-color_processing"] if colour else []) + [f"--decode-image={path}:{out}"], capture_output=True, text=True)
    if r.returncode != 0 or "cannot decode" in r.stdout:
        return r.stdout.strip()
    return tio.read_pgm(out).astype(np.uint8)


def libjpeg_luma(path):
    im = Image.open(path)
    im.draft("L", im.size)
    assert im.mode == "L"
    return np.asarray(im)


def libjpeg_blue(path):
    im = Image.open(path)
    return np.asarray(im) if im.mode == "L" else np.asarray(im.convert("RGB"))[..., 2]


def synthetic(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 100 * np.cos(x / 5.0 - y / 9.0), (x * 3 + y * 5) % 256], -1) + rng.normal(0, 12, (h, w, 3))
    return np.clip(a, 0, 255).astype(np.uint8)


SIZES = [(1, 1), (8, 8), (17, 16), (37, 53), (3, 200), (201, 5), (123, 257)]


@pytest.mark.parametrize("progressive", [False, True])
@pytest.mark.parametrize("subsampling", [0, 1, 2])          # 4:4:4, 4:2:2, 4:2:0
def test_every_sample_equals_libjpeg(tmp_path, subsampling, progressive):
    """sizes that are not whole MCUs (edge blocks, the replicated chroma rows of the fancy upsampling), one- and two-sample-wide
    chroma planes (plain replication instead of the triangle filter), qualities from heavy quantisation to 100 (16-bit products in
    the inverse DCT), default and optimised Huffman tables — luminance (the default path) and the blue of the colour path"""
    p, out = str(tmp_path / "a.jpg"), str(tmp_path / "o.pgm")
    checked = 0
    for k, (h, w) in enumerate(SIZES):
        for quality in (35, 90, 100):
            for optimize in (False, True):
                try:
                    Image.fromarray(synthetic(h, w, k)).save(p, quality=quality, subsampling=subsampling, progressive=progressive, optimize=optimize)
                except OSError:                      # Pillow's encoder buffer is too small for a few of these; nothing to compare
                    continue
                for colour in (False, True):
                    got, want = ours(p, out, colour), (libjpeg_blue(p) if colour else libjpeg_luma(p))
                    assert not isinstance(got, str), got
                    assert got.shape == want.shape and np.array_equal(got, want), (h, w, quality, optimize, colour, int((got != want).sum()))
                    checked += 1
    assert checked >= 60


def test_restart_intervals_and_gray_files(tmp_path):
    p, out = str(tmp_path / "a.jpg"), str(tmp_path / "o.pgm")
    rgb = synthetic(123, 257, 3)
    for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=3), dict(restart_marker_blocks=7, subsampling=2),
               dict(restart_marker_blocks=2, progressive=True, subsampling=1), dict(restart_marker_rows=1)):
        try:
            Image.fromarray(rgb).save(p, quality=80, **kw)
        except TypeError:                            # a Pillow without the restart options
            pytest.skip("this Pillow cannot write restart markers")
        assert b"\xff\xdd" in open(p, "rb").read()   # a DRI segment is there
        for colour in (False, True):
            got = ours(p, out, colour)
            assert not isinstance(got, str), got
            assert np.array_equal(got, libjpeg_blue(p) if colour else libjpeg_luma(p)), kw
    for progressive in (False, True):                # single-component files: the one plane, whichever path asks
        Image.fromarray(rgb[..., 0]).save(p, quality=85, progressive=progressive)
        for colour in (False, True):
            assert np.array_equal(ours(p, out, colour), np.asarray(Image.open(p)))


PHOTOS = ["/usr/local/lib/python3.10/dist-packages/sklearn/datasets/images/china.jpg",
          "/usr/local/lib/python3.10/dist-packages/sklearn/datasets/images/flower.jpg",
          "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/sample_data/grace_hopper.jpg"]


def test_photographs_that_ship_with_the_image(tmp_path):
    """camera JPEGs written by other encoders (the sample images of scikit-learn and matplotlib, where the image has them)"""
    out = str(tmp_path / "o.pgm")
    seen = 0
    for f in PHOTOS:
        if not os.path.exists(f):
            continue
        seen += 1
        for colour in (False, True):
            got = ours(f, out, colour)
            assert not isinstance(got, str), got
            assert np.array_equal(got, libjpeg_blue(f) if colour else libjpeg_luma(f)), (f, colour)
    if not seen:
        pytest.skip("no sample photographs in this image")


def test_what_is_not_supported_is_refused_not_guessed(tmp_path):
    p, out = str(tmp_path / "a.jpg"), str(tmp_path / "o.pgm")
    rgb = synthetic(40, 56, 9)
    Image.fromarray(rgb).convert("CMYK").save(p, quality=90)
    assert "components" in ours(p, out)
    Image.fromarray(rgb).save(p, quality=90)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[: len(raw) // 2])        # truncated: libjpeg would pad with gray; a matcher input must not be half an image
    assert "cannot decode" in ours(p, out)
    Image.fromarray(rgb).save(p, quality=90, progressive=True)
    raw = open(p, "rb").read()
    scans = [i for i in range(len(raw) - 1) if raw[i] == 0xFF and raw[i + 1] == 0xDA]
    assert len(scans) >= 4
    open(p, "wb").write(raw[: scans[2]] + b"\xff\xd9")   # a progressive file that ends, well-formed, after two scans: libjpeg would smooth its blocks
    assert "low frequencies" in ours(p, out)
    open(p, "wb").write(b"\x89PNG not a jpeg at all")
    assert "cannot decode" in ours(p, out)
    assert "cannot decode" in ours(str(tmp_path / "missing.jpg"), out)


def test_convert_image_takes_libjpegs_gray_not_a_conversion_of_its_rgb(tmp_path):
    """python -m tsar_mvs_amd.io convert: the PGM a JPEG turns into is what imread(..., IMREAD_GRAYSCALE) returns (the luminance
    plane), and differs from the RGB -> L conversion earlier rounds took wherever the image is coloured"""
    p, pgm = str(tmp_path / "a.jpg"), str(tmp_path / "a.pgm")
    Image.fromarray(synthetic(64, 96, 4)).save(p, quality=90, subsampling=2)
    tio.convert_image(p, pgm)
    got = tio.read_pgm(pgm).astype(np.uint8)
    assert np.array_equal(got, libjpeg_luma(p))
    via_rgb = np.asarray(Image.open(p).convert("L"))
    assert (got != via_rgb).any() and np.abs(got.astype(int) - via_rgb).max() <= 8      # a few levels where colours saturate


def test_a_view_named_pgm_falls_back_to_the_jpeg_beside_it(tmp_path):
    """the tools keep the reference's command lines (`<ref>.jpg <src>.jpg ...`, --all's %08d names): the PGM of that name where it
    exists (a user's own conversion wins), else the JPEG of the same stem"""
    rgb = synthetic(48, 64, 6)
    j, out = str(tmp_path / "00000003.jpg"), str(tmp_path / "o.pgm")
    Image.fromarray(rgb).save(j, quality=90)
    want = libjpeg_luma(j)
    assert np.array_equal(ours(str(tmp_path / "00000003.pgm"), out), want)        # no such PGM: the JPEG
    tio.write_pgm(str(tmp_path / "00000003.pgm"), np.full((48, 64), 7, np.float32))
    assert (ours(str(tmp_path / "00000003.pgm"), out) == 7).all()                 # the PGM wins when it is there
