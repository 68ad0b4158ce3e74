"""Row N1: the reference's file formats (cams/%08d_cam.txt, pair.txt, .dmb) and the C++ CLI that keeps
its command line."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tsar_mvs_amd import io as tio
from tsar_mvs_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tsar-mvs_amd", "tsar_gipuma")


def test_dmb_layout_is_the_reference_layout(tmp_path):
    """int32 type=1, h, w, channels, then float32 row-major (reference fileIoUtils.h:333-381)"""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    p = str(tmp_path / "d.dmb")
    tio.write_dmb(p, a)
    raw = open(p, "rb").read()
    assert struct.unpack("<iiii", raw[:16]) == (1, 2, 3, 1)
    assert np.array_equal(np.frombuffer(raw[16:], "<f4"), a.ravel())
    assert np.array_equal(tio.read_dmb(p), a)
    n = np.random.default_rng(0).normal(size=(4, 5, 3)).astype(np.float32)
    tio.write_dmb(p, n)
    assert struct.unpack("<iiii", open(p, "rb").read(16)) == (1, 4, 5, 3)
    assert np.array_equal(tio.read_dmb(p), n)
    open(p, "wb").write(struct.pack("<iiii", 2, 1, 1, 1) + b"\0" * 4)
    with pytest.raises(ValueError):
        tio.read_dmb(p)            # the reference only supports float (type 1)


def test_cam_and_pair_files_round_trip(tmp_path):
    sc = synth.make_scene(64, 48, 3, seed=2)
    root = str(tmp_path)
    tio.export_scene(sc, root)
    for k in range(4):
        K, R, t, dmin, dmax = tio.read_cam(os.path.join(root, "cams", f"{k:08d}_cam.txt"))
        assert np.allclose(K, sc.K[k]) and np.allclose(R, sc.R[k]) and np.allclose(t, sc.t[k])
        assert abs(dmin - sc.depth_min) < 1e-6 and abs(dmax - sc.depth_max) < 1e-6
        img = tio.read_pgm(os.path.join(root, "images", f"{k:08d}.pgm"))
        assert np.array_equal(img, sc.images[k].numpy())
    pairs = tio.read_pairs(os.path.join(root, "pair.txt"))
    assert sorted(pairs) == [0, 1, 2, 3] and [s for s, _ in pairs[1]] == [0, 2, 3]
    # slot of a source view in the reference's argv list: id if id > ref else id + 1 (main.cpp:1371-1375)
    assert tio.source_slots(1, [0, 2, 3]) == [1, 2, 3]
    assert tio.source_slots(0, [1, 2, 3]) == [1, 2, 3]
    assert tio.source_slots(3, [0, 1, 2]) == [1, 2, 3]


def test_cli_builds_and_keeps_the_reference_flags():
    import __graft_entry__ as ge
    if not os.path.exists(CLI):
        ge.build()
    out = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "-mslp_folder" in out.stdout and "--blocksize" in out.stdout
    bad = subprocess.run([CLI, "a.pgm", "b.pgm", "-mslp_folder", "x/", "-images_folder", "y/", "--blocksize=10"], capture_output=True, text=True)
    assert bad.returncode != 0 and "positive odd number" in bad.stdout       # main.cpp:823-831
    warn = subprocess.run([CLI, "--frobnicate", "--help"], capture_output=True, text=True)
    assert "unknown option --frobnicate" in warn.stdout                       # unknown flags only warn (main.cpp:941-944)


@pytest.mark.gpu
def test_cli_matches_the_library(tmp_path):
    """the per-view command line of the reference's shell loop (scripts/courtyard.sh:44) end to end"""
    from tsar_mvs_amd import api
    sc = synth.make_scene(128, 96, 3, seed=5)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    names = [f"{k:08d}.pgm" for k in (1, 0, 2, 3)]          # reference view 1 first, then every other image
    cmd = [CLI, *names, "-mslp_folder", root, "-images_folder", root + "images/", "-krt_file", "unused", "-output_folder", root + "out/",
           "-no_display", "--cam_scale=1", "--iterations=2", "--blocksize=11", "--cost_gamma=10", "--cost_comb=best_n", "--n_best=1", "--seed=7"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    depth = tio.read_dmb(root + "APD/00000001/TSAR_disp.dmb")
    normal = tio.read_dmb(root + "APD/00000001/TSAR_normals.dmb")
    assert depth.shape == (96, 128) and normal.shape == (96, 128, 3)
    order = [1, 0, 2, 3]
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max, seed=7 + 1))
    m.set_views([sc.images[k] for k in order], sc.K[order], sc.R[order], sc.t[order])
    m.set_view_subset(tio.source_slots(1, [0, 2, 3]))
    m.pm_init()
    m.pm_iterate(2)
    m.compute_disp()
    res = m.get_result()
    assert np.array_equal(depth, res["depth"]) and np.array_equal(normal, res["normal"])
    m.close()
    # --all: every view of pair.txt, one thread per GPU
    out = subprocess.run([CLI, "--all", "--gpus=1", "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=1", "--blocksize=11", "--n_best=1"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert all(os.path.exists(root + f"APD/{k:08d}/TSAR_disp.dmb") for k in range(4))


@pytest.mark.gpu
def test_cli_reads_a_scene_of_jpegs_like_the_reference(tmp_path):
    """scripts/courtyard.sh:7,16,44: the scene folder holds JPEGs and the command line names them.  The tool decodes them itself
    (host/tsar_jpeg.h: libjpeg's grayscale output, what imread(..., IMREAD_GRAYSCALE) returns, main.cpp:1302): same maps, byte for
    byte, as on PGMs holding libjpeg's own decode of the same files; --all finds the JPEGs behind its %08d names too"""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    import shutil
    sc = synth.make_scene(128, 96, 3, seed=5)
    a, b = str(tmp_path / "jpg") + "/", str(tmp_path / "pgm") + "/"
    tio.export_scene(sc, a)
    shutil.copytree(a, b)
    for k in range(4):
        g = np.clip(sc.images[k].cpu().numpy(), 0, 255).astype(np.uint8)
        rgb = np.stack([g, np.clip(0.8 * g.astype(np.float32) + 30, 0, 255).astype(np.uint8), 255 - g], -1)      # a coloured photograph
        os.remove(a + f"images/{k:08d}.pgm")
        os.remove(b + f"images/{k:08d}.pgm")
        Image.fromarray(rgb).save(a + f"images/{k:08d}.jpg", quality=92, subsampling=2)
        tio.convert_image(a + f"images/{k:08d}.jpg", b + f"images/{k:08d}.pgm")
    outs = []
    for root, ext in ((a, "jpg"), (b, "pgm")):
        names = [f"{k:08d}.{ext}" for k in (1, 0, 2, 3)]
        out = subprocess.run([CLI, *names, "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=2", "--blocksize=11", "--n_best=1", "--seed=7"],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        outs.append((open(root + "APD/00000001/TSAR_disp.dmb", "rb").read(), open(root + "APD/00000001/TSAR_normals.dmb", "rb").read()))
    assert outs[0] == outs[1]
    out = subprocess.run([CLI, "--all", "--gpus=1", "-mslp_folder", a, "-images_folder", a + "images/", "--iterations=1", "--blocksize=11", "--n_best=1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert all(os.path.exists(a + f"APD/{k:08d}/TSAR_disp.dmb") for k in range(4))
    bad = subprocess.run([CLI, "00000001.jpg", "00000000.jpg", "00000002.jpg", "00000009.jpg", "-mslp_folder", a, "-images_folder", a + "images/", "--iterations=1"], capture_output=True, text=True)
    assert bad.returncode != 0 and "cannot read image" in bad.stderr


FUSION = os.path.join(ROOT, "tsar-mvs_amd", "tsar_fusion")


@pytest.mark.gpu
def test_fusion_cli_reads_what_the_matcher_cli_writes(tmp_path):
    """the reference's two-stage pipeline: per-view matcher runs, then `Fusion <dir> --num_consistent= 1 ...`
    (x/1.sh:30, with its blank after '=') producing APD/APD_TSAR.ply"""
    sc = synth.make_scene(160, 120, 3, seed=6)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    out = subprocess.run([CLI, "--all", "--gpus=1", "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=3", "--blocksize=11", "--n_best=1"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    out = subprocess.run([FUSION, root, "--num_consistent=", "2", "--reproj_error=", "2", "--depth_diff=", "0.01", "--angle=", "15", "--used_list=", "1"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "num_consistent: 2" in out.stdout
    raw = open(root + "APD/APD_TSAR.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([ln for ln in head.decode().splitlines() if ln.startswith("element vertex")][0].split()[-1])
    assert n > 1000 and len(body) == n * 27
    rec = np.frombuffer(body, dtype=np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("c", "u1", 3)]))
    assert np.isfinite(rec["p"]).all() and np.allclose(np.linalg.norm(rec["n"], axis=1), 1, atol=1e-4)


@pytest.mark.gpu
def test_all_fuse_in_one_process_equals_the_two_stage_pipeline(tmp_path):
    """tsar_gipuma --all --fuse keeps every view's maps on the GPU, gathers them to the fusing GPU (tsar_peer_copy; with one
    GPU the copies are device-local) and fuses there: the cloud must be byte-identical to what the file-based second stage
    (tsar_fusion on the .dmb files of the same run) writes"""
    sc = synth.make_scene(160, 120, 3, seed=6)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    out = subprocess.run([CLI, "--all", "--gpus=1", "--workers=2", "--fuse", "--num_consistent=2", "-mslp_folder", root, "-images_folder", root + "images/",
                          "--iterations=3", "--blocksize=11", "--n_best=1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fused 4 views on gpu 0" in out.stdout
    in_process = open(root + "APD/APD_TSAR.ply", "rb").read()
    os.remove(root + "APD/APD_TSAR.ply")
    out = subprocess.run([FUSION, root, "--num_consistent=", "2"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert open(root + "APD/APD_TSAR.ply", "rb").read() == in_process
    assert len(in_process) > 27 * 1000


def test_weak_png_mask_decoding(tmp_path):
    """weak.png of the reference's live path (main.cpp:1499-1514): white / pure green / pure red pixels are
    reliable.  The C++ reader (host/tsar_io.h, zlib) must undo every PNG row filter."""
    import __graft_entry__ as ge
    if not os.path.exists(CLI):
        ge.build()
    rng = np.random.default_rng(0)
    h, w = 37, 53
    rgb = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    m = rng.uniform(size=(h, w))
    rgb[m < 0.3] = (255, 255, 255)
    rgb[(m >= 0.3) & (m < 0.4)] = (0, 255, 0)
    rgb[(m >= 0.4) & (m < 0.5)] = (255, 0, 0)
    rgb[(m >= 0.5) & (m < 0.6)] = (0, 0, 255)          # pure blue is NOT reliable
    reliable = np.all(rgb == 255, -1) | np.all(rgb == (0, 255, 0), -1) | np.all(rgb == (255, 0, 0), -1)
    idx = np.nonzero(reliable.ravel())[0]
    want = f"mask {w} x {h} reliable {reliable.sum()} checksum {int((idx % 9973).sum())}"
    for name, filters in (("plain", None), ("filtered", [0, 1, 2, 3, 4]), ("paeth", [4])):
        p = str(tmp_path / f"{name}.png")
        tio.write_png(p, rgb, filters=filters)
        out = subprocess.run([CLI, f"--check-mask={p}"], capture_output=True, text=True).stdout
        assert want in out, (name, out)
    g = str(tmp_path / "gray.png")
    tio.write_reliable_mask(g, reliable)
    assert want in subprocess.run([CLI, f"--check-mask={g}"], capture_output=True, text=True).stdout
    open(g, "wb").write(b"not a png")
    assert "cannot decode" in subprocess.run([CLI, f"--check-mask={g}"], capture_output=True, text=True).stdout


@pytest.mark.gpu
def test_cli_tsar_mode_is_the_reference_live_path(tmp_path):
    """--mode=tsar: external depth/normal .dmb + weak.png -> weak-texture regions -> region RANSAC -> plane fill ->
    TSAR_disp.dmb / TSAR_normals.dmb (runGipuma, main.cpp:1458-1860), against the same sequence through the library"""
    from tsar_mvs_amd import api
    sc = synth.make_scene(1216, 832, 2, seed=5, textureless=True, flat_cell=3.0)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    h, w = sc.h, sc.w
    rng = np.random.default_rng(3)
    textured = sc.textured.numpy()
    gt = sc.gt_depth.numpy()
    depth = (gt * (1 + rng.normal(0, 0.002, gt.shape))).astype(np.float32)
    junk = rng.uniform(sc.depth_min, sc.depth_max, gt.shape).astype(np.float32)
    good = textured | (rng.uniform(size=gt.shape) < 0.1)                      # the external matcher fails on most flat pixels
    depth[~good] = junk[~good]
    normal_world = np.ascontiguousarray((sc.gt_normal.numpy() @ sc.R[0]).astype(np.float32))   # R^T n_cam per pixel
    apd = root + "APD/00000000/"
    os.makedirs(apd, exist_ok=True)
    tio.write_dmb(apd + "depths_geom.dmb", depth)
    tio.write_dmb(apd + "normals.dmb", normal_world)
    tio.write_reliable_mask(apd + "weak.png", good)
    names = [f"{k:08d}.pgm" for k in (0, 1, 2)]
    cmd = [CLI, *names, "-mslp_folder", root, "-images_folder", root + "images/", "--blocksize=11", "--n_best=1", "--mode=tsar"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got_d = tio.read_dmb(apd + "TSAR_disp.dmb")
    got_n = tio.read_dmb(apd + "TSAR_normals.dmb")
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max))
    m.set_views(sc.images, sc.K, sc.R, sc.t)
    m.set_view_subset(tio.source_slots(0, [1, 2]))
    m.load_planes(depth, normal_world)
    m.set_reliable_mask(good.astype(np.float32))
    labels, text, size = m.detect_weak_texture()
    m.getview()
    m.ransac_regions()
    m.fake_depth()
    m.fill_textureless()
    res = m.get_result(("depth", "normal"))
    assert np.array_equal(got_d, res["depth"]) and np.array_equal(got_n, res["normal"])
    # and it did something: inside the detected weak regions the junk depth was replaced by a plane close to the truth
    weak = np.isin(labels, np.nonzero(text == -1)[0]) & ~good
    assert weak.sum() > 5000
    err_before = np.abs(depth[weak] - gt[weak]) / gt[weak]
    err_after = np.abs(res["depth"][weak] - gt[weak]) / gt[weak]
    assert np.median(err_after) < 0.02 < np.median(err_before)
    m.close()


@pytest.mark.gpu
def test_cli_all_views_tsar_mode_equals_one_process_per_view(tmp_path):
    """--all --mode=tsar (one process, inputs of the next views read ahead into a ring of page-locked buffers while a view is on
    the GPU, reference-only contexts) writes the same bytes as one invocation per view"""
    sc = synth.make_scene(608, 416, 4, seed=6, textureless=True, flat_cell=3.0, all_gt=True)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    rng = np.random.default_rng(4)
    n = len(sc.images)
    good = sc.textured.numpy() | (rng.uniform(size=(sc.h, sc.w)) < 0.1)
    for v in range(n):
        gt, nc = sc.meta["gt_all"][v]
        gt = gt.numpy()
        depth = (gt * (1 + rng.normal(0, 0.002, gt.shape))).astype(np.float32)
        junk = rng.uniform(sc.depth_min, sc.depth_max, gt.shape).astype(np.float32)
        depth[~good] = junk[~good]
        apd = root + f"APD/{v:08d}/"
        os.makedirs(apd, exist_ok=True)
        tio.write_dmb(apd + "depths_geom.dmb", depth)
        tio.write_dmb(apd + "normals.dmb", np.ascontiguousarray((nc.numpy() @ sc.R[v]).astype(np.float32)))
        tio.write_reliable_mask(apd + "weak.png", good)
    common = ["-mslp_folder", root, "-images_folder", root + "images/", "--blocksize=11", "--n_best=1", "--mode=tsar"]
    out = subprocess.run([CLI, "--all", "--gpus=1", *common], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    together = {v: (open(root + f"APD/{v:08d}/TSAR_disp.dmb", "rb").read(), open(root + f"APD/{v:08d}/TSAR_normals.dmb", "rb").read()) for v in range(n)}
    for v in range(n):
        os.remove(root + f"APD/{v:08d}/TSAR_disp.dmb")
        os.remove(root + f"APD/{v:08d}/TSAR_normals.dmb")
        names = [f"{v:08d}.pgm"] + [f"{s:08d}.pgm" for s in range(n) if s != v]     # the reference's argv: reference first
        one = subprocess.run([CLI, *names, *common], capture_output=True, text=True)
        assert one.returncode == 0, one.stdout + one.stderr
        assert open(root + f"APD/{v:08d}/TSAR_disp.dmb", "rb").read() == together[v][0]
        assert open(root + f"APD/{v:08d}/TSAR_normals.dmb", "rb").read() == together[v][1]
    # the refinement did run: the output differs from the external depth inside the unreliable pixels
    d0 = tio.read_dmb(root + "APD/00000000/TSAR_disp.dmb")
    ext0 = tio.read_dmb(root + "APD/00000000/depths_geom.dmb")
    assert (d0 != ext0).mean() > 0.01


@pytest.mark.gpu
def test_cli_color_processing_matches_on_the_blue_channel(tmp_path):
    """-color_processing: the reference uploads BGRA float4 textures but its cost fetches tex2D<float>, i.e. the first
    (blue) channel (gipuma.cu:247,262,265; main.cpp:1427-1447).  PPM in, same outputs as the gray run on that channel."""
    sc = synth.make_scene(128, 96, 3, seed=5)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    rng = np.random.default_rng(0)
    for k in range(4):
        blue = sc.images[k].numpy()
        rgb = np.stack([rng.integers(0, 256, blue.shape), rng.integers(0, 256, blue.shape), blue], -1)   # R, G are decoys
        tio.write_ppm(root + f"images/{k:08d}.ppm", rgb)
    names = [f"{k:08d}.ppm" for k in (1, 0, 2, 3)]
    common = ["-mslp_folder", root, "-images_folder", root + "images/", "--iterations=2", "--blocksize=11", "--n_best=1", "--seed=7"]
    out = subprocess.run([CLI, *names, *common, "-color_processing"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    d_col = tio.read_dmb(root + "APD/00000001/TSAR_disp.dmb")
    out = subprocess.run([CLI, *[n.replace(".ppm", ".pgm") for n in names], *common], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert np.array_equal(d_col, tio.read_dmb(root + "APD/00000001/TSAR_disp.dmb"))


@pytest.mark.gpu
def test_cli_display_outputs(tmp_path):
    """--display_outputs: TSAR_normals.png (16-bit, n * 32767 + 32767, displayUtils.h:239-245) and TSAR_model.ply (one
    vertex per pixel: world point, world normal, gray x3; displayUtils.h:78-150)"""
    import struct
    import zlib
    sc = synth.make_scene(96, 64, 2, seed=5)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    names = [f"{k:08d}.pgm" for k in (0, 1, 2)]
    out = subprocess.run([CLI, *names, "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=2", "--blocksize=11", "--n_best=1",
                          "--display_outputs"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    apd = root + "APD/00000000/"
    depth, normal = tio.read_dmb(apd + "TSAR_disp.dmb"), tio.read_dmb(apd + "TSAR_normals.dmb")
    h, w = depth.shape
    raw = open(apd + "TSAR_normals.png", "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xffffffff == struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    assert ihdr == (w, h, 16, 2, 0, 0, 0)
    px = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 6)
    assert (px[:, 0] == 0).all()
    vis = px[:, 1:].reshape(h, w, 3, 2).astype(np.int32)
    vis = vis[..., 0] * 256 + vis[..., 1]
    want = np.clip(np.rint(normal.astype(np.float32) * np.float32(32767) + np.float32(32767)), 0, 65535).astype(np.int32)
    assert np.abs(vis - want).max() <= 1
    ply = open(apd + "TSAR_model.ply", "rb").read()
    head, body = ply.split(b"end_header\n", 1)
    assert f"element vertex {w * h}".encode() in head and len(body) == w * h * 27
    rec = np.frombuffer(body, dtype=np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("c", "u1", 3)])).reshape(w, h)   # column by column
    K, R, t = sc.K[0].astype(np.float64), sc.R[0].astype(np.float64), sc.t[0].astype(np.float64)
    ys, xs = np.mgrid[0:h, 0:w]
    cam = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs, float)], -1) * depth[..., None] - t
    world = cam @ R                                            # R^T c per pixel
    assert np.allclose(rec["p"].transpose(1, 0, 2), world, atol=1e-4)
    assert np.array_equal(rec["n"].transpose(1, 0, 2), normal)
    assert np.array_equal(rec["c"][..., 0].T, sc.images[0].numpy().astype(np.uint8))


@pytest.mark.gpu
def test_cli_more_gpus_requested_than_present_requeues_onto_a_present_one(tmp_path):
    """--all --gpus=N with N beyond the devices of the box: the views dealt to the missing device fail with a message (no crash, no
    hang) and — round 5 — are retried on the next device's turn, which exists: every view's files are written, byte-identical to
    a run that asked for the devices that are there, and the exit status is 0.  (Before the re-queue the run ended non-zero with
    those views missing.)"""
    import torch
    n_dev = torch.cuda.device_count()
    sc = synth.make_scene(128, 96, 3, seed=5)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    common = ["-mslp_folder", root, "-images_folder", root + "images/", "--iterations=1", "--blocksize=11", "--n_best=1"]
    out = subprocess.run([CLI, "--all", f"--gpus={n_dev + 1}", *common], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "tsar_create" in out.stderr and f"FAILED on gpu {n_dev}: retrying once on gpu 0 with a fresh context" in out.stdout
    got = {v: open(root + f"APD/{v:08d}/TSAR_disp.dmb", "rb").read() for v in range(4)}
    ok = subprocess.run([CLI, "--all", f"--gpus={n_dev}", "--force", *common], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "retry" not in ok.stdout
    assert got == {v: open(root + f"APD/{v:08d}/TSAR_disp.dmb", "rb").read() for v in range(4)}


@pytest.mark.gpu
def test_cli_all_resumes_from_the_output_files(tmp_path):
    """--all skips a view whose TSAR_disp.dmb + TSAR_normals.dmb are complete (the reference's output contract, main.cpp:1817-1860: the
    per-view files are the checkpoints); a missing or truncated file, or a header of another size, is not 'done'; --force recomputes;
    --fuse after a resume reads the skipped views' maps back and gives the same cloud"""
    sc = synth.make_scene(160, 120, 4, seed=8)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    common = ["--all", "--gpus=1", "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=2", "--blocksize=11", "--n_best=1", "--seed=5"]
    first = subprocess.run([CLI, *common, "--fuse"], capture_output=True, text=True)
    assert first.returncode == 0 and "skipped" not in first.stdout, first.stdout + first.stderr
    files = {v: [root + f"APD/{v:08d}/TSAR_disp.dmb", root + f"APD/{v:08d}/TSAR_normals.dmb"] for v in range(5)}
    want = {v: [open(p, "rb").read() for p in ps] for v, ps in files.items()}
    cloud = open(root + "APD/APD_TSAR.ply", "rb").read()
    stamp = {p: os.stat(p).st_mtime_ns for ps in files.values() for p in ps}
    # everything present: nothing is matched, nothing is rewritten
    again = subprocess.run([CLI, *common], capture_output=True, text=True)
    assert again.returncode == 0 and again.stdout.count("outputs present, skipped") == 5 and "resuming: 5 of 5" in again.stdout, again.stdout
    assert all(os.stat(p).st_mtime_ns == t for p, t in stamp.items())
    # one view's file deleted, one truncated, one with the header of another size: exactly those three are matched again, to the same bytes
    os.remove(files[1][0])
    open(files[2][1], "wb").write(want[2][1][:-40])
    open(files[3][0], "wb").write(struct.pack("<iiii", 1, 119, 160, 1) + want[3][0][16:])
    third = subprocess.run([CLI, *common, "--fuse"], capture_output=True, text=True)
    assert third.returncode == 0 and third.stdout.count("outputs present, skipped") == 2 and "resuming: 2 of 5" in third.stdout, third.stdout + third.stderr
    for v in range(5):
        assert [open(p, "rb").read() for p in files[v]] == want[v], v
    assert all(os.stat(p).st_mtime_ns == stamp[p] for v in (0, 4) for p in files[v])
    assert open(root + "APD/APD_TSAR.ply", "rb").read() == cloud          # the skipped views' maps were read back for the fuser
    forced = subprocess.run([CLI, *common, "--force"], capture_output=True, text=True)
    assert forced.returncode == 0 and "skipped" not in forced.stdout
    assert all(os.stat(p).st_mtime_ns != t for p, t in stamp.items())
    for v in range(5):
        assert [open(p, "rb").read() for p in files[v]] == want[v], v


@pytest.mark.gpu
def test_cli_all_requeues_a_failed_view_once(tmp_path):
    """a view that fails on its worker is retried once with a fresh context (on the next GPU's turn; the same device on a one-GPU box)
    and the run still exits 0 with every file there; a view that fails twice leaves a non-zero exit status, the other views'
    files, and no partial file of its own.  The failure is injected (TSAR_GIPUMA_INJECT_FAILURE=<view>[:<times>]): the context of
    the failing attempt is dropped exactly as after a device-side error."""
    sc = synth.make_scene(160, 120, 3, seed=9)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    common = ["--all", "--gpus=1", "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=2", "--blocksize=11", "--n_best=1", "--seed=5"]
    clean = subprocess.run([CLI, *common], capture_output=True, text=True)
    assert clean.returncode == 0, clean.stdout + clean.stderr
    want = {v: open(root + f"APD/{v:08d}/TSAR_disp.dmb", "rb").read() for v in range(4)}
    env = dict(os.environ, TSAR_GIPUMA_INJECT_FAILURE="2")
    once = subprocess.run([CLI, *common, "--force"], capture_output=True, text=True, env=env)
    assert once.returncode == 0, once.stdout + once.stderr
    assert "injected failure" in once.stderr and "view 00000002 FAILED on gpu 0: retrying once on gpu 0 with a fresh context" in once.stdout
    assert "view 00000002 on gpu 0 (retry): ok" in once.stdout
    for v in range(4):
        assert open(root + f"APD/{v:08d}/TSAR_disp.dmb", "rb").read() == want[v], v      # the retry gives the same bytes; the views after the failure too
    for v in range(4):
        os.remove(root + f"APD/{v:08d}/TSAR_disp.dmb")
    env["TSAR_GIPUMA_INJECT_FAILURE"] = "1:2"
    twice = subprocess.run([CLI, *common], capture_output=True, text=True, env=env)
    assert twice.returncode != 0
    assert "view 00000001 on gpu 0 (retry): FAILED" in twice.stdout and "view 00000001: outputs missing or incomplete" in twice.stderr
    assert not os.path.exists(root + "APD/00000001/TSAR_disp.dmb")
    for v in (0, 2, 3):
        assert open(root + f"APD/{v:08d}/TSAR_disp.dmb", "rb").read() == want[v], v
    # and the next plain run completes exactly the missing view
    fix = subprocess.run([CLI, *common], capture_output=True, text=True)
    assert fix.returncode == 0 and fix.stdout.count("outputs present, skipped") == 3
    assert open(root + "APD/00000001/TSAR_disp.dmb", "rb").read() == want[1]


def test_cli_all_resume_decision_needs_no_gpu(tmp_path):
    """the resume half of --all on the CPU: with every view's output files complete the tool skips them all and exits 0 without ever
    creating a context (so it runs here, where there is no device); with one file truncated that view is attempted, cannot get a
    device, is retried once, and the exit status says a view is missing"""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the failing half of this test would match the view instead")
    sc = synth.make_scene(64, 48, 2, seed=2)
    root = str(tmp_path) + "/"
    tio.export_scene(sc, root)
    for v in range(3):
        os.makedirs(root + f"APD/{v:08d}", exist_ok=True)
        tio.write_dmb(root + f"APD/{v:08d}/TSAR_disp.dmb", np.ones((48, 64), np.float32))
        tio.write_dmb(root + f"APD/{v:08d}/TSAR_normals.dmb", np.zeros((48, 64, 3), np.float32))
    cmd = [CLI, "--all", "--gpus=1", "-mslp_folder", root, "-images_folder", root + "images/", "--iterations=1", "--blocksize=11", "--n_best=1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("outputs present, skipped") == 3 and "resuming: 3 of 3" in out.stdout
    raw = open(root + "APD/00000001/TSAR_normals.dmb", "rb").read()
    open(root + "APD/00000001/TSAR_normals.dmb", "wb").write(raw[:-4])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert out.stdout.count("outputs present, skipped") == 2 and "retrying once" in out.stdout
    assert "view 00000001: outputs missing or incomplete" in out.stderr
