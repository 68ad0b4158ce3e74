"""File formats of the reference's process-level contract — what its shell loop
feeds `gipuma` and what `Fusion.exe` reads back:

  cams/%08d_cam.txt   MVSNet style: "extrinsic" 4x4 world->camera, "intrinsic" 3x3,
                      "depth_min interval depth_num depth_max"   (reference fileIoUtils.h:111-163)
  pair.txt            MVSNet style view graph                      (reference main.cpp:1351-1376)
  *.dmb               int32 type(=1 float), int32 h, w, channels, then h*w*c float32 row-major
                                                                   (reference fileIoUtils.h:333-381)
  images              the reference decodes JPEG with OpenCV (absent here); the C++ CLI reads binary
                      PGM (P5), `convert_image` turns anything PIL can open into that.
"""
from __future__ import annotations

import os
import struct

import numpy as np


# ---- .dmb ---------------------------------------------------------------------------------------------
def write_dmb(path: str, arr) -> None:
    a = np.ascontiguousarray(arr, dtype=np.float32)
    if a.ndim == 2:
        h, w, c = a.shape[0], a.shape[1], 1
    elif a.ndim == 3:
        h, w, c = a.shape
    else:
        raise ValueError("dmb holds [h][w] or [h][w][c] float maps")
    with open(path, "wb") as f:
        f.write(struct.pack("<iiii", 1, h, w, c))
        f.write(a.tobytes())


def read_dmb(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        type_, h, w, c = struct.unpack("<iiii", f.read(16))
        if type_ != 1:
            raise ValueError(f"{path}: only float dmb (type 1) is supported, got type {type_}")   # fileIoUtils.h:283-286
        data = np.frombuffer(f.read(4 * h * w * c), dtype="<f4")
    if data.size != h * w * c:
        raise ValueError(f"{path}: truncated")
    return data.reshape((h, w) if c == 1 else (h, w, c)).copy()


# ---- cams/%08d_cam.txt --------------------------------------------------------------------------------
def write_cam(path: str, K, R, t, depth_min: float, depth_max: float, depth_num: int = 192) -> None:
    K = np.asarray(K, np.float64).reshape(3, 3)
    R = np.asarray(R, np.float64).reshape(3, 3)
    t = np.asarray(t, np.float64).reshape(3)
    interval = (depth_max - depth_min) / max(depth_num - 1, 1)
    with open(path, "w") as f:
        f.write("extrinsic\n")
        for r in range(3):
            f.write(" ".join(repr(float(v)) for v in (*R[r], t[r])) + "\n")
        f.write("0.0 0.0 0.0 1.0\n\nintrinsic\n")
        for r in range(3):
            f.write(" ".join(repr(float(v)) for v in K[r]) + "\n")
        f.write(f"\n{depth_min!r} {interval!r} {depth_num} {depth_max!r}\n")


def read_cam(path: str):
    """-> K[3,3], R[3,3], t[3], depth_min, depth_max (token order as the reference parses it)"""
    tok = open(path).read().split()
    if tok[0] != "extrinsic":
        raise ValueError(f"{path}: expected 'extrinsic'")
    v = [float(x) for x in tok[1:17]]
    E = np.array(v, np.float64).reshape(4, 4)
    if tok[17] != "intrinsic":
        raise ValueError(f"{path}: expected 'intrinsic'")
    K = np.array([float(x) for x in tok[18:27]], np.float64).reshape(3, 3)
    depth_min, _interval, _num, depth_max = (float(x) for x in tok[27:31])
    return K.astype(np.float32), E[:3, :3].astype(np.float32), E[:3, 3].astype(np.float32), float(depth_min), float(depth_max)


# ---- pair.txt -----------------------------------------------------------------------------------------
def write_pairs(path: str, pairs: dict) -> None:
    """pairs: {ref_id: [(src_id, score), ...]}"""
    with open(path, "w") as f:
        f.write(f"{len(pairs)}\n")
        for ref in sorted(pairs):
            f.write(f"{ref}\n")
            f.write(f"{len(pairs[ref])} " + " ".join(f"{s} {sc}" for s, sc in pairs[ref]) + "\n")


def read_pairs(path: str) -> dict:
    lines = [ln.strip() for ln in open(path).read().splitlines() if ln.strip()]
    n = int(lines[0])
    out = {}
    for i in range(n):
        ref = int(lines[1 + 2 * i])
        tok = lines[2 + 2 * i].split()
        k = int(tok[0])
        out[ref] = [(int(tok[1 + 2 * j]), float(tok[2 + 2 * j])) for j in range(k)]
    return out


def source_slots(ref_id: int, src_ids) -> list:
    """Index of each source view in the reference's argv image list (reference first, then every other
    image in order): id if id > ref else id + 1 (reference main.cpp:1371-1375)."""
    return [s if s > ref_id else s + 1 for s in src_ids]


# ---- images -------------------------------------------------------------------------------------------
def write_pgm(path: str, gray) -> None:
    a = np.asarray(gray)
    a8 = np.clip(np.rint(a), 0, 255).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (a8.shape[1], a8.shape[0]))
        f.write(a8.tobytes())


def write_ppm(path: str, rgb) -> None:
    """binary PPM (P6) for `tsar_gipuma -color_processing`, which like the reference matches on the BLUE channel
    (tex2D<float> on a BGRA float4 texture, gipuma.cu:247,262,265)"""
    a8 = np.clip(np.rint(np.asarray(rgb)), 0, 255).astype(np.uint8)
    assert a8.ndim == 3 and a8.shape[2] == 3
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (a8.shape[1], a8.shape[0]))
        f.write(a8.tobytes())


def read_pgm(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"P5":
        raise ValueError(f"{path}: not a binary PGM")
    # header: P5 <ws> w <ws> h <ws> maxval <single ws> data ; '#' comments allowed
    pos, vals = 2, []
    while len(vals) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        vals.append(int(data[pos:end]))
        pos = end
    pos += 1
    w, h, mx = vals
    if mx > 255:
        raise ValueError("only 8-bit PGM is supported")
    return np.frombuffer(data, np.uint8, count=w * h, offset=pos).reshape(h, w).astype(np.float32)


def write_png(path: str, rgb, filters=None) -> None:
    """8-bit PNG writer (gray [h, w] or RGB [h, w, 3]) on zlib, for the reliability mask APD/<id>/weak.png of the
    reference's live path (main.cpp:1499).  filters: optional per-row PNG filter types 0..4 (tests of the C++ reader)."""
    import struct
    import zlib
    a = np.ascontiguousarray(rgb, np.uint8)
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    rows = a.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft = 0 if filters is None else int(filters[y % len(filters)])
        cur = rows[y]
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        ul = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            pred = np.zeros_like(cur)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) // 2
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        raw.append(ft)
        raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def write_reliable_mask(path: str, reliable) -> None:
    """weak.png as the reference reads it: white = reliable, black = not (main.cpp:1503-1513)"""
    m = np.asarray(reliable).astype(bool)
    rgb = np.zeros(m.shape + (3,), np.uint8)
    rgb[m] = 255
    write_png(path, rgb)


def convert_image(src: str, dst: str) -> None:
    """Decode any PIL-readable image -> PGM (8-bit gray like cv::IMREAD_GRAYSCALE, main.cpp:1302) or, when dst ends in .ppm,
    -> PPM (RGB, for -color_processing).  A JPEG's gray is libjpeg's own grayscale output — the luminance component as decoded,
    which is what OpenCV's imread returns — not a conversion of its RGB decode (a few levels apart in coloured regions); the
    C++ tools read a scene's JPEGs directly with the same result bit for bit (host/tsar_jpeg.h, tests/test_jpeg_decode.py)."""
    from PIL import Image
    im = Image.open(src)
    if dst.lower().endswith(".ppm"):
        write_ppm(dst, np.asarray(im.convert("RGB"), np.float32))
        return
    if im.format == "JPEG" and im.mode in ("RGB", "YCbCr", "L"):
        im.draft("L", im.size)                     # libjpeg: out_color_space = JCS_GRAYSCALE, full size
    write_pgm(dst, np.asarray(im.convert("L"), np.float32))


# ---- a whole synthetic scene on disk, laid out like data/TRAIN/<scene>/ ---------------------------------
def export_scene(scene, root: str, ref_ids=None) -> None:
    """Write `scene` (tsar_mvs_amd.synth.Scene; view k gets image id k) as images/%08d.pgm,
    cams/%08d_cam.txt and pair.txt (every view paired with all others)."""
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "cams"), exist_ok=True)
    n = len(scene.images)
    for k in range(n):
        write_pgm(os.path.join(root, "images", f"{k:08d}.pgm"), scene.images[k].cpu().numpy())
        write_cam(os.path.join(root, "cams", f"{k:08d}_cam.txt"), scene.K[k], scene.R[k], scene.t[k], scene.depth_min, scene.depth_max)
    refs = range(n) if ref_ids is None else ref_ids
    write_pairs(os.path.join(root, "pair.txt"), {r: [(s, 1.0) for s in range(n) if s != r] for r in refs})


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 4 and sys.argv[1] == "convert":
        convert_image(sys.argv[2], sys.argv[3])
    else:
        print("usage: python -m tsar_mvs_amd.io convert <image> <out.pgm | out.ppm>")
