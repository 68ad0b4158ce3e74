// tsar_io.h — the reference's on-disk formats for the C++ host tools (SURVEY §8b process-level contract):
// cams/%08d_cam.txt (fileIoUtils.h:117-153), pair.txt (main.cpp:1351-1376), .dmb (fileIoUtils.h:260-381), binary PGM.
#pragma once
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <map>
#include <string>
#include <vector>
#include <zlib.h>

#include "../../include/tsar.h"

struct CamFile {
    tsar_camera cam;
    float depth_min, depth_max;
};

static inline bool read_cam(const std::string& path, CamFile& out) {   // fileIoUtils.h:117-153
    std::ifstream f(path);
    if (!f) return false;
    std::string word;
    f >> word;   // "extrinsic"
    for (int r = 0; r < 3; r++) f >> out.cam.R[3 * r] >> out.cam.R[3 * r + 1] >> out.cam.R[3 * r + 2] >> out.cam.t[r];
    float tmp;
    f >> tmp >> tmp >> tmp >> tmp;
    f >> word;   // "intrinsic"
    for (int r = 0; r < 3; r++) f >> out.cam.K[3 * r] >> out.cam.K[3 * r + 1] >> out.cam.K[3 * r + 2];
    float interval, num;
    f >> out.depth_min >> interval >> num >> out.depth_max;
    return !f.fail();
}

static inline bool read_pairs(const std::string& path, std::map<int, std::vector<int>>& pairs) {   // main.cpp:1351-1376
    std::ifstream f(path);
    if (!f) return false;
    int n = 0;
    f >> n;
    for (int i = 0; i < n; i++) {
        int ref, k;
        f >> ref >> k;
        std::vector<int> src(k);
        for (int j = 0; j < k; j++) { float score; f >> src[j] >> score; }
        if (f.fail()) return false;
        pairs[ref] = src;
    }
    return true;
}

template <class FloatVec>
static inline bool read_pgm(const std::string& path, FloatVec& img, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int maxv = 0, got = 0, vals[3];
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P5") != 0) { fclose(f); return false; }
    while (got < 3) {
        int c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == EOF) { fclose(f); return false; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        ungetc(c, f);
        if (fscanf(f, "%d", &vals[got]) != 1) { fclose(f); return false; }
        got++;
    }
    fgetc(f);   // single whitespace after maxval
    w = vals[0]; h = vals[1]; maxv = vals[2];
    if (maxv > 255 || w <= 0 || h <= 0) { fclose(f); return false; }
    std::vector<unsigned char> raw((size_t)w * h);
    const bool ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
    fclose(f);
    if (!ok) return false;
    img.resize(raw.size());
    for (size_t i = 0; i < raw.size(); i++) img[i] = (float)raw[i];   // convertTo(CV_32FC1), main.cpp:1423
    return true;
}

// Binary PPM (P6), one channel extracted.  -color_processing in the reference uploads float4 (B, G, R, alpha) textures
// but the matching cost fetches them with tex2D<float> (gipuma.cu:247,262,265), i.e. it matches on the first channel of
// OpenCV's BGR order: blue.  channel: 0 = R, 1 = G, 2 = B of the PPM.
// a binary PGM as it is on disk: w*h bytes, read straight into the vector (no float copy: the library widens on the device,
// tsar_set_views_u8)
static inline bool read_pgm_u8(const std::string& path, std::vector<uint8_t>& img, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int got = 0, vals[3];
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P5") != 0) { fclose(f); return false; }
    while (got < 3) {
        int c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == EOF) { fclose(f); return false; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        ungetc(c, f);
        if (fscanf(f, "%d", &vals[got]) != 1) { fclose(f); return false; }
        got++;
    }
    fgetc(f);   // single whitespace after maxval
    w = vals[0]; h = vals[1];
    if (vals[2] > 255 || w <= 0 || h <= 0) { fclose(f); return false; }
    img.resize((size_t)w * h);
    const bool ok = fread(img.data(), 1, img.size(), f) == img.size();
    fclose(f);
    return ok;
}

// size of a binary PGM / PPM from its header alone (a resumed --all run checks finished views without decoding anything)
static inline bool pnm_size(const std::string& path, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int got = 0, vals[2] = {0, 0};
    bool ok = fscanf(f, "%2s", magic) == 1 && (strcmp(magic, "P5") == 0 || strcmp(magic, "P6") == 0);
    while (ok && got < 2) {
        int c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == EOF) { ok = false; break; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        ungetc(c, f);
        if (fscanf(f, "%d", &vals[got]) != 1) { ok = false; break; }
        got++;
    }
    fclose(f);
    w = vals[0]; h = vals[1];
    return ok && w > 0 && h > 0;
}

template <class FloatVec>
static inline bool read_ppm_channel(const std::string& path, int channel, FloatVec& img, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int got = 0, vals[3];
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P6") != 0) { fclose(f); return false; }
    while (got < 3) {
        int c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == EOF) { fclose(f); return false; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        ungetc(c, f);
        if (fscanf(f, "%d", &vals[got]) != 1) { fclose(f); return false; }
        got++;
    }
    fgetc(f);
    w = vals[0]; h = vals[1];
    if (vals[2] > 255 || w <= 0 || h <= 0) { fclose(f); return false; }
    std::vector<unsigned char> raw((size_t)w * h * 3);
    const bool ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
    fclose(f);
    if (!ok) return false;
    img.resize((size_t)w * h);
    for (size_t i = 0; i < img.size(); i++) img[i] = (float)raw[i * 3 + channel];
    return true;
}

static inline bool write_dmb(const std::string& path, const float* data, int h, int w, int nb) {   // fileIoUtils.h:333-381
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "Error opening file %s\n", path.c_str()); return false; }
    const int32_t hdr[4] = {1, h, w, nb};
    bool ok = fwrite(hdr, sizeof(int32_t), 4, f) == 4;
    ok = ok && fwrite(data, sizeof(float), (size_t)h * w * nb, f) == (size_t)h * w * nb;
    fclose(f);
    return ok;
}
// a .dmb that is all there: the reference's header (type 1, h, w, nb: fileIoUtils.h:333-381) and exactly h * w * nb floats behind it —
// the done marker of a view (the per-view output files are the coarse checkpoints)
static inline bool dmb_complete(const std::string& path, int h, int w, int nb) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    int32_t hd[4] = {0, 0, 0, 0};
    bool ok = fread(hd, 4, 4, f) == 4 && hd[0] == 1 && hd[1] == h && hd[2] == w && hd[3] == nb;
    if (ok) ok = fseek(f, 0, SEEK_END) == 0 && ftell(f) == (long)(16 + (size_t)h * w * nb * 4);
    fclose(f);
    return ok;
}

template <class FloatVector>   // std::vector<float> with any allocator
static inline bool read_dmb(const std::string& path, FloatVector& data, int& h, int& w, int& nb) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    int32_t hdr[4];
    if (fread(hdr, sizeof(int32_t), 4, f) != 4 || hdr[0] != 1) { fclose(f); return false; }
    h = hdr[1]; w = hdr[2]; nb = hdr[3];
    data.resize((size_t)h * w * nb);
    const bool ok = fread(data.data(), sizeof(float), data.size(), f) == data.size();
    fclose(f);
    return ok;
}

// Minimal PNG reader (8-bit gray / RGB / RGBA / palette, non-interlaced) on zlib: the reliability mask of the
// reference's live path is APD/<id>/weak.png, which it reads with cv::imread(IMREAD_COLOR) (main.cpp:1499-1514).
// Output: interleaved RGB, 3 bytes per pixel.
static inline bool read_png_rgb(const std::string& path, std::vector<unsigned char>& rgb, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | (uint32_t)file[o + 3]; };
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat, plte;
    w = h = 0;
    for (size_t o = 8; o + 12 <= file.size();) {
        const uint32_t len = be32(o);
        if (o + 12 + (size_t)len > file.size()) return false;
        const char* type = (const char*)&file[o + 4];
        const unsigned char* data = &file[o + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = (int)be32(o + 8); h = (int)be32(o + 12);
            depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        o += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) return false;
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
    std::vector<unsigned char> img(stride * (size_t)h);
    for (int y = 0; y < h; y++) {                       // undo the per-row filters (PNG spec 9.2)
        const unsigned char ft = raw[(stride + 1) * y];
        const unsigned char* in = &raw[(stride + 1) * y + 1];
        unsigned char* out = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? out[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) return false;
            out[i] = (unsigned char)(in[i] + pred);
        }
    }
    rgb.resize((size_t)w * h * 3);
    for (size_t p = 0; p < (size_t)w * h; p++) {
        const unsigned char* s = &img[p * ch];
        unsigned char r, g, b;
        if (ctype == 3) { const size_t k = (size_t)s[0] * 3; if (k + 2 >= plte.size()) return false; r = plte[k]; g = plte[k + 1]; b = plte[k + 2]; }
        else if (ch <= 2) { r = g = b = s[0]; }
        else { r = s[0]; g = s[1]; b = s[2]; }
        rgb[p * 3] = r; rgb[p * 3 + 1] = g; rgb[p * 3 + 2] = b;
    }
    return true;
}

// 16-bit RGB PNG writer (zlib), for TSAR_normals.png: the reference shows world normals as n * 32767 + 32767 per channel
// (getNormalsForDisplay, displayUtils.h:239-245, then imwrite of the BGR-swapped image: x -> R, y -> G, z -> B in the file).
static inline bool write_png_rgb16(const std::string& path, const uint16_t* rgb, int w, int h) {
    std::vector<unsigned char> raw((size_t)h * (1 + (size_t)w * 6));
    for (int y = 0; y < h; y++) {
        unsigned char* row = &raw[(size_t)y * (1 + (size_t)w * 6)];
        row[0] = 0;                                          // filter type none
        for (int k = 0; k < w * 3; k++) { const uint16_t v = rgb[(size_t)y * w * 3 + k]; row[1 + 2 * k] = (unsigned char)(v >> 8); row[2 + 2 * k] = (unsigned char)(v & 255); }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 1) != Z_OK) return false;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    auto chunk = [&](const char* tag, const unsigned char* data, uint32_t len) {
        unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len, (unsigned char)tag[0], (unsigned char)tag[1], (unsigned char)tag[2], (unsigned char)tag[3]};
        fwrite(hdr, 1, 8, f);
        if (len) fwrite(data, 1, len, f);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (len) crc = crc32(crc, data, len);
        const unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
        fwrite(c, 1, 4, f);
    };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, f);
    const unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w, (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h, 16, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    return fclose(f) == 0;
}

// TSAR_model.ply: one vertex per pixel — world point of the depth, world normal, gray three times (storePlyFileBinary,
// displayUtils.h:78-150; X = R^T (depth K^-1 (x, y, 1) - t), cameraGeometryUtils.h:53-65; non-finite points become 0).
// The reference writes the vertices from an OpenMP loop in arbitrary order; here column by column as its loop nest reads.
static inline bool write_view_ply(const std::string& path, const float* depth, const float* normal, const float* gray, int w, int h, const tsar_camera& cam) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
               "property float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n", w * h);
    const double fx = cam.K[0], sk = cam.K[1], cx = cam.K[2], fy = cam.K[4], cy = cam.K[5];
    std::vector<unsigned char> col((size_t)h * 27);
    for (int x = 0; x < w; x++) {
        for (int y = 0; y < h; y++) {
            const size_t p = (size_t)y * w + x;
            const double d = depth[p];
            const double yc = (y - cy) / fy, xc = (x - cx - sk * yc) / fx;           // K^-1 (x, y, 1), K upper triangular
            const double c[3] = {d * xc - cam.t[0], d * yc - cam.t[1], d - cam.t[2]};
            float rec[6];
            for (int k = 0; k < 3; k++) rec[k] = (float)(cam.R[k] * c[0] + cam.R[3 + k] * c[1] + cam.R[6 + k] * c[2]);   // R^T c
            if (!(rec[0] < FLT_MAX && rec[0] > -FLT_MAX && rec[1] < FLT_MAX && rec[1] > -FLT_MAX && rec[2] < FLT_MAX && rec[2] > -FLT_MAX)) rec[0] = rec[1] = rec[2] = 0.f;
            rec[3] = normal[p * 3]; rec[4] = normal[p * 3 + 1]; rec[5] = normal[p * 3 + 2];
            unsigned char* o = &col[(size_t)y * 27];
            memcpy(o, rec, 24);
            const float g = gray[p];
            o[24] = o[25] = o[26] = (unsigned char)(g < 0.f ? 0 : g > 255.f ? 255 : (int)g);
        }
        if (fwrite(col.data(), 1, col.size(), f) != col.size()) { fclose(f); return false; }
    }
    return fclose(f) == 0;
}

// weak.png -> lines->scale: white, pure green and pure red pixels are reliable (main.cpp:1503-1513; scale starts at 0)
static inline bool read_reliable_mask(const std::string& path, std::vector<float>& scale, int& w, int& h) {
    std::vector<unsigned char> rgb;
    if (!read_png_rgb(path, rgb, w, h)) return false;
    scale.assign((size_t)w * h, 0.0f);
    for (size_t p = 0; p < scale.size(); p++) {
        const unsigned char r = rgb[p * 3], g = rgb[p * 3 + 1], b = rgb[p * 3 + 2];
        if ((r == 255 && g == 255 && b == 255) || (r == 0 && g == 255 && b == 0) || (r == 255 && g == 0 && b == 0)) scale[p] = 1.0f;
    }
    return true;
}

// fused cloud, binary little-endian: x y z nx ny nz red green blue per point (records of 9 floats from tsar_fuse)
static bool write_cloud_ply(const std::string& path, const float* pts, int64_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %lld\nproperty float x\nproperty float y\nproperty float z\n"
               "property float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n", (long long)n);
    // 27-byte records assembled a megapoint at a time: one fwrite per chunk instead of two per point
    const int64_t chunk = 1 << 20;
    std::vector<unsigned char> rec((size_t)chunk * 27);
    bool ok = true;
    for (int64_t i0 = 0; i0 < n && ok; i0 += chunk) {
        const int64_t m = n - i0 < chunk ? n - i0 : chunk;
        for (int64_t i = 0; i < m; i++) {
            const float* p = pts + 9 * (i0 + i);
            unsigned char* o = rec.data() + 27 * i;
            memcpy(o, p, 24);
            const float g = p[6] < 0 ? 0 : (p[6] > 255 ? 255 : p[6]);
            o[24] = o[25] = o[26] = (unsigned char)(g + 0.5f);
        }
        ok = fwrite(rec.data(), 27, (size_t)m, f) == (size_t)m;
    }
    fclose(f);
    return ok;
}

