// tsar_io.h — the reference's on-disk formats for the C++ host tools (SURVEY §8b process-level contract):
// cams/%08d_cam.txt (fileIoUtils.h:117-153), pair.txt (main.cpp:1351-1376), .dmb (fileIoUtils.h:260-381), binary PGM.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <map>
#include <string>
#include <vector>

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

static inline bool read_pgm(const std::string& path, std::vector<float>& img, int& w, int& h) {
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

static inline bool write_dmb(const std::string& path, const float* data, int h, int w, int nb) {   // fileIoUtils.h:333-381
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "Error opening file %s\n", path.c_str()); return false; }
    const int32_t hdr[4] = {1, h, w, nb};
    bool ok = fwrite(hdr, sizeof(int32_t), 4, f) == 4;
    ok = ok && fwrite(data, sizeof(float), (size_t)h * w * nb, f) == (size_t)h * w * nb;
    fclose(f);
    return ok;
}
static inline bool read_dmb(const std::string& path, std::vector<float>& data, int& h, int& w, int& nb) {
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

