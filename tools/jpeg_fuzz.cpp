// tools/jpeg_fuzz.cpp — host/tsar_jpeg.h under AddressSanitizer + UndefinedBehaviorSanitizer on damaged files (tools/sanitize_cpu.sh):
// every file given is decoded as it is, then N times with random damage (byte flips, overwritten runs, truncations, spliced
// segments); each decode must return — an image or a refusal — without touching memory it does not own.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -o jpeg_fuzz tools/jpeg_fuzz.cpp && ./jpeg_fuzz 2000 a.jpg b.jpg ...
#include <stdlib.h>

#include <random>

#include "../tsar-mvs_amd/host/tsar_jpeg.h"

int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: jpeg_fuzz N file.jpg...\n"); return 2; }
    const int n = atoi(argv[1]);
    std::mt19937_64 rng(12345);
    size_t decoded = 0, refused = 0;
    const std::string tmp = std::string(getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp") + "/jpeg_fuzz_case.jpg";
    for (int a = 2; a < argc; a++) {
        tsar_jpeg::Decoder d;
        if (!d.load(argv[a])) { printf("cannot load %s\n", argv[a]); return 2; }
        const std::vector<uint8_t> clean = d.file;
        for (int it = 0; it <= n; it++) {
            std::vector<uint8_t> f = clean;
            if (it > 0) {
                const int kind = (int)(rng() % 5);
                if (kind == 0) for (int k = 0, m = 1 + (int)(rng() % 8); k < m; k++) f[rng() % f.size()] ^= (uint8_t)(1u << (rng() % 8));
                else if (kind == 1) { size_t at = rng() % f.size(), len = 1 + rng() % 64; for (size_t k = at; k < f.size() && k < at + len; k++) f[k] = (uint8_t)rng(); }
                else if (kind == 2) f.resize(2 + rng() % (f.size() - 2));
                else if (kind == 3) { size_t at = rng() % f.size(), len = rng() % 512; f.erase(f.begin() + at, f.begin() + std::min(f.size(), at + len)); }
                else { size_t at = rng() % std::min<size_t>(f.size(), 700); f[at] = (uint8_t)rng(); if (at + 1 < f.size()) f[at + 1] = (uint8_t)rng(); }   // the headers
            }
            FILE* o = fopen(tmp.c_str(), "wb");
            if (!o || fwrite(f.data(), 1, f.size(), o) != f.size()) { printf("cannot write %s\n", tmp.c_str()); return 2; }
            fclose(o);
            for (int want = 0; want < 2; want++) {
                std::vector<uint8_t> px;
                int w = 0, h = 0;
                std::string why;
                if (tsar_jpeg::read(tmp, want ? tsar_jpeg::BLUE : tsar_jpeg::LUMA, px, w, h, &why)) {
                    if (px.size() != (size_t)w * h) { printf("size mismatch\n"); return 1; }
                    decoded++;
                } else refused++;
            }
            int sw = 0, sh = 0;
            tsar_jpeg::size(tmp, sw, sh);
        }
    }
    remove(tmp.c_str());
    printf("jpeg_fuzz: %zu decoded, %zu refused, no fault\n", decoded, refused);
    return 0;
}
