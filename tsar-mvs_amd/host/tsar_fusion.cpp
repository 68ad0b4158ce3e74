// tsar_fusion — C++ host tool with the command line of the reference's fuser
// (x/1.sh:30:  Fusion <mslp_dir> --num_consistent= 1 --reproj_error= 2 --depth_diff= 0.01 --angle= 15 --used_list= 1 ;
// note the blank after '=' in the reference's scripts — both spellings are accepted).
// Reads <dir>/pair.txt, <dir>/cams/%08d_cam.txt, <dir>/images/%08d.pgm (else the .jpg of that name) and, per view,
// <dir>/APD/%08d/TSAR_disp.dmb + TSAR_normals.dmb (what tsar_gipuma / the reference write), fuses them on
// the GPU (tsar_fuse) and writes <dir>/APD/APD_TSAR.ply (binary little-endian: x y z nx ny nz red green blue).
#include <stdlib.h>

#include <future>
#include <memory>

#include "tsar_io.h"
#include "tsar_jpeg.h"

int main(int argc, char** argv) {
    if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) {
        printf("usage: tsar_fusion <mslp_dir> [--num_consistent= N] [--reproj_error= PX] [--depth_diff= REL] [--angle= DEG] [--used_list= 0|1] [--gpu=K]\n");
        return argc < 2 ? 1 : 0;
    }
    std::string dir = argv[1];
    if (dir.back() != '/') dir += '/';
    tsar_fusion_params prm;
    tsar_default_fusion_params(&prm);
    int gpu = 0;
    for (int i = 2; i < argc; i++) {
        const char* a = argv[i];
        auto value = [&](const char* opt) -> const char* {      // "--opt=VALUE" or "--opt= VALUE"
            const size_t n = strlen(opt);
            if (strncmp(a, opt, n) != 0) return nullptr;
            if (a[n] != '\0') return a + n;
            return i + 1 < argc ? argv[++i] : "";
        };
        const char* v;
        if ((v = value("--num_consistent="))) prm.num_consistent = atoi(v);
        else if ((v = value("--reproj_error="))) prm.reproj_error = (float)atof(v);
        else if ((v = value("--depth_diff="))) prm.depth_diff = (float)atof(v);
        else if ((v = value("--angle="))) prm.angle_deg = (float)atof(v);
        else if ((v = value("--used_list="))) prm.used_list = atoi(v);
        else if ((v = value("--gpu="))) gpu = atoi(v);
        else printf("Command-line parameter warning: unknown option %s\n", a);
    }
    printf("num_consistent: %d\nreproj_error: %g\ndepth_diff: %g\nangle: %g\nused_list: %d\n", prm.num_consistent, prm.reproj_error, prm.depth_diff, prm.angle_deg, prm.used_list);
    std::map<int, std::vector<int>> pairs;
    if (!read_pairs(dir + "pair.txt", pairs)) { fprintf(stderr, "cannot read %spair.txt\n", dir.c_str()); return 1; }
    std::vector<int> ids;
    for (auto& kv : pairs) ids.push_back(kv.first);
    std::map<int, int> slot;
    for (size_t k = 0; k < ids.size(); k++) slot[ids[k]] = (int)k;
    const int n = (int)ids.size();
    std::vector<tsar_camera> cams(n);
    std::vector<std::vector<float>> depth(n), normal(n), gray(n);
    std::vector<const float*> pd(n), pn(n), pg(n);
    int w = 0, h = 0;
    // every view's three files are read by its own helper thread (0.39 GB of maps per full-size view)
    std::vector<int> vw(n, 0), vh(n, 0);
    std::vector<std::string> problem(n);
    auto load = [&](int k) {
        char name[32];
        snprintf(name, sizeof name, "%08d", ids[k]);
        CamFile cf;
        if (!read_cam(dir + "cams/" + name + "_cam.txt", cf)) { problem[k] = std::string("cannot read camera of view ") + name; return; }
        cams[k] = cf.cam;
        int hh, ww, nb;
        if (!read_dmb(dir + "APD/" + name + "/TSAR_disp.dmb", depth[k], hh, ww, nb) || nb != 1) { problem[k] = std::string("cannot read APD/") + name + "/TSAR_disp.dmb"; return; }
        vw[k] = ww; vh[k] = hh;
        int h2, w2;
        if (!read_dmb(dir + "APD/" + name + "/TSAR_normals.dmb", normal[k], h2, w2, nb) || nb != 3 || w2 != ww || h2 != hh) { problem[k] = std::string("cannot read APD/") + name + "/TSAR_normals.dmb"; return; }
        int iw, ih;
        bool have = read_pgm(dir + "images/" + name + ".pgm", gray[k], iw, ih);
        for (const char* ext : {".jpg", ".JPG", ".jpeg", ".JPEG"}) {          // the scene's own JPEG (host/tsar_jpeg.h), like tsar_gipuma
            if (have) break;
            std::vector<uint8_t> px;
            if (tsar_jpeg::read(dir + "images/" + name + ext, tsar_jpeg::LUMA, px, iw, ih)) { gray[k].assign(px.begin(), px.end()); have = true; }
        }
        if (!have || iw != ww || ih != hh) gray[k].assign((size_t)ww * hh, 128.f);   // colour is cosmetic
        pd[k] = depth[k].data(); pn[k] = normal[k].data(); pg[k] = gray[k].data();
    };
    {
        const int par = 8;                                   // views in flight
        for (int k0 = 0; k0 < n; k0 += par) {
            std::vector<std::future<void>> jobs;
            for (int k = k0; k < n && k < k0 + par; k++) jobs.push_back(std::async(std::launch::async, load, k));
            for (auto& j : jobs) j.get();
        }
    }
    for (int k = 0; k < n; k++) {
        if (!problem[k].empty()) { fprintf(stderr, "%s\n", problem[k].c_str()); return 1; }
        if (k == 0) { w = vw[0]; h = vh[0]; }
        if (vw[k] != w || vh[k] != h) { fprintf(stderr, "view %08d has a different size\n", ids[k]); return 1; }
    }
    std::vector<int32_t> off(n + 1, 0), idx;
    for (int k = 0; k < n; k++) {
        for (int s : pairs[ids[k]])
            if (slot.count(s)) idx.push_back(slot[s]);
        off[k + 1] = (int32_t)idx.size();
    }
    if (idx.empty()) idx.push_back(0);
    const int64_t cap = (int64_t)n * w * h;
    std::unique_ptr<float[]> pts(new float[(size_t)cap * 9]);   // not zero-filled: only the fused points' pages are ever touched
    int64_t cnt = 0;
    const int rc = tsar_fuse(gpu, n, w, h, cams.data(), pd.data(), pn.data(), pg.data(), TSAR_MEM_HOST, off.data(), idx.data(), &prm, pts.get(), cap, &cnt);
    if (rc != TSAR_OK) { fprintf(stderr, "tsar_fuse failed: %d\n", rc); return 1; }
    if (cnt > cap) cnt = cap;
    const std::string out = dir + "APD/APD_TSAR.ply";
    if (!write_cloud_ply(out, pts.get(), cnt)) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    printf("%lld points -> %s\n", (long long)cnt, out.c_str());
    return 0;
}
