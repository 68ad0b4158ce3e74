// tsar_gipuma — C++ host driver above the C ABI (include/tsar.h), keeping the reference's
// process-level contract (reference main.cpp:708-1009 flags, :1351-1376 pair.txt,
// fileIoUtils.h:111-163 cam files, :333-381 .dmb) so that the per-view shell loop
// (reference scripts/courtyard.sh:29-48) and the fuser (x/1.sh:30) keep working:
//
//   tsar_gipuma <ref.pgm> <src.pgm...> -images_folder D/images/ -mslp_folder D/ -krt_file X
//               -output_folder O --cam_scale=1 --iterations=8 --blocksize=11 --cost_comb=best_n --n_best=1
//
// writes D/APD/<id>/TSAR_disp.dmb (depth) and TSAR_normals.dmb (world normals), the two files
// Fusion reads.  Beyond the reference:
//   --all [--gpus=N]   process every reference view of pair.txt, dealt round-robin to N GPUs, one host
//                      thread + one tsar_ctx per GPU (replaces the shell loop)
//   --mode=tsar is the reference's live path (external planes + weak.png -> region RANSAC -> plane fill);
//   --mode=patchmatch  (default) random init + iterations; --mode=load starts from
//                      APD/<id>/depths_geom.dmb + normals.dmb like the reference snapshot (main.cpp:1462-1490)
//   --all --fuse       after matching, every view's depth / normal map is gathered from the GPU that produced it to GPU 0
//                      over xGMI (tsar_peer_copy) and fused there (tsar_fuse) into D/APD/APD_TSAR.ply — the fuser of
//                      x/1.sh:30 without the round trip through .dmb files (which are still written)
//   --all resumes: a view whose APD/<id>/TSAR_disp.dmb and TSAR_normals.dmb are complete (the reference's header, main.cpp:1817-1860 /
//                      fileIoUtils.h:333-381, and exactly h*w*nb floats behind it) is skipped — the output files are the per-view
//                      checkpoints; --force recomputes.  A view that fails on one GPU is retried once on the next
//                      GPU's worker with a fresh context; the exit status is non-zero if any view's outputs are still missing.
//   --num_consistent= --reproj_error= --depth_diff= --angle= --used_list=   the fuser's options (x/1.sh:20-30), for --fuse
//   --seed=S, --strict, --fix-quirks, --texture-filter-8bit (TSAR_FLAG_TEX_FILTER_8BIT)
// Images: the scene's JPEGs as they are (host/tsar_jpeg.h: libjpeg's grayscale output = what the reference's imread returns,
// main.cpp:1302; bit-identical to libjpeg-turbo, tests/test_jpeg_decode.py), or binary PGM / PPM.  A name is looked up as the same
// stem + .pgm (.ppm with -color_processing) first — a user's own conversion wins — then as the JPEG of that stem.
//   --decode-image=IN[:OUT.pgm]   no GPU: decode one image the way a run would (after -color_processing: the blue channel), print
//                                 its size and checksums, optionally write it as PGM
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <future>
#include <map>
#include <math.h>
#include <memory>
#include <mutex>
#include <sstream>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/tsar.h"

struct Options {
    std::vector<std::string> images;
    std::string images_folder, mslp_folder, krt_file, output_folder;
    int iterations = 8, blocksize = 19, n_best = 2, cost_comb = TSAR_COMB_BEST_N;   // algorithmparameters.h:21-52
    float cam_scale = 1.0f, depth_min = -1.f, depth_max = -1.f;
    bool all = false, strict = false, fix_quirks = false, color = false, display_outputs = false, fuse = false, tex8 = false, timing = false, force = false;
    tsar_fusion_params fusion{};
    int gpus = 1, workers = 1;      // --all: worker threads per GPU; each overlaps its file output with the next view's kernels
    uint64_t seed = 0;
    std::string mode = "patchmatch";
};

#include "tsar_io.h"
#include "tsar_jpeg.h"

// The result buffers (depth / normal maps out) are page-locked (tsar_host_alloc): the DMA engine then writes them directly
// instead of going through the runtime's bounce buffers (a 6048 x 4032 view's results: 9 ms instead of 30), and one set serves
// every view of a worker.  Decoded images stay BYTES on the host (round 5: tsar_set_views_u8 widens them on the device) and are
// not page-locked: each is uploaded once per GPU (DeviceImageCache).
// Falls back to ordinary memory when page-locking fails.
static bool g_pin_results = false;     // set by main() for --all
template <class T>
struct PinnedAllocator {
    typedef T value_type;
    PinnedAllocator() = default;
    template <class U> PinnedAllocator(const PinnedAllocator<U>&) {}
    // never destroyed: buffers owned by objects with static storage (the image cache) are released after main() returns
    static std::mutex& mu() { static std::mutex* m = new std::mutex; return *m; }
    static std::map<void*, bool>& pinned() { static std::map<void*, bool>* s = new std::map<void*, bool>; return *s; }
    T* allocate(size_t n) {
        // page-locked only where a buffer is reused (--all: one set per worker for every view).  One view per process: page-locking
        // 0.39 GB on a helper thread beside the kernels contends with the runtime while pm_init's code object loads (that step
        // 13 -> 81 ms) to save 5 ms of copy: pageable there (profiles/r05/cli_single_view_breakdown.txt).  TSAR_GIPUMA_PIN=0/1 overrides.
        static const char* knob = getenv("TSAR_GIPUMA_PIN");
        const bool pin = knob ? knob[0] != '0' : g_pin_results;
        void* p = pin ? tsar_host_alloc(n * sizeof(T)) : nullptr;
        const bool is_pinned = p != nullptr;
        if (!p) p = malloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        std::lock_guard<std::mutex> lk(mu());
        pinned()[p] = is_pinned;
        return (T*)p;
    }
    void deallocate(T* p, size_t) {
        bool is_pinned = false;
        {
            std::lock_guard<std::mutex> lk(mu());
            auto it = pinned().find(p);
            if (it != pinned().end()) { is_pinned = it->second; pinned().erase(it); }
        }
        if (is_pinned) tsar_host_free(p); else free(p);
    }
    template <class U> bool operator==(const PinnedAllocator<U>&) const { return true; }
    template <class U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
typedef std::vector<float, PinnedAllocator<float>> PinnedFloats;

static std::string stem8(const std::string& name) { return name.substr(0, 8); }   // main.cpp:1460
static std::string pnm_name(const std::string& name, const char* want) {
    const size_t dot = name.find_last_of('.');
    const std::string ext = dot == std::string::npos ? "" : name.substr(dot);
    return (ext == want) ? name : name.substr(0, dot) + want;
}
// A view's image: the binary PGM (PPM with -color_processing) of that name where it exists, else the JPEG of the same stem beside
// it — what the reference's scene folders hold (scripts/courtyard.sh:7,16) — decoded like the reference's imread (host/tsar_jpeg.h).
static bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
static bool is_jpeg_name(const std::string& p) {
    const size_t dot = p.find_last_of('.');
    if (dot == std::string::npos) return false;
    std::string e = p.substr(dot);
    for (char& c : e) c = (char)tolower((unsigned char)c);
    return e == ".jpg" || e == ".jpeg";
}
static std::string resolve_view_image(const std::string& path) {
    if (file_exists(path)) return path;
    const size_t dot = path.find_last_of('.');
    const std::string stem = dot == std::string::npos ? path : path.substr(0, dot);
    for (const char* ext : {".jpg", ".JPG", ".jpeg", ".JPEG"})
        if (file_exists(stem + ext)) return stem + ext;
    return path;
}
static bool read_view_image(const std::string& path, bool blue, std::vector<uint8_t>& px, int& w, int& h, std::string* why = nullptr) {
    const std::string p = resolve_view_image(path);
    if (is_jpeg_name(p)) return tsar_jpeg::read(p, blue ? tsar_jpeg::BLUE : tsar_jpeg::LUMA, px, w, h, why);
    const bool ok = blue ? read_ppm_channel(p, 2, px, w, h) : read_pgm_u8(p, px, w, h);
    if (!ok && why) *why = std::string("not a readable binary ") + (blue ? "PPM" : "PGM") + " and no JPEG of that name beside it";
    return ok;
}
static bool view_image_size(const std::string& path, int& w, int& h) {
    const std::string p = resolve_view_image(path);
    return is_jpeg_name(p) ? tsar_jpeg::size(p, w, h) : pnm_size(p, w, h);
}
static void mkdirs(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i < path.size(); i++) {
        cur += path[i];
        if (path[i] == '/' || i + 1 == path.size()) mkdir(cur.c_str(), 0777);
    }
}

static void usage() {
    printf("usage: tsar_gipuma <ref image> <source images...> -images_folder DIR/ -mslp_folder DIR/ [-krt_file F] [-output_folder DIR]\n"
           "                   [--iterations=N] [--blocksize=N] [--cost_comb=all|best_n|angle|good] [--n_best=N] [--cam_scale=S]\n"
           "                   [--depth_min=D --depth_max=D] [--mode=patchmatch|load|tsar] [--all --gpus=N --workers=W] [--seed=S] [--strict] [--fix-quirks] [--texture-filter-8bit] [-color_processing] [--display_outputs] [--timing]\n"
           "       tsar_gipuma --all [--gpus=N] [--force] [--fuse [--num_consistent=N --reproj_error=PX --depth_diff=REL --angle=DEG --used_list=0|1]]\n"
           "                   -images_folder DIR/ -mslp_folder DIR/ [options]\n");
}

static int parse_args(int argc, char** argv, Options& o) {   // main.cpp:708-946: same spellings, unknown options only warn
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        auto starts = [&](const char* p) { return strncmp(a, p, strlen(p)) == 0; };
        if (a[0] != '-') o.images.push_back(a);
        else if (starts("--iterations=")) o.iterations = atoi(a + 13);
        else if (starts("--blocksize=")) {
            const int k = atoi(a + 12);
            if (k < 1 || k % 2 != 1) { printf("Command-line parameter error: The block size (--blocksize=<...>) must be a positive odd number\n"); return -1; }
            o.blocksize = k;
        } else if (starts("--n_best=")) o.n_best = atoi(a + 9);
        else if (starts("--cost_comb=")) {
            const char* v = a + 12;
            if (!strcmp(v, "all")) o.cost_comb = TSAR_COMB_ALL;
            else if (!strcmp(v, "best_n")) o.cost_comb = TSAR_COMB_BEST_N;
            else if (!strcmp(v, "angle")) o.cost_comb = TSAR_COMB_ANGLE;      // main.cpp:787-790; the kernels treat both as "all" (gipuma.cu:496-499)
            else if (!strcmp(v, "good")) o.cost_comb = TSAR_COMB_GOOD;
            else { printf("Command-line parameter error: Unknown cost combination method\n\n"); usage(); return -1; }
        } else if (starts("--cam_scale=")) o.cam_scale = (float)atof(a + 12);
        else if (starts("--depth_min=")) o.depth_min = (float)atof(a + 12);
        else if (starts("--depth_max=")) o.depth_max = (float)atof(a + 12);
        else if (starts("--gpus=")) o.gpus = atoi(a + 7);
        else if (starts("--workers=")) o.workers = atoi(a + 10);
        else if (starts("--seed=")) o.seed = strtoull(a + 7, nullptr, 10);
        else if (starts("--mode=")) o.mode = a + 7;
        else if (starts("--check-mask=")) {          // diagnostics, no GPU: decode a weak.png the way --mode=tsar does
            std::vector<float> scale;
            int mw = 0, mh = 0;
            if (!read_reliable_mask(a + 13, scale, mw, mh)) { printf("cannot decode %s\n", a + 13); return -1; }
            size_t ones = 0, wsum = 0;
            for (size_t k = 0; k < scale.size(); k++)
                if (scale[k] == 1.0f) { ones++; wsum += k % 9973; }
            printf("mask %d x %d reliable %zu checksum %zu\n", mw, mh, ones, wsum);
            return 1;
        }
        else if (starts("--decode-image=")) {        // no GPU: decode an image the way a run would, --decode-image=IN[:OUT.pgm] [-color_processing first]
            std::string in = a + 15, out;
            const size_t colon = in.rfind(':');
            if (colon != std::string::npos) { out = in.substr(colon + 1); in = in.substr(0, colon); }
            std::vector<uint8_t> px;
            int iw = 0, ih = 0;
            std::string why;
            if (!read_view_image(in, o.color, px, iw, ih, &why)) { printf("cannot decode %s: %s\n", in.c_str(), why.c_str()); return -1; }
            uint64_t sum = 0, mix = 1469598103934665603ull;
            for (uint8_t v : px) { sum += v; mix = (mix ^ v) * 1099511628211ull; }
            printf("image %d x %d sum %llu fnv1a %016llx\n", iw, ih, (unsigned long long)sum, (unsigned long long)mix);
            if (!out.empty()) {
                FILE* f = fopen(out.c_str(), "wb");
                if (!f || fprintf(f, "P5\n%d %d\n255\n", iw, ih) < 0 || fwrite(px.data(), 1, px.size(), f) != px.size()) { printf("cannot write %s\n", out.c_str()); if (f) fclose(f); return -1; }
                fclose(f);
            }
            return 1;
        }
        else if (!strcmp(a, "--all")) o.all = true;
        else if (!strcmp(a, "--fuse")) o.fuse = true;
        else if (!strcmp(a, "--force")) o.force = true;                        // --all: recompute views whose outputs are already there
        else if (starts("--num_consistent=")) o.fusion.num_consistent = atoi(a + 17);
        else if (starts("--reproj_error=")) o.fusion.reproj_error = (float)atof(a + 15);
        else if (starts("--depth_diff=")) o.fusion.depth_diff = (float)atof(a + 13);
        else if (starts("--angle=")) o.fusion.angle_deg = (float)atof(a + 8);
        else if (starts("--used_list=")) o.fusion.used_list = atoi(a + 12);
        else if (!strcmp(a, "--strict")) o.strict = true;
        else if (!strcmp(a, "--fix-quirks")) o.fix_quirks = true;
        else if (!strcmp(a, "--texture-filter-8bit")) o.tex8 = true;      // bilinear weights with 8 fractional bits, like the CUDA texture unit
        else if (!strcmp(a, "-images_folder") && i + 1 < argc) o.images_folder = argv[++i];
        else if (!strcmp(a, "-mslp_folder") && i + 1 < argc) o.mslp_folder = argv[++i];
        else if (!strcmp(a, "-krt_file") && i + 1 < argc) o.krt_file = argv[++i];
        else if (!strcmp(a, "-output_folder") && i + 1 < argc) o.output_folder = argv[++i];
        else if (!strcmp(a, "-color_processing")) o.color = true;      // main.cpp:727,909
        else if (!strcmp(a, "--timing")) o.timing = true;                     // wall time of each host-side step of a view, on stdout
        else if (!strcmp(a, "--display_outputs")) o.display_outputs = true;   // TSAR_normals.png + TSAR_model.ply (main.cpp:1800-1838)
        else if (!strcmp(a, "-no_display") || starts("--cost_gamma=") || starts("--min_angle=") || starts("--max_angle=") || starts("--cost_tau_color=") ||
                 starts("--cost_tau_gradient=") || starts("--cost_alpha=") || starts("--max_views=") || starts("--num_img_processed=")) {
            // accepted for script compatibility; these feed cost functions / view selection the GPU path does not use
        } else if (!strcmp(a, "-h") || !strcmp(a, "--help")) { usage(); return 1; }
        else printf("Command-line parameter warning: unknown option %s\n", a);
    }
    return 0;
}

// Decoded images shared by all views of a run (--all visits every image as a reference once and as a source ~N times).
struct ImageCache {
    struct Entry { std::vector<uint8_t> gray; int w = 0, h = 0; bool ok = false; std::string why; };      // the 8-bit decode as it is: widened to float on the device (tsar_set_views_u8)
    std::mutex mu;
    std::map<std::string, std::shared_ptr<Entry>> items;
    std::shared_ptr<Entry> get(const std::string& path) {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = items.find(path);
            if (it != items.end()) return it->second;
        }
        auto e = std::make_shared<Entry>();                       // decode outside the lock; a rare double decode is harmless
        const bool ppm = path.size() > 4 && path.compare(path.size() - 4, 4, ".ppm") == 0;
        e->ok = read_view_image(path, ppm, e->gray, e->w, e->h, &e->why);   // the PGM / PPM of that name, else the JPEG beside it (colour: blue, see tsar_io.h)
        std::lock_guard<std::mutex> lk(mu);
        auto ins = items.emplace(path, e);
        return ins.first->second;
    }
};
static ImageCache g_images;

// --all: every GPU keeps every image of the scene it has used resident (44 x 97.5 MB = 4.3 GB at ETH3D size, of
// 288 GB), uploaded once; a view then hands device pointers to tsar_set_views instead of pushing its 1 + N images over PCIe
// again (a scene's images are each the reference once and a source ~N times).
struct DeviceImageCache {
    std::mutex mu;
    std::map<std::pair<int, std::string>, uint8_t*> items;      // (device, path) -> device copy (bytes)
    const uint8_t* get(int device, const std::string& path, const ImageCache::Entry& host) {
        std::lock_guard<std::mutex> lk(mu);                       // uploads are rare (once per image and device): serialised
        auto it = items.find({device, path});
        if (it != items.end()) return it->second;
        const size_t bytes = (size_t)host.w * host.h;
        uint8_t* d = (uint8_t*)tsar_device_alloc(device, bytes);
        if (!d) return nullptr;
        if (tsar_device_write(device, d, host.gray.data(), bytes) != TSAR_OK) { tsar_device_free(device, d); return nullptr; }
        items[{device, path}] = d;
        return d;
    }
    void release() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& kv : items) tsar_device_free(kv.first.first, kv.second);
        items.clear();
    }
};
static DeviceImageCache g_device_images;

// --fuse: what a matched view leaves on its GPU for the gather (device memory, owned by the run)
struct DeviceResult {
    int device = -1, w = 0, h = 0;
    float *depth = nullptr, *normal = nullptr;
    tsar_camera cam{};
};

// What the refinement modes read per view besides the reference image: the external depth / normal maps and (--mode=tsar)
// weak.png.  Loaded by helper threads; in --all runs a worker keeps two of these, page-locked, and starts loading view k+1's
// while view k is on the GPU (inflating a full-size weak.png alone takes longer than the view's kernels).
struct ExternalInputs {
    bool pinned = false;                       // --all: buffers reused by every view of the worker, worth page-locking
    PinnedFloats pdepth, pnormal;
    std::vector<float> vdepth, vnormal, scale;
    int dh = 0, dw = 0, dnb = 0, nh = 0, nw = 0, nnb = 0, mw = 0, mh = 0;
    bool depth_ok = false, normal_ok = false, mask_ok = false, started = false;
    std::string dir;
    std::future<void> maps, mask;
    const float* depth() const { return pinned ? pdepth.data() : vdepth.data(); }
    const float* normal() const { return pinned ? pnormal.data() : vnormal.data(); }
    void start(const std::string& view_dir, bool want_mask, const std::string& ref_image_path);
};
void ExternalInputs::start(const std::string& view_dir, bool want_mask, const std::string& ref_image_path) {
    dir = view_dir;
    started = true;
    depth_ok = normal_ok = mask_ok = false;
    maps = std::async(std::launch::async, [this, ref_image_path]() {
        if (pinned) {
            depth_ok = read_dmb(dir + "depths_geom.dmb", pdepth, dh, dw, dnb);
            normal_ok = read_dmb(dir + "normals.dmb", pnormal, nh, nw, nnb);
        } else {
            depth_ok = read_dmb(dir + "depths_geom.dmb", vdepth, dh, dw, dnb);
            normal_ok = read_dmb(dir + "normals.dmb", vnormal, nh, nw, nnb);
        }
        if (!ref_image_path.empty()) g_images.get(ref_image_path);      // a prefetch: decoded into the cache for the view's own start
    });
    if (want_mask) mask = std::async(std::launch::async, [this]() { mask_ok = read_reliable_mask(dir + "weak.png", scale, mw, mh); });
}

// one reference view: images[0] is the reference, the rest the candidate sources in argv order
// page-locked result buffers of one worker, allocated once and reused for every view it processes (page-locking 390 MB per view
// would cost more than the copy it speeds up)
struct HostResult {
    PinnedFloats depth, normal;
    std::string out_dir;        // where the view's .dmb files go
    int w = 0, h = 0;
    tsar_ctx** shared_ctx = nullptr;   // the worker's context, kept across its views (device planes are allocated once)
    bool device_image_cache = false;   // --all: images stay resident on the device across views (a one-view process would only hold every image twice)
};
static bool write_view_files(const HostResult& r) {   // the two files side by side: a write is a copy into the page cache
    auto normals = std::async(std::launch::async, [&r]() { return write_dmb(r.out_dir + "TSAR_normals.dmb", r.normal.data(), r.h, r.w, 3); });
    const bool depth_ok = write_dmb(r.out_dir + "TSAR_disp.dmb", r.depth.data(), r.h, r.w, 1);
    return normals.get() && depth_ok;
}

// Fault injection for the re-queue path (tests): TSAR_GIPUMA_INJECT_FAILURE=<view id>[:<times>] makes the first <times> (default 1)
// attempts at that view fail after its context exists, the way a device-side error would (the context is dropped).
#include <atomic>
static int g_inject_view = -1;
static std::atomic<int> g_inject_left{0};
static void read_injection() {
    const char* e = getenv("TSAR_GIPUMA_INJECT_FAILURE");
    if (!e || !*e) return;
    g_inject_view = atoi(e);
    const char* c = strchr(e, ':');
    g_inject_left = c ? atoi(c + 1) : 1;
}
static std::string view_dir_of(const Options& o, int ref) { char b[32]; snprintf(b, sizeof b, "%08d", ref); return o.mslp_folder + "APD/" + b + "/"; }
static std::string view_image_of(const Options& o, int ref) { char b[32]; snprintf(b, sizeof b, "%08d.pgm", ref); return o.images_folder + pnm_name(b, o.color ? ".ppm" : ".pgm"); }
// the done marker of a view: both output maps complete for the size of its reference image
static bool outputs_complete(const Options& o, int ref) {
    int w = 0, h = 0;
    if (!view_image_size(view_image_of(o, ref), w, h)) return false;
    const std::string d = view_dir_of(o, ref);
    return dmb_complete(d + "TSAR_disp.dmb", h, w, 1) && dmb_complete(d + "TSAR_normals.dmb", h, w, 3);
}

static int run_view(const Options& o, int device, const std::vector<std::string>& names, const std::vector<int>& subset_slots, int ref_id, double* seconds,
                    DeviceResult* keep = nullptr, HostResult* reuse = nullptr, bool defer_write = false, ExternalInputs* preloaded = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    auto t_last = t0;
    std::string steps;                                            // --timing: "step ms | step ms | ..."
    auto stamp = [&](const char* what) {
        if (!o.timing) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof buf, "%s%s %.1f", steps.empty() ? "" : " | ", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        steps += buf;
        t_last = now;
    };
    // The refinement modes never read a source image (load_planes, weak-texture detection, region RANSAC and fill work on the
    // reference view and the plane maps): only the reference image is decoded and handed to the library there.
    const bool tsar_mode = o.mode == "tsar", external = o.mode == "load" || tsar_mode;
    const int n_all = (int)names.size(), n = external ? 1 : n_all;
    const std::string out_dir = o.mslp_folder + "APD/" + stem8(names[0]) + "/";   // main.cpp:1462, 1813-1830
    // Everything a process has to do before its first kernel runs side by side: the HIP context (~0.2 s in a fresh process), the
    // decode of the 1 + N images (~45 ms each at ETH3D size), and — refinement modes — the external maps and weak.png (inflating
    // a full-size PNG takes ~0.3 s).  In --all runs the context and the images are already there for every view but the first.
    tsar_ctx** shared = reuse ? reuse->shared_ctx : nullptr;
    tsar_ctx* ctx = shared ? *shared : nullptr;
    std::future<int> creating;
    double ms_create = 0.0, ms_decode = 0.0;                        // --timing: the two concurrent legs of the start-up, each on its own clock
    if (!ctx) creating = std::async(std::launch::async, [&ctx, device, &ms_create]() {
        const auto c0 = std::chrono::steady_clock::now();
        const int rc = tsar_create(device, &ctx);
        ms_create = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
        return rc;
    });
    ExternalInputs own_inputs;
    ExternalInputs& ext = (preloaded && preloaded->started && preloaded->dir == out_dir) ? *preloaded : own_inputs;   // --all: started a view ago
    if (external && !ext.started) ext.start(out_dir, tsar_mode, "");
    struct Consumed { ExternalInputs& e; ~Consumed() { if (e.maps.valid()) e.maps.get(); if (e.mask.valid()) e.mask.get(); e.started = false; } } consumed{ext};
    std::vector<std::shared_ptr<ImageCache::Entry>> gray(n);
    {
        std::vector<std::future<std::shared_ptr<ImageCache::Entry>>> decoding;
        for (int i = 0; i < n; i++)
            decoding.push_back(std::async(std::launch::async, [&o, &names, i]() { return g_images.get(o.images_folder + pnm_name(names[i], o.color ? ".ppm" : ".pgm")); }));
        for (int i = 0; i < n; i++) gray[i] = decoding[i].get();
        ms_decode = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // (the map / mask readers are joined where their data is needed — set_views and load_planes run while weak.png is still
    // inflating — or by `consumed` on an early return)
    const int create_rc = creating.valid() ? creating.get() : TSAR_OK;
    if (create_rc != TSAR_OK) { fprintf(stderr, "tsar_create(device %d) failed: %d\n", device, create_rc); return create_rc; }
    if (shared) *shared = ctx;
    // a context is destroyed here only when this call owns it, or after a failure (the next view then starts from a fresh one)
    auto drop_ctx = [&]() { tsar_destroy(ctx); if (shared) *shared = nullptr; };
    auto fail = [&](const char* what) { fprintf(stderr, "%s: %s\n", what, tsar_last_error(ctx)); drop_ctx(); return -1; };
    std::vector<const uint8_t*> ptrs(n), dev_ptrs;
    std::vector<tsar_camera> cams(n);
    int w = 0, h = 0;
    float dmin = o.depth_min, dmax = o.depth_max;
    for (int i = 0; i < n; i++) {
        const std::string ip = o.images_folder + pnm_name(names[i], o.color ? ".ppm" : ".pgm");
        if (!gray[i]->ok) { fprintf(stderr, "cannot read image %s: %s\n", ip.c_str(), gray[i]->why.c_str()); drop_ctx(); return -1; }
        const int wi = gray[i]->w, hi = gray[i]->h;
        if (i == 0) { w = wi; h = hi; }
        if (wi != w || hi != h) { fprintf(stderr, "image %s has a different size\n", ip.c_str()); drop_ctx(); return -1; }
        ptrs[i] = gray[i]->gray.data();
        if (reuse && reuse->device_image_cache) {                 // --all: resident device copy (falls back to the host buffer)
            const uint8_t* d = g_device_images.get(device, ip, *gray[i]);
            if (d) dev_ptrs.push_back(d);
        }
        CamFile cf;
        const std::string cp = o.mslp_folder + "cams/" + stem8(names[i]) + "_cam.txt";
        if (!read_cam(cp, cf)) { fprintf(stderr, "cannot read camera %s\n", cp.c_str()); drop_ctx(); return -1; }
        cams[i] = cf.cam;
        if (i == 0) {   // depth range of the reference view (fileIoUtils.h:150-153) unless given on the command line
            if (dmin <= 0) dmin = cf.depth_min;
            if (dmax <= 0) dmax = cf.depth_max;
        }
    }
    tsar_params p;
    tsar_default_params(&p);
    p.box_hsize = p.box_vsize = o.blocksize;
    p.n_best = o.n_best; p.cost_comb = o.cost_comb; p.cam_scale = o.cam_scale;
    p.depth_min = dmin; p.depth_max = dmax;
    p.seed = o.seed + (uint64_t)ref_id;
    p.flags = (o.strict ? TSAR_FLAG_STRICT_DIV : 0) | (o.fix_quirks ? (TSAR_FLAG_FIX_DOWN_FAR_SEED | TSAR_FLAG_FIX_RIGHT_FAR_CMP) : 0) |
              (o.tex8 ? TSAR_FLAG_TEX_FILTER_8BIT : 0);
    if (tsar_set_params(ctx, &p) != TSAR_OK) return fail("tsar_set_params");
    if (ref_id == g_inject_view && g_inject_left.fetch_sub(1) > 0) {
        fprintf(stderr, "view %08d on gpu %d: injected failure (TSAR_GIPUMA_INJECT_FAILURE)\n", ref_id, device);
        drop_ctx();
        return -1;
    }
    if (o.timing) { tsar_enable_kernel_timing(ctx, 1); tsar_reset_kernel_timing(ctx); }
    stamp(external ? "context + reference image (external maps and weak.png still loading)" : "context + images + cameras (concurrent)");
    if (o.timing) {
        char buf[96];
        snprintf(buf, sizeof buf, " [tsar_create %.1f beside decode of %d images %.1f]", ms_create, n, ms_decode);
        steps += buf;
    }
    const bool resident = (int)dev_ptrs.size() == n;
    if (tsar_set_views_u8(ctx, n, w, h, resident ? dev_ptrs.data() : ptrs.data(), resident ? TSAR_MEM_DEVICE : TSAR_MEM_HOST, cams.data()) != TSAR_OK) return fail("tsar_set_views_u8");
    if (!subset_slots.empty() && !external) {
        std::vector<int32_t> s(subset_slots.begin(), subset_slots.end());
        if (tsar_set_view_subset(ctx, (int)s.size(), s.data()) != TSAR_OK) return fail("tsar_set_view_subset");
    }
    stamp("set_views");
    // (One view per process: releasing the decoded images — 1.07 GB of host memory at ETH3D size — here, on a helper thread beside the
    // kernels, instead of leaving them to the exit path was measured and removed: the caller waits ~40 ms per GB the process still
    // holds when main() leaves, but the munmap contends with the runtime's own mappings while pm_init's code object loads, 65 -> 200 ms
    // for that step, and the invocation got slower, 1239 -> 1323 ms median: profiles/r05/cli_single_view_breakdown.txt.)
    mkdirs(out_dir);
    const size_t np = (size_t)w * h;
    // sizing (and, in --all, page-locking once per worker) of the result buffers happens beside the kernels
    HostResult local;
    HostResult& hr = reuse ? *reuse : local;
    std::future<void> sizing;
    if (hr.depth.size() != np) sizing = std::async(std::launch::async, [&hr, np]() { hr.depth.resize(np); hr.normal.resize(3 * np); });
    struct JoinSizing { std::future<void>& f; ~JoinSizing() { if (f.valid()) f.get(); } } join_sizing{sizing};   // on every return path
    if (external) {
        if (ext.maps.valid()) ext.maps.get();
        if (!ext.depth_ok || ext.dh != h || ext.dw != w || ext.dnb != 1) { fprintf(stderr, "cannot read %sdepths_geom.dmb\n", out_dir.c_str()); drop_ctx(); return -1; }
        if (!ext.normal_ok || ext.nh != h || ext.nw != w || ext.nnb != 3) { fprintf(stderr, "cannot read %snormals.dmb\n", out_dir.c_str()); drop_ctx(); return -1; }
        if (tsar_load_planes(ctx, ext.depth(), ext.normal(), TSAR_MEM_HOST) != TSAR_OK) return fail("tsar_load_planes");
        stamp("load_planes");
    } else {
        if (tsar_pm_init(ctx) != TSAR_OK) return fail("tsar_pm_init");
        if (o.timing) { tsar_synchronize(ctx); stamp("pm_init (first launch of its code object)"); }
        if (tsar_pm_iterate(ctx, o.iterations) != TSAR_OK) return fail("tsar_pm_iterate");
        stamp(o.timing ? "pm_iterate" : "pm_init + pm_iterate");
    }
    if (tsar_mode) {
        // the reference's live path, runGipuma main.cpp:1493-1783: external planes (above: firstcuda) -> reliability mask
        // from weak.png -> weak-texture regions of the reference image (texture(), main.cpp:214-596) -> sliccuda
        // (gipuma_getview) -> per-region plane RANSAC (:1520-1730) -> fakecuda -> fillcuda
        if (ext.mask.valid()) ext.mask.get();
        stamp("weak.png (rest of its inflate)");
        if (!ext.mask_ok || ext.mw != w || ext.mh != h) { fprintf(stderr, "cannot read %sweak.png (8-bit PNG of the image size)\n", out_dir.c_str()); drop_ctx(); return -1; }
        if (tsar_set_reliable_mask(ctx, ext.scale.data(), TSAR_MEM_HOST) != TSAR_OK) return fail("tsar_set_reliable_mask");
        stamp("set_reliable_mask");
        int n_regions = 0;
        if (tsar_detect_weak_texture(ctx, nullptr, TSAR_MEM_HOST, &n_regions, nullptr, nullptr, 0) != TSAR_OK) return fail("tsar_detect_weak_texture");
        stamp("detect_weak_texture");
        if (tsar_getview(ctx) != TSAR_OK) return fail("tsar_getview");
        std::vector<float> planes((size_t)4 * (n_regions > 0 ? n_regions : 1)), ratio((size_t)(n_regions > 0 ? n_regions : 1));
        if (tsar_ransac_regions(ctx, planes.data(), ratio.data()) != TSAR_OK) return fail("tsar_ransac_regions");
        stamp("getview + ransac_regions");
        if (tsar_fake_depth(ctx, nullptr, TSAR_MEM_HOST) != TSAR_OK) return fail("tsar_fake_depth");
        if (tsar_fill_textureless(ctx) != TSAR_OK) return fail("tsar_fill_textureless");
        stamp("fake_depth + fill_textureless");
        printf("view %08d: %d regions labelled, textureless ones refitted and filled\n", ref_id, n_regions);
    } else if (tsar_compute_disp(ctx) != TSAR_OK) return fail("tsar_compute_disp");
    if (sizing.valid()) sizing.get();
    PinnedFloats &depth = hr.depth, &normal = hr.normal;
    if (tsar_get_result(ctx, depth.data(), normal.data(), nullptr, nullptr, TSAR_MEM_HOST) != TSAR_OK) return fail("tsar_get_result");
    stamp("get_result");
    if (keep) {   // the same maps stay on this GPU for the gather to the fusing device
        keep->device = device; keep->w = w; keep->h = h; keep->cam = cams[0];
        keep->depth = (float*)tsar_device_alloc(device, np * 4);
        keep->normal = (float*)tsar_device_alloc(device, np * 12);
        if (!keep->depth || !keep->normal) return fail("tsar_device_alloc");
        if (tsar_get_result(ctx, keep->depth, keep->normal, nullptr, nullptr, TSAR_MEM_DEVICE) != TSAR_OK) return fail("tsar_get_result (device)");
    }
    hr.out_dir = out_dir; hr.w = w; hr.h = h;
    if (!defer_write && !write_view_files(hr)) return -1;      // deferred: the caller writes while the next view is being matched
    if (!defer_write) stamp("write .dmb");
    if (o.timing) {
        printf("view %08d steps (ms): %s\n", ref_id, steps.c_str());
        tsar_kernel_timing kt[64];
        int nk = 0;
        if (tsar_get_kernel_timing(ctx, kt, 64, &nk) == TSAR_OK) {
            printf("view %08d kernels (launches x mean ms):", ref_id);
            for (int k = 0; k < nk && k < 64; k++) printf(" %s %d x %.3f |", kt[k].name, kt[k].launches, kt[k].launches ? kt[k].total_ms / kt[k].launches : 0.0);
            printf("\n");
        }
    }
    if (!shared) tsar_destroy(ctx);
    if (o.display_outputs) {   // the reference always writes these two; here on request (a full-size view's PLY is 0.66 GB)
        std::vector<uint16_t> vis(3 * np);
        for (size_t k = 0; k < 3 * np; k++) {
            const float v = normal[k] * 32767.f + 32767.f;               // convertTo(CV_16U, 32767, 32767): saturate + round
            vis[k] = (uint16_t)(v <= 0.f ? 0 : v >= 65535.f ? 65535 : (int)lrintf(v));
        }
        if (!write_png_rgb16(out_dir + "TSAR_normals.png", vis.data(), w, h)) return -1;
        const std::vector<float> ref_gray(gray[0]->gray.begin(), gray[0]->gray.end());
        if (!write_view_ply(out_dir + "TSAR_model.ply", depth.data(), normal.data(), ref_gray.data(), w, h, cams[0])) return -1;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (seconds) *seconds = sec;
    FILE* rf = fopen((out_dir + "TSAR_results.txt").c_str(), "a");   // main.cpp:1854-1860
    if (rf) { fprintf(rf, "Total runtime: %g sec ( %g min)\n", sec, sec / 60.0); fclose(rf); }
    return 0;
}

// --timing, one view per process: milliseconds from exec() to now, from the process's start time in /proc (10 ms resolution)
static double ms_since_exec() {
    FILE* f = fopen("/proc/self/stat", "r");
    if (!f) return -1.0;
    char buf[2048];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char* p = strrchr(buf, ')');                 // the command name may contain blanks
    if (!p) return -1.0;
    unsigned long long start_ticks = 0;
    int field = 2;
    for (p++; *p && field < 22; p++)
        if (*p == ' ') { field++; if (field == 22) start_ticks = strtoull(p + 1, nullptr, 10); }
    double up = 0.0;
    f = fopen("/proc/uptime", "r");
    if (!f || fscanf(f, "%lf", &up) != 1) { if (f) fclose(f); return -1.0; }
    fclose(f);
    return (up - (double)start_ticks / (double)sysconf(_SC_CLK_TCK)) * 1e3;
}

int main(int argc, char** argv) {
    const double ms_exec_to_main = ms_since_exec();
    Options o;
    tsar_default_fusion_params(&o.fusion);
    const int pr = parse_args(argc, argv, o);
    if (pr != 0) return pr < 0 ? 1 : 0;
    if (o.mslp_folder.empty() || o.images_folder.empty()) { usage(); return 1; }
    if (o.mslp_folder.back() != '/') o.mslp_folder += '/';
    if (o.images_folder.back() != '/') o.images_folder += '/';
    std::map<int, std::vector<int>> pairs;
    const bool have_pairs = read_pairs(o.mslp_folder + "pair.txt", pairs);
    read_injection();
    if (o.all) {
        g_pin_results = true;
        if (!have_pairs) { fprintf(stderr, "--all needs %spair.txt\n", o.mslp_folder.c_str()); return 1; }
        std::vector<int> refs;
        for (auto& kv : pairs) refs.push_back(kv.first);
        const int ngpu = o.gpus < 1 ? 1 : o.gpus;
        const int nthr = ngpu * (o.workers < 1 ? 1 : o.workers);   // worker t drives GPU t % ngpu with its own context
        std::vector<int> status(nthr, 0);
        std::vector<DeviceResult> kept(o.fuse ? refs.size() : 0);
        // resume: the views whose output files are complete are not matched again (decided up front, so that nothing is read ahead for them)
        std::vector<char> skip(refs.size(), 0);
        std::vector<int> view_rc(refs.size(), 0), view_gpu(refs.size(), -1);
        size_t n_skip = 0;
        if (!o.force)
            for (size_t k = 0; k < refs.size(); k++) n_skip += (skip[k] = outputs_complete(o, refs[k]) ? 1 : 0);
        if (n_skip) printf("resuming: %zu of %zu views already have complete TSAR_disp.dmb / TSAR_normals.dmb and are skipped (--force recomputes them)\n", n_skip, refs.size());
        // --fuse needs a skipped view's maps on a device all the same: read back from its files
        auto load_kept = [&](size_t k, int g) {
            std::vector<float> d, nr;
            int h = 0, w = 0, nb = 0, h2 = 0, w2 = 0, nb2 = 0;
            const std::string dir = view_dir_of(o, refs[k]);
            if (!read_dmb(dir + "TSAR_disp.dmb", d, h, w, nb) || !read_dmb(dir + "TSAR_normals.dmb", nr, h2, w2, nb2) || h != h2 || w != w2 || nb != 1 || nb2 != 3) return false;
            DeviceResult& r = kept[k];
            r.device = g; r.w = w; r.h = h;
            r.depth = (float*)tsar_device_alloc(g, d.size() * 4);
            r.normal = (float*)tsar_device_alloc(g, nr.size() * 4);
            return r.depth && r.normal && tsar_device_write(g, r.depth, d.data(), d.size() * 4) == TSAR_OK && tsar_device_write(g, r.normal, nr.data(), nr.size() * 4) == TSAR_OK;
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; t++)
            th.emplace_back([&, t]() {
                const int g = t % ngpu;
                // two page-locked result sets per worker: the .dmb files of view k are written by a helper thread while the
                // kernels of view k+1 run (file output is ~0.1 s of a 0.5 s view at ETH3D size)
                HostResult host_result[2];
                tsar_ctx* worker_ctx = nullptr;
                host_result[0].shared_ctx = host_result[1].shared_ctx = &worker_ctx;
                host_result[0].device_image_cache = host_result[1].device_image_cache = true;
                std::future<bool> writing[2];
                // refinement modes: a ring of (page-locked) input buffers; the maps, weak.png and reference image of the next seven
                // views are read while view k is on the GPU (one weak.png inflates in ~0.3 s, a view's kernels take ~0.1 s)
                const bool external = o.mode == "load" || o.mode == "tsar";
                // (about sixteen sets in flight per process: eight with one worker, two per worker on an 8-GPU node — each set
                // page-locks 0.39 GB at ETH3D size, and the inflates run on as many host threads)
                const size_t RING = std::max<size_t>(2, std::min<size_t>(8, 16 / (size_t)nthr));
                std::vector<ExternalInputs> inputs(RING);
                for (auto& in : inputs) in.pinned = true;
                auto view_dir = [&](int ref) { return view_dir_of(o, ref); };
                auto ref_image = [&](int ref) { return view_image_of(o, ref); };
                size_t turn = 0;
                for (size_t k = t; k < refs.size(); k += nthr, turn++) {   // round-robin: every view of a scene costs the same
                    const int ref = refs[k];
                    view_gpu[k] = g;
                    if (skip[k]) {
                        printf("view %08d: outputs present, skipped\n", ref);
                        if (o.fuse && !load_kept(k, g)) { fprintf(stderr, "view %08d: cannot read its output files back for --fuse\n", ref); view_rc[k] = -1; }
                        continue;
                    }
                    char buf[32];
                    std::vector<std::string> names;
                    snprintf(buf, sizeof buf, "%08d.pgm", ref);
                    names.push_back(buf);
                    for (int s : pairs[ref]) { snprintf(buf, sizeof buf, "%08d.pgm", s); names.push_back(buf); }
                    double sec = 0;
                    HostResult& hr = host_result[turn & 1];
                    if (writing[turn & 1].valid() && !writing[turn & 1].get()) status[t] = -1;      // the set's previous files are on disk
                    if (external) {
                        if (!inputs[turn % RING].started) inputs[turn % RING].start(view_dir(ref), o.mode == "tsar", "");
                        for (size_t j = 1; j < RING; j++) {
                            ExternalInputs& ahead = inputs[(turn + j) % RING];
                            const size_t kk = k + j * nthr;
                            if (kk < refs.size() && !skip[kk] && !ahead.started) ahead.start(view_dir(refs[kk]), o.mode == "tsar", ref_image(refs[kk]));
                        }
                    }
                    const int rc = run_view(o, g, names, {}, ref, &sec, o.fuse ? &kept[k] : nullptr, &hr, /*defer_write*/ true, external ? &inputs[turn % RING] : nullptr);
                    printf("view %08d on gpu %d: %s (%.2f s)\n", ref, g, rc == 0 ? "ok" : "FAILED", sec);
                    view_rc[k] = rc;               // a failed view does not stop the others; it is retried below
                    if (rc == 0) writing[turn & 1] = std::async(std::launch::async, [&hr]() { return write_view_files(hr); });
                }
                for (auto& f : writing)
                    if (f.valid() && !f.get()) status[t] = -1;
                tsar_destroy(worker_ctx);
            });
        for (auto& t : th) t.join();
        // re-queue: a view that failed is tried once more on the NEXT gpu's turn (the same one when there is only one) with a
        // context of its own, created fresh and destroyed with the view — never the worker context the failure left behind
        for (size_t k = 0; k < refs.size(); k++) {
            if (view_rc[k] == 0 || skip[k]) continue;
            const int g2 = (view_gpu[k] + 1) % ngpu;
            printf("view %08d FAILED on gpu %d: retrying once on gpu %d with a fresh context\n", refs[k], view_gpu[k], g2);
            if (o.fuse) {
                if (kept[k].depth) tsar_device_free(kept[k].device, kept[k].depth);
                if (kept[k].normal) tsar_device_free(kept[k].device, kept[k].normal);
                kept[k] = DeviceResult{};
            }
            char buf[32];
            std::vector<std::string> names;
            snprintf(buf, sizeof buf, "%08d.pgm", refs[k]);
            names.push_back(buf);
            for (int sv : pairs[refs[k]]) { snprintf(buf, sizeof buf, "%08d.pgm", sv); names.push_back(buf); }
            double sec = 0;
            view_rc[k] = run_view(o, g2, names, {}, refs[k], &sec, o.fuse ? &kept[k] : nullptr);
            printf("view %08d on gpu %d (retry): %s (%.2f s)\n", refs[k], g2, view_rc[k] == 0 ? "ok" : "FAILED", sec);
        }
        g_device_images.release();
        int missing = 0;
        for (size_t k = 0; k < refs.size(); k++)
            if (view_rc[k] != 0 || !outputs_complete(o, refs[k])) { fprintf(stderr, "view %08d: outputs missing or incomplete\n", refs[k]); missing++; }
        for (int s : status)
            if (s != 0) missing++;      // a file of an otherwise matched view could not be written
        if (missing) return 1;
        if (o.fuse) {
            // gather: every view's maps to GPU 0 (peer copies over xGMI; views matched on GPU 0 are already there), then fuse
            const auto t0 = std::chrono::steady_clock::now();
            if (refs.empty() || kept.empty()) { fprintf(stderr, "--fuse: no view was matched\n"); return 1; }
            const int n = (int)refs.size(), fw = kept[0].w, fh = kept[0].h;
            const size_t np = (size_t)fw * fh;
            std::map<int, int> slot;
            for (int k = 0; k < n; k++) slot[refs[k]] = k;
            std::vector<const float*> pd(n), pn(n), pg(n);
            std::vector<tsar_camera> cams(n);
            std::vector<void*> owned;                       // device-0 buffers to release
            auto release = [&]() { for (void* q : owned) tsar_device_free(0, q); };
            size_t moved = 0;
            for (int k = 0; k < n; k++) {
                if (kept[k].w != fw || kept[k].h != fh) { fprintf(stderr, "--fuse: views differ in size\n"); release(); return 1; }
                float *d = kept[k].depth, *nr = kept[k].normal;
                if (kept[k].device != 0) {
                    float* d0 = (float*)tsar_device_alloc(0, np * 4);
                    float* n0 = (float*)tsar_device_alloc(0, np * 12);
                    if (d0) owned.push_back(d0);            // released on every error path below
                    if (n0) owned.push_back(n0);
                    if (!d0 || !n0 || tsar_peer_copy(0, d0, kept[k].device, d, np * 4) != TSAR_OK || tsar_peer_copy(0, n0, kept[k].device, nr, np * 12) != TSAR_OK) {
                        fprintf(stderr, "--fuse: gather of view %08d from gpu %d failed\n", refs[k], kept[k].device);
                        release();
                        return 1;
                    }
                    tsar_device_free(kept[k].device, d);
                    tsar_device_free(kept[k].device, nr);
                    d = d0; nr = n0;
                    moved += np * 16;
                } else {
                    owned.push_back(d); owned.push_back(nr);
                }
                char buf[32];
                snprintf(buf, sizeof buf, "%08d.pgm", refs[k]);
                auto img = g_images.get(o.images_folder + pnm_name(buf, o.color ? ".ppm" : ".pgm"));
                float* g0 = (float*)tsar_device_alloc(0, np * 4);
                if (g0) owned.push_back(g0);
                const std::vector<float> img_f(img->gray.begin(), img->gray.end());      // the fuser colours its points from float images
                if (!img->ok || img_f.size() != np || !g0 || tsar_device_write(0, g0, img_f.data(), np * 4) != TSAR_OK) { fprintf(stderr, "--fuse: image of view %08d\n", refs[k]); release(); return 1; }
                pd[k] = d; pn[k] = nr; pg[k] = g0;
                char cname[32];
                snprintf(cname, sizeof cname, "%08d", refs[k]);
                CamFile cf;
                if (!read_cam(o.mslp_folder + "cams/" + cname + "_cam.txt", cf)) { fprintf(stderr, "--fuse: camera of view %08d\n", refs[k]); release(); return 1; }
                cams[k] = cf.cam;
            }
            const double t_gather = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::vector<int32_t> off(n + 1, 0), idx;
            for (int k = 0; k < n; k++) {
                for (int sv : pairs[refs[k]])
                    if (slot.count(sv)) idx.push_back(slot[sv]);
                off[k + 1] = (int32_t)idx.size();
            }
            if (idx.empty()) idx.push_back(0);
            const int64_t cap = (int64_t)n * fw * fh;
            std::unique_ptr<float[]> pts(new float[(size_t)cap * 9]);   // not zero-filled: only the fused points' pages are ever touched
            int64_t cnt = 0;
            const int rc = tsar_fuse(0, n, fw, fh, cams.data(), pd.data(), pn.data(), pg.data(), TSAR_MEM_DEVICE, off.data(), idx.data(), &o.fusion, pts.get(), cap, &cnt);
            release();
            if (rc != TSAR_OK) { fprintf(stderr, "tsar_fuse failed: %d\n", rc); return 1; }
            if (cnt > cap) cnt = cap;
            const std::string out = o.mslp_folder + "APD/APD_TSAR.ply";
            if (!write_cloud_ply(out, pts.get(), cnt)) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
            const double t_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("fused %d views on gpu 0: %lld points -> %s (gather of %.1f MB from other gpus + uploads %.3f s, total %.3f s)\n", n, (long long)cnt, out.c_str(),
                   moved / 1e6, t_gather, t_all);
        }
        return 0;
    }
    if (o.images.size() < 2) { usage(); return 1; }
    // camera id from the reference image name, source slots from pair.txt (main.cpp:1347-1376)
    const int camera_id = atoi(o.images[0].substr(4, 8).c_str());
    std::vector<int> slots;
    if (have_pairs && pairs.count(camera_id))
        for (int s : pairs[camera_id]) slots.push_back(s > camera_id ? s : s + 1);
    for (int s : slots)
        if (s < 1 || s >= (int)o.images.size()) { fprintf(stderr, "pair.txt refers to view slot %d but only %zu images were given\n", s, o.images.size()); return 1; }
    double sec = 0;
    // one view per process (the reference's shell loop): the context and the buffers stay alive until the process ends, and the
    // process ends without tearing them down one by one — the files are on disk, the driver reclaims the rest (0.25 s of a 1.4 s
    // invocation at ETH3D size went into freeing 1.5 GB of host buffers, the context and the runtime's own shutdown)
    static HostResult single;
    static tsar_ctx* single_ctx = nullptr;
    single.shared_ctx = &single_ctx;
    const int rc = run_view(o, 0, o.images, slots, camera_id, &sec, nullptr, &single);
    printf("Total runtime including disk i/o: %gsec\n", sec);
    if (o.timing) printf("process (ms since exec, 10 ms resolution): main entered at %.0f, leaving at %.0f (what the caller waits for beyond that is the teardown of the process's GPU state by the driver)\n",
                         ms_exec_to_main, ms_since_exec());
    fflush(nullptr);
    _exit(rc == 0 ? 0 : 1);
}
