// gipuma_tsar_shim.cpp — what a maintainer of ZhenlongYuan/TSAR-MVS drops in place of gipuma.cu to run the GPU operators of
// the reference on an MI355X through libtsar_hip.so (include/tsar.h).
//
// NOT part of this repository's build: it includes the reference's own headers (gipuma.h, globalstate.h, main.h) and OpenCV,
// none of which exist in this image, and it is written against them as they are in the reference snapshot.  Every helper it
// uses is defined here or in host/tsar_io.h (plain C++, zlib only).  Required changes on the reference side, all in main.cpp:
//   1. managed.h: `Managed::operator new` -> plain `new` / `calloc` (the state objects become ordinary host memory; the
//      library owns all device memory).
//   2. main.cpp:1445-1449  addImageToTextureFloat{Gray,Color}(...)  ->  tsar_shim_set_views(inputFiles, img_grayscale_float, algParams, *gs);
//   3. main.cpp:1473-1490  keep the fill loop, add  tsar_shim_set_external_planes(ref_depth, ref_normal);  (the library
//      needs the depth map as read, not the disparity the loop derives from it)
//   4. main.cpp:1520-1730  (CPU RANSAC per region) is deleted: sliccuda() below runs it on the GPU and fills cannylines->norm4.
//   5. link libtsar_hip.so instead of the CUDA runtime / cuRAND.
// The four operator entry points keep their names, signatures and "always return 0" convention (gipuma.h:2-6, gipuma.cu:1879-1913).
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>

#include "gipuma.h"          // the reference's header: firstcuda / sliccuda / fillcuda / fakecuda
#include "globalstate.h"     // GlobalState, LineState, CameraParameters_cu (reference)
#include "main.h"            // InputFiles, AlgorithmParameters (reference)
#include "tsar.h"            // this repository: include/tsar.h
#include "tsar_io.h"         // this repository: tsar-mvs_amd/host/tsar_io.h (read_cam: cams/%08d_cam.txt, fileIoUtils.h:117-153)

namespace {
tsar_ctx* g_ctx = nullptr;
std::vector<float> g_ext_depth, g_ext_normal;          // external planes as read from depths_geom.dmb / normals.dmb
bool g_run_patchmatch = false;                         // true: the loop the reference has commented out (gipuma.cu:1741-1754)
int g_iterations = 8;

void report(const char* what) { fprintf(stderr, "[tsar shim] %s: %s\n", what, tsar_last_error(g_ctx)); }

// lines->canny is a float plane holding region ids (linestate.h:17); the library takes int32 labels
std::vector<int32_t> canny_labels_as_int(const GlobalState& gs) {
    const size_t np = (size_t)gs.col * gs.row;
    std::vector<int32_t> lb(np);
    for (size_t p = 0; p < np; p++) lb[p] = (int32_t)gs.lines->canny[p];
    return lb;
}
}   // namespace

// Replaces the texture upload (main.cpp:1190-1228, 1445-1449) and the camera copy into Camera_cu (cameraGeometryUtils.h:311-356).
// img_grayscale_float: CV_32F gray images in argv order (index 0 = reference view).  Cameras are re-read from the cam files, so
// that the library derives the reference-relative cameras itself (no decomposeProjectionMatrix round trip).
int tsar_shim_set_views(const InputFiles& in, const std::vector<cv::Mat>& img_grayscale_float, const AlgorithmParameters& alg, GlobalState& gs) {
    if (!g_ctx && tsar_create(/*device*/ 0, &g_ctx) != TSAR_OK) { fprintf(stderr, "[tsar shim] no HIP device\n"); return -1; }
    const int n = (int)img_grayscale_float.size();
    std::vector<tsar_camera> cams(n);
    float depth_min = 0, depth_max = 0;
    for (int i = 0; i < n; i++) {
        CamFile cf;
        const std::string path = in.mslp_folder + "cams/" + in.img_filenames[i].substr(0, 8) + "_cam.txt";     // main.cpp:1348
        if (!read_cam(path, cf)) { fprintf(stderr, "[tsar shim] cannot read %s\n", path.c_str()); return -1; }
        cams[i] = cf.cam;
        if (i == 0) { depth_min = cf.depth_min; depth_max = cf.depth_max; }                                     // main.cpp:1386-1391
    }
    tsar_params p;
    tsar_default_params(&p);
    p.box_hsize = alg.box_hsize;  p.box_vsize = alg.box_vsize;      // algorithmparameters.h:57-58
    p.n_best = alg.n_best;        p.cost_comb = alg.cost_comb;      // :75-76
    p.cam_scale = alg.cam_scale;
    p.depth_min = depth_min;      p.depth_max = depth_max;
    p.seed = 0;                                                     // the reference seeds from clock64(): any value is "the reference"
    g_iterations = alg.iterations;
    if (tsar_set_params(g_ctx, &p) != TSAR_OK) { report("tsar_set_params"); return -1; }
    std::vector<const float*> gray(n);
    for (int i = 0; i < n; i++) {
        if (!img_grayscale_float[i].isContinuous() || img_grayscale_float[i].type() != CV_32FC1) return -1;
        gray[i] = img_grayscale_float[i].ptr<float>(0);
    }
    if (tsar_set_views(g_ctx, n, gs.col, gs.row, gray.data(), TSAR_MEM_HOST, cams.data()) != TSAR_OK) { report("tsar_set_views"); return -1; }
    // viewSelectionSubset holds slots into the argv image list (main.cpp:1351-1384)
    std::vector<int32_t> subset(gs.cameras->viewSelectionSubset, gs.cameras->viewSelectionSubset + gs.cameras->viewSelectionSubsetNumber);
    if (!subset.empty() && tsar_set_view_subset(g_ctx, (int)subset.size(), subset.data()) != TSAR_OK) { report("tsar_set_view_subset"); return -1; }
    return 0;
}

// main.cpp:1473-1490: depth [h][w] and world normals [h][w] of the external MVS, as read from the .dmb files
void tsar_shim_set_external_planes(const cv::Mat_<float>& ref_depth, const cv::Mat_<cv::Vec3f>& ref_normal) {
    const size_t np = (size_t)ref_depth.rows * ref_depth.cols;
    g_ext_depth.resize(np);
    g_ext_normal.resize(3 * np);
    for (int y = 0; y < ref_depth.rows; y++)
        for (int x = 0; x < ref_depth.cols; x++) {
            const size_t p = (size_t)y * ref_depth.cols + x;
            g_ext_depth[p] = ref_depth(y, x);
            for (int c = 0; c < 3; c++) g_ext_normal[3 * p + c] = ref_normal(y, x)[c];
        }
}
void tsar_shim_run_patchmatch(bool on) { g_run_patchmatch = on; }

int firstcuda(GlobalState& gs) {                 // gipuma.cu:1879-1886 -> gipuma_first (:1700-1776)
    if (!g_ctx) return 0;
    if (g_run_patchmatch) {                      // gipuma_init_cu2 + the red/black loop (gipuma.cu:1741-1754, commented out upstream)
        if (tsar_pm_init(g_ctx) != TSAR_OK || tsar_pm_iterate(g_ctx, g_iterations) != TSAR_OK) report("patchmatch");
    } else {                                     // live path: gipuma_get_disp on the external planes (gipuma.cu:1755)
        if (tsar_load_planes(g_ctx, g_ext_depth.data(), g_ext_normal.data(), TSAR_MEM_HOST) != TSAR_OK) report("tsar_load_planes");
    }
    return 0;
}

int sliccuda(GlobalState& gs) {                  // gipuma.cu:1888-1895 -> gipuma_slic: gipuma_getview (:1806)
    if (!g_ctx) return 0;
    if (tsar_set_reliable_mask(g_ctx, gs.lines->scale, TSAR_MEM_HOST) != TSAR_OK) report("tsar_set_reliable_mask");       // weak.png, main.cpp:1499-1514
    if (tsar_getview(g_ctx) != TSAR_OK) report("tsar_getview");
    // texture() filled lines->canny and cannylines->{text,size} on the host (main.cpp:559-593); the per-region CPU RANSAC
    // (main.cpp:1520-1730) runs here on the GPU and leaves the planes where the host loop would have: cannylines->norm4
    const std::vector<int32_t> labels = canny_labels_as_int(gs);
    const int n_regions = gs.cannylines->n;
    if (tsar_set_regions(g_ctx, labels.data(), n_regions, gs.cannylines->text, gs.cannylines->size, TSAR_MEM_HOST) != TSAR_OK) { report("tsar_set_regions"); return 0; }
    if (tsar_ransac_regions(g_ctx, (float*)gs.cannylines->norm4, nullptr) != TSAR_OK) report("tsar_ransac_regions");
    return 0;
}

int fakecuda(GlobalState& gs) {                  // gipuma.cu:1906-1913 -> gipuma_fake: gipuma_update_scale_2 (:1875)
    if (g_ctx && tsar_fake_depth(g_ctx, gs.lines->fakedepth, TSAR_MEM_HOST) != TSAR_OK) report("tsar_fake_depth");
    return 0;
}

int fillcuda(GlobalState& gs) {                  // gipuma.cu:1897-1904 -> gipuma_fill: gipuma_update_scale + gipuma_compute_disp (:1842, :1848)
    if (!g_ctx) return 0;
    if (tsar_fill_textureless(g_ctx) != TSAR_OK) { report("tsar_fill_textureless"); return 0; }
    // main.cpp:1785-1795 reads lines->norm4 = (n_world, depth) per pixel
    const size_t np = (size_t)gs.col * gs.row;
    std::vector<float> depth(np), normal(3 * np);
    if (tsar_get_result(g_ctx, depth.data(), normal.data(), nullptr, nullptr, TSAR_MEM_HOST) != TSAR_OK) { report("tsar_get_result"); return 0; }
    for (size_t p = 0; p < np; p++) {
        gs.lines->norm4[p].x = normal[3 * p];
        gs.lines->norm4[p].y = normal[3 * p + 1];
        gs.lines->norm4[p].z = normal[3 * p + 2];
        gs.lines->norm4[p].w = depth[p];
    }
    return 0;
}

void tsar_shim_shutdown() {                      // delTexture + cudaDeviceReset of the reference (main.cpp:1098-1104, 1863)
    tsar_destroy(g_ctx);
    g_ctx = nullptr;
}
