// slic_ref.cpp — C-ABI harness around the REFERENCE's own gSLICr host/device-shared functions.
// TEST INFRASTRUCTURE ONLY (oracle/): built by `make -C oracle ref` into oracle/_ref/libslic_ref.so, in the build
// container only, straight from the sources where they lie under /root/reference:
//     g++ -std=c++14 -O2 -ffp-contract=off -DCOMPILE_WITHOUT_CUDA -I /root/reference ...
// COMPILE_WITHOUT_CUDA is the reference's own switch (ORUtils/MemoryBlock.h:7,125-225); _CPU_AND_GPU_CODE_ expands to nothing on
// a host compiler (ORUtils/PlatformIndependence.h:7-11).  No header, type or function of the reference is stubbed or restated here:
// this file only loops the reference's per-pixel functions (gSLICr_Lib/engines/gSLICr_seg_engine_shared.h:7-204) over arrays the
// way the reference's __global__ wrappers do (gSLICr_seg_engine_GPU.cu:213-258,359-379), so that tests/golden/make_slic_ref_golden.py
// can record THEIR outputs.  What it cannot reach: Update_Cluster_Center_device (GPU.cu:260-357, a __global__ with shared memory)
// and pow() as CUDA's libdevice evaluates it — the host build calls glibc's powf.
#include "gSLICr_Lib/engines/gSLICr_seg_engine_shared.h"
#include <cstdint>
#include <cstring>

using gSLICr::Vector2i;
using gSLICr::Vector4f;
using gSLICr::Vector4u;
using gSLICr::objects::spixel_info;

static_assert(sizeof(spixel_info) == 32, "spixel_info is (center 2f, color 4f, id, no_pixels)");
static_assert(sizeof(Vector4f) == 16 && sizeof(Vector4u) == 4, "packed vector types");

extern "C" {

int ref_spixel_bytes(void) { return (int)sizeof(spixel_info); }

// colour_space: 0 CIELAB, 1 XYZ, 2 RGB (gSLICr_defines.h:73-78); Cvt_Img_Space_device GPU.cu:213-221
void ref_cvt_img_space(const uint8_t* bgra, float* out, int w, int h, int color_space) {
    const Vector2i sz(w, h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            cvt_img_space_shared((const Vector4u*)bgra, (Vector4f*)out, sz, x, y, (gSLICr::COLOR_SPACE)color_space);
}
void ref_rgb2xyz(const uint8_t* bgra, float* out, int n) {
    for (int i = 0; i < n; i++) rgb2xyz(((const Vector4u*)bgra)[i], ((Vector4f*)out)[i]);
}
void ref_rgb2lab(const uint8_t* bgra, float* out, int n) {
    for (int i = 0; i < n; i++) rgb2CIELab(((const Vector4u*)bgra)[i], ((Vector4f*)out)[i]);
}
// Init_Cluster_Centers_device GPU.cu:234-245
void ref_init_cluster_centers(const float* img, void* spixels, int mw, int mh, int w, int h, int spixel_size) {
    for (int y = 0; y < mh; y++)
        for (int x = 0; x < mw; x++)
            init_cluster_centers_shared((const Vector4f*)img, (spixel_info*)spixels, Vector2i(mw, mh), Vector2i(w, h), spixel_size, x, y);
}
float ref_slic_distance(const float* pix, int x, int y, const void* center, float weight, float normalizer_xy, float normalizer_color) {
    return compute_slic_distance(*(const Vector4f*)pix, x, y, *(const spixel_info*)center, weight, normalizer_xy, normalizer_color);
}
// Find_Center_Association_device GPU.cu:247-258
void ref_find_center_association(const float* img, const void* spixels, int32_t* idx, int mw, int mh, int w, int h, int spixel_size,
                                 float weight, float max_xy_dist, float max_color_dist) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            find_center_association_shared((const Vector4f*)img, (const spixel_info*)spixels, idx, Vector2i(mw, mh), Vector2i(w, h), spixel_size,
                                           weight, x, y, max_xy_dist, max_color_dist);
}
// Finalize_Reduction_Result_device GPU.cu:359-369
void ref_finalize_reduction_result(const void* accum, void* spixels, int mw, int mh, int no_blocks_per_spixel) {
    for (int y = 0; y < mh; y++)
        for (int x = 0; x < mw; x++)
            finalize_reduction_result_shared((const spixel_info*)accum, (spixel_info*)spixels, Vector2i(mw, mh), no_blocks_per_spixel, x, y);
}
// Enforce_Connectivity_device GPU.cu:371-379
void ref_supress_local_lable(const int32_t* in, int32_t* out, int w, int h) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) supress_local_lable(in, out, Vector2i(w, h), x, y);
}

}  // extern "C"
