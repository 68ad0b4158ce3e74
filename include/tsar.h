/*
 * tsar.h — C ABI of the MI355X-native PatchMatch-MVS matcher (libtsar_hip.so).
 *
 * This is the drop-in boundary for the GPU operator layer of ZhenlongYuan/TSAR-MVS.
 * The reference crosses host→device at four C++ functions taking a CUDA-managed
 * `GlobalState&` (reference gipuma.h:2-6, bodies gipuma.cu:1700-1913) plus the gSLICr
 * `core_engine` (reference gSLICr_Lib/engines/gSLICr_core_engine.h:19-32).  A managed-memory
 * C++ object cannot cross an FFI, so the same operators are exported here as plain
 * `extern "C"` functions over an opaque context, plain pointers and sizes:
 *
 *   reference operator (file:line)                      this header
 *   --------------------------------------------------  ---------------------------------------
 *   GlobalState ctor + LineState::resize                 tsar_create / tsar_destroy
 *     (globalstate.h:41-53, linestate.h:71-110)
 *   getCameraParameters + addImageToTextureFloatGray     tsar_set_views
 *     (cameraGeometryUtils.h:174-364, main.cpp:1190-1228)
 *   AlgorithmParameters fill (main.cpp:1386-1416)        tsar_set_params
 *   viewSelectionSubset fill (main.cpp:1351-1384)        tsar_set_view_subset
 *   gipuma_init_cu2            (gipuma.cu:678-729)       tsar_pm_init
 *   red/black prop+refine loop (gipuma.cu:1744-1754,     tsar_pm_iterate
 *     bodies :846-1138)
 *   the same kernels with `final == true`                tsar_pm_iterate_final
 *     (gipuma.cu:856,1063,559-562,669-672; lines->text)
 *   pmCostMultiview_cu on a given plane map              tsar_pm_cost_planes (test / diagnostics hook)
 *     (gipuma.cu:455-518)
 *   host fill of norm4/depth/c + firstcuda               tsar_load_planes
 *     (main.cpp:1479-1493, gipuma_get_disp gipuma.cu:731-755)
 *   weak.png → lines->scale (main.cpp:1499-1514)         tsar_set_reliable_mask / tsar_get_reliable_mask
 *   gipuma_getlrdiff           (gipuma.cu:1160-1186)     tsar_lrdiff
 *   sliccuda → gipuma_getview  (gipuma.cu:1188-1213)     tsar_getview
 *   gipuma_WMF / gipuma_WMF_Final (gipuma.cu:1294-1698)  tsar_wmf
 *   texture() output canny[]/text[] (main.cpp:559-593)   tsar_set_regions
 *   texture() itself (main.cpp:365-596, CPU + OpenCV)     tsar_detect_weak_texture
 *   CPU RANSAC per region      (main.cpp:1520-1730)      tsar_ransac_regions
 *   fakecuda → gipuma_update_scale_2 (gipuma.cu:1261-92) tsar_fake_depth
 *   fillcuda → gipuma_update_scale + gipuma_compute_disp tsar_fill_textureless
 *     (gipuma.cu:1215-1259, 810-844)
 *   gipuma_compute_disp alone  (gipuma.cu:810-844)       tsar_compute_disp
 *   gipuma_compute_disp_final  (gipuma.cu:757-808)       tsar_compute_disp_final
 *   gipuma_dptow               (gipuma.cu:1140-1158)     tsar_depth_to_plane
 *   copy-out of norm4 (main.cpp:1785-1795)               tsar_get_result / tsar_get_plane
 *   gSLICr core_engine::Process_Frame + Get_Seg_Res      tsar_slic
 *   Fusion.exe (binary only; flags x/1.sh:20-30)         tsar_fuse
 *
 * Conventions
 *   - every function returns an int status (TSAR_OK = 0, negative = error) and never exits the
 *     process (the reference's checkCudaErrors calls exit(), helper_cuda.h);
 *     tsar_last_error() returns a human-readable message for the last failure on that context.
 *   - the caller owns every buffer it passes; the library owns all device memory inside tsar_ctx.
 *   - `mem` arguments say where caller buffers live: TSAR_MEM_HOST or TSAR_MEM_DEVICE (HIP device
 *     pointer on the context's device).  No unified memory.
 *   - one tsar_ctx per device and per host thread; all work of a context is issued on one HIP
 *     stream (tsar_get_stream) and the call returns after that work is complete unless the
 *     function says it is asynchronous.
 *   - TSAR_MEM_DEVICE inputs are read on the context's own (non-blocking) stream, which is not ordered against
 *     any other stream: the work that produces them (a kernel on another stream, an RCCL collective, a copy)
 *     must be COMPLETE before the call, e.g. by synchronising the producing stream or by making it wait
 *     on an event the caller then synchronises.  TSAR_MEM_DEVICE outputs are complete when the call returns.
 *   - host buffers from tsar_host_alloc are page-locked: copies to and from them run at PCIe rate without the
 *     runtime's bounce buffers.  Any other host memory works too, slower.
 *   - images are row-major float32 gray, values as produced by an 8-bit decode (0..255); planes are
 *     row-major float32 [h][w] (or [h][w][3]/[h][w][4]).
 */
#ifndef TSAR_H_
#define TSAR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSAR_MAX_VIEWS 64        /* reference MAX_IMAGES 512 (config.h:2); ≤32 views are ever selected */
#define TSAR_MAXCOST 2.0f        /* reference config.h:22 */

/* status codes */
#define TSAR_OK 0
#define TSAR_ERR_INVALID (-1)    /* bad argument */
#define TSAR_ERR_HIP (-2)        /* HIP runtime failure */
#define TSAR_ERR_STATE (-3)      /* call order violated (e.g. iterate before set_views) */
#define TSAR_ERR_NOMEM (-4)

#define TSAR_MEM_HOST 0
#define TSAR_MEM_DEVICE 1

/* cost combination, reference algorithmparameters.h:17 */
#define TSAR_COMB_ALL 0
#define TSAR_COMB_BEST_N 1
#define TSAR_COMB_ANGLE 2   /* accepted like the reference: only BEST_N is distinguished on the GPU path (gipuma.cu:496-499), */
#define TSAR_COMB_GOOD 3    /* so ANGLE and GOOD average all valid views exactly as ALL does */

/* behaviour flags (tsar_params.flags).  Default 0 = the reference's behaviour wherever it is
 * well defined. */
#define TSAR_FLAG_FIX_DOWN_FAR_SEED  (1u << 0) /* seed the down_far arm's minimum with c[down_far]
                                                  (reference seeds with c[up_far], gipuma.cu:906) */
#define TSAR_FLAG_FIX_RIGHT_FAR_CMP  (1u << 1) /* right_far arm picks the minimum
                                                  (reference comparison is inverted, gipuma.cu:943) */
#define TSAR_FLAG_STRICT_DIV         (1u << 2) /* the reference's arithmetic, operation for operation: IEEE divisions in the per-tap
                                                  perspective divide, the tap position as (m0 x + m1 y) + m2 (getCorrespondingPoint_cu
                                                  gipuma.cu:161-171), (w r) s, window columns, the reference's homography and blend
                                                  (bit-exact against the CPU oracle; ~28 % slower than the default arithmetic, whose
                                                  seven rounding-level liberties oracle/tsar_oracle.c S7 lists) */
#define TSAR_FLAG_NO_LINE_CLOSING    (1u << 4) /* tsar_detect_weak_texture: skip the Hough boundary closing of large regions
                                                  (main.cpp:385-435; on by default like the reference's HoughLinesP step) */

#define TSAR_FLAG_TEX_FILTER_8BIT    (1u << 5) /* bilinear fractions rounded to 8 fractional bits before the blend, as the CUDA
                                                  texture unit the reference samples through stores them (linear filtering with
                                                  1.8 fixed-point weights, main.cpp:1215-1219); the unit's rounding rule is
                                                  unpublished: round-to-nearest-even here.  Runs the general-window tap loop at every box
                                                  (box 11: ~15 % below the default filter). */

#define TSAR_FLAG_FIX_INIT_RADIUS     (1u << 6) /* tsar_pm_init on the sweeps' window, radius (box - 1) / 2.  The reference's
                                                  gipuma_init_cu2 uses box / 2 (gipuma.cu:693-694), every other kernel
                                                  (box - 1) / 2 (:858-859): an even box initialises on a larger window.
                                                  Reproduced by default. */

typedef struct tsar_ctx tsar_ctx;

/* One calibrated view as read from an MVSNet-style cams/%08d_cam.txt (reference
 * fileIoUtils.h:117-153): intrinsics and world→camera extrinsics.  Row-major. */
typedef struct tsar_camera {
    float K[9];
    float R[9];
    float t[3];
} tsar_camera;

/* Subset of the reference's AlgorithmParameters (algorithmparameters.h:54-88) that the GPU path
 * reads.  Zero-initialise, then tsar_default_params(). */
typedef struct tsar_params {
    int32_t box_hsize;    /* --blocksize (scripts pass 11; default 19).  1..63.  tsar_set_views refuses (TSAR_ERR_INVALID, with the
                             reason in tsar_last_error) a box above 23 when the shared weight table of the general-window loop
                             cannot serve it: images that are not an 8-bit decode; rectangular boxes with more than 144 distinct
                             tap distances (radii of mixed parity, e.g. 63 x 61); fast mode on a device whose D16 LDS-load probe
                             failed.  An even box initialises on radius box / 2 and sweeps on (box - 1) / 2 like the reference
                             (TSAR_FLAG_FIX_INIT_RADIUS); a context holding the reference view alone takes any box. */
    int32_t box_vsize;
    int32_t n_best;       /* --n_best (scripts 1; default 2) */
    int32_t cost_comb;    /* --cost_comb: TSAR_COMB_* */
    float depth_min;      /* from the reference view's cam file */
    float depth_max;
    float cam_scale;      /* --cam_scale: K is divided by it (cameraGeometryUtils.h:143-154) */
    uint32_t flags;       /* TSAR_FLAG_* */
    uint64_t seed;        /* RNG stream seed (the reference seeds with clock64(), gipuma.cu:700) */
} tsar_params;

/* gSLICr settings actually set by the reference (main.cpp:608-615). */
typedef struct tsar_slic_settings {
    int32_t spixel_size;      /* 20 */
    int32_t no_iters;         /* 5 */
    float coh_weight;         /* 5.0 */
    int32_t do_enforce_connectivity; /* 0 in the reference */
    int32_t color_space;      /* 0 = CIELAB (reference), 1 = XYZ, 2 = RGB */
} tsar_slic_settings;

/* Per-kernel timing record (tsar_get_kernel_timing). */
typedef struct tsar_kernel_timing {
    char name[48];
    int32_t launches;
    float total_ms;           /* sum of hipEventElapsedTime over the launches */
} tsar_kernel_timing;

/* ---- lifecycle ------------------------------------------------------------------------- */
int tsar_create(int device, tsar_ctx** out);
int tsar_destroy(tsar_ctx* ctx);
const char* tsar_last_error(const tsar_ctx* ctx);
const char* tsar_version(void);
/* the HIP stream (hipStream_t) all kernels of this context are launched on */
int tsar_get_stream(tsar_ctx* ctx, void** stream_out);
int tsar_synchronize(tsar_ctx* ctx);

/* ---- inputs ---------------------------------------------------------------------------- */
void tsar_default_params(tsar_params* p);
int tsar_set_params(tsar_ctx* ctx, const tsar_params* p);
/* views[0] is the reference view, views[1..n-1] the source views, in the order of the reference's
 * argv image list.  gray[i] points at w*h float32.  Cameras are re-origined so that the reference
 * camera is K[I|0] (cameraGeometryUtils.h:270-302).  Must follow tsar_set_params.
 * n_views = 1 (the reference view alone) is enough for the textureless-refinement operators (tsar_load_planes, weak-texture
 * detection, region RANSAC, fill — none of them reads a source image); the matching entry points (tsar_pm_*, tsar_lrdiff)
 * then return TSAR_ERR_STATE. */
int tsar_set_views(tsar_ctx* ctx, int n_views, int w, int h, const float* const* gray, int mem,
                   const tsar_camera* cams);
/* The same with the 8-bit decode itself: gray[i] points at w*h bytes, widened to float on the device — bit for bit what
 * tsar_set_views does with (float)gray[i][p], which is how the reference fills its textures (imread(..., GRAYSCALE) ->
 * convertTo(CV_32F), main.cpp:1302,1423).  A quarter of the bytes cross PCIe and the caller holds no float copies of its images
 * (1.07 GB at ETH3D size with ten sources): what tsar_gipuma hands over. */
int tsar_set_views_u8(tsar_ctx* ctx, int n_views, int w, int h, const uint8_t* const* gray, int mem,
                      const tsar_camera* cams);
/* indices (1..n_views-1) of the source views used for matching, in pair.txt order; at most 32 (the reference's
 * costVector[32], gipuma.cu:467).  Default after tsar_set_views: the first min(n_views - 1, 32) source views.
 * Changing the subset invalidates the stored costs' meaning: the next sweep re-scores neighbours it would otherwise skip. */
int tsar_set_view_subset(tsar_ctx* ctx, int n, const int32_t* view_idx);

/* ---- PatchMatch (the north-star path) --------------------------------------------------- */
int tsar_pm_init(tsar_ctx* ctx);
/* `iters` red/black iterations; each = black (prop+refine) then red (prop+refine). */
int tsar_pm_iterate(tsar_ctx* ctx, int iters);
/* The same loop with the kernels' `final` argument true (dormant in the reference: nothing passes true).
 * text [h][w] = lines->text: pixels with text == -1 keep their plane and cost (gipuma.cu:856, :1063) and
 * accepted hypotheses do not update ratio / best view (gipuma.cu:559-562, :669-672). */
int tsar_pm_iterate_final(tsar_ctx* ctx, int iters, const float* text, int mem);
/* Diagnostics: multi-view cost of caller-supplied planes.  planes = [h][w][4] (n_x,n_y,n_z,d) in
 * reference-camera coordinates; outputs [h][w]; beview/ratio may be NULL. */
int tsar_pm_cost_planes(tsar_ctx* ctx, const float* planes, int mem, float* cost_out,
                        int32_t* beview_out, float* ratio_out);
/* Diagnostics: overwrite / read the raw matcher state: planes [h][w][4] + cost [h][w]. */
int tsar_set_plane(tsar_ctx* ctx, const float* planes, const float* cost, int mem);
/* Diagnostics: ONE half-iteration (colour 0 = black: (x+y) even, 1 = red), optionally only its
 * propagation or only its refinement half (the reference's four kernels gipuma.cu:1096-1138), and
 * the RNG stream counter (number of half-iterations done so far) that keys the refinement draws. */
int tsar_pm_sweep(tsar_ctx* ctx, int colour, int do_prop, int do_refine);
int tsar_set_sweep_counter(tsar_ctx* ctx, int n);
int tsar_get_plane(tsar_ctx* ctx, float* planes, float* cost, int32_t* beview, float* ratio, int mem);

/* ---- plane <-> depth (reference gipuma.cu:731-844, 1140-1158) ---------------------------- */
/* depth [h][w], normal_world [h][w][3]: planes from an external MVS; cost is set to 1. */
int tsar_load_planes(tsar_ctx* ctx, const float* depth, const float* normal_world, int mem);
int tsar_compute_disp(tsar_ctx* ctx);
int tsar_compute_disp_final(tsar_ctx* ctx, const float* resize_planes, const float* text, int mem);
int tsar_depth_to_plane(tsar_ctx* ctx);
/* After tsar_compute_disp: depth [h][w] (0 where cost == MAXCOST), normal_world [h][w][3],
 * cost [h][w], confid [h][w]; any may be NULL. */
int tsar_get_result(tsar_ctx* ctx, float* depth, float* normal_world, float* cost, float* confid,
                    int mem);

/* ---- TSAR textureless refinement (reference gipuma.cu:1160-1698, main.cpp:1499-1783) ------ */
int tsar_set_reliable_mask(tsar_ctx* ctx, const float* scale, int mem);          /* lines->scale */
int tsar_get_reliable_mask(tsar_ctx* ctx, float* scale, int mem);                /* lines->scale as tsar_wmf leaves it */
int tsar_lrdiff(tsar_ctx* ctx);
int tsar_getview(tsar_ctx* ctx);
int tsar_wmf(tsar_ctx* ctx, int iters, int final_pass);
/* labels [h][w] = region id per pixel (lines->canny); region_text[n_regions] = -1 for textureless
 * regions (cannylines->text).  Every label must lie in [0, n_regions): checked, TSAR_ERR_INVALID otherwise. */
int tsar_set_regions(tsar_ctx* ctx, const int32_t* labels, int n_regions, const float* region_text,
                     const float* region_size, int mem);
/* Weak-texture region detection of the reference view on the GPU (reference texture(), main.cpp:365-596):
 * computes lines->canny / cannylines->text / size and installs them like tsar_set_regions.  labels_out
 * [h][w] int32, text_out/size_out [cap] may be NULL.  The boundary closing of large regions (main.cpp:385-435) runs a
 * deterministic Hough transform with the reference's parameters in place of OpenCV's randomised HoughLinesP, whose
 * arithmetic is not in the reference's sources (parity unpinned for that step; TSAR_FLAG_NO_LINE_CLOSING skips it). */
int tsar_detect_weak_texture(tsar_ctx* ctx, int32_t* labels_out, int mem, int* n_regions_out, float* text_out,
                             float* size_out, int cap);
/* GPU replacement of the per-region CPU RANSAC; region_planes_out [n_regions][4] may be NULL */
int tsar_ransac_regions(tsar_ctx* ctx, float* region_planes_out, float* inlier_ratio_out);
int tsar_set_region_planes(tsar_ctx* ctx, const float* region_planes);            /* host [n][4] */
int tsar_fake_depth(tsar_ctx* ctx, float* fakedepth_out, int mem);
int tsar_fill_textureless(tsar_ctx* ctx);

/* ---- gSLICr superpixels ------------------------------------------------------------------ */
void tsar_default_slic_settings(tsar_slic_settings* s);
/* bgra: [h][w][4] uint8 (the reference feeds a 1/4-resolution BGR image, main.cpp:617-640);
 * labels_out [h][w] int32. */
int tsar_slic(tsar_ctx* ctx, const uint8_t* bgra, int w, int h, const tsar_slic_settings* s,
              int32_t* labels_out, int mem);

/* ---- depth-map fusion (row N3; the reference ships it only as Fusion.exe, flags x/1.sh:20-30) ----- */
typedef struct tsar_fusion_params {
    int32_t num_consistent;   /* --num_consistent=  (scripts: 1) */
    float reproj_error;       /* --reproj_error=    (2 px) */
    float depth_diff;         /* --depth_diff=      (0.01 relative) */
    float angle_deg;          /* --angle=           (15 degrees between normals) */
    int32_t used_list;        /* --used_list=       (1: pixels that contributed to a point are not fused again) */
} tsar_fusion_params;
void tsar_default_fusion_params(tsar_fusion_params* p);
/* Fuses n_views depth/normal maps (what tsar_get_result exports: depth [h][w], world normals [h][w][3]) into
 * one point cloud.  cams: K and world->camera R, t per view; gray: the views' images (point colour).  The
 * source views of view v are src_idx[src_off[v] .. src_off[v+1]) (pair.txt as CSR).  points_out: up to `cap`
 * records of 9 floats (x y z, nx ny nz, gray, number of agreeing views, reference view), in view order then
 * raster order; *n_points_out is the number found (may exceed cap).
 * tsar_fuse_ctx runs on the context's device and stream and takes every temporary from the context's scratch arena (no device
 * allocation from the second call of a size on); tsar_fuse is the context-free form for a one-shot fuser process: it creates a
 * context on `device` for the duration of the call. */
int tsar_fuse_ctx(tsar_ctx* ctx, int n_views, int w, int h, const tsar_camera* cams, const float* const* depth,
                  const float* const* normal_world, const float* const* gray, int mem, const int32_t* src_off,
                  const int32_t* src_idx, const tsar_fusion_params* params, float* points_out, int64_t cap,
                  int64_t* n_points_out);
int tsar_fuse(int device, int n_views, int w, int h, const tsar_camera* cams, const float* const* depth,
              const float* const* normal_world, const float* const* gray, int mem, const int32_t* src_off,
              const int32_t* src_idx, const tsar_fusion_params* params, float* points_out, int64_t cap,
              int64_t* n_points_out);

/* ---- page-locked host buffers --------------------------------------------------------------- */
/* NULL on failure.  Replaces the reference's cudaMallocManaged host-visible planes (managed.h:7-15) on the host side of the
 * boundary: the caller's image / result buffers, allocated here, are DMA targets. */
void* tsar_host_alloc(size_t bytes);
void tsar_host_free(void* p);

/* ---- device buffers for a multi-GPU host -------------------------------------------------------- */
/* Plain device memory on `device` (NULL on failure) and a synchronous device-to-device copy between two devices of the node
 * (xGMI peer copy; also valid with dst_device == src_device).  tsar_gipuma --all --fuse keeps each view's result on the GPU
 * that matched it and gathers them to the fusing GPU with these, where the reference's pipeline goes through files
 * (scripts/courtyard.sh:29-48, x/1.sh:30). */
void* tsar_device_alloc(int device, size_t bytes);
void tsar_device_free(int device, void* p);
int tsar_device_write(int device, void* dst, const void* host_src, size_t bytes);     /* synchronous host -> device copy */
int tsar_peer_copy(int dst_device, void* dst, int src_device, const void* src, size_t bytes);

/* ---- measurement ------------------------------------------------------------------------- */
/* When enabled every kernel launch is bracketed by hipEvents on the context's stream. */
int tsar_enable_kernel_timing(tsar_ctx* ctx, int enable);
int tsar_reset_kernel_timing(tsar_ctx* ctx);
/* fills up to `cap` records, returns the number of distinct kernels in *n_out */
int tsar_get_kernel_timing(tsar_ctx* ctx, tsar_kernel_timing* out, int cap, int* n_out);

/* ---- self-tests ---------------------------------------------------------------------------- */
/* TSAR_FLAG_STRICT_DIV's perspective divide u = X / Z, v = Y / Z (getCorrespondingPoint_cu gipuma.cu:161-171, vecdiv4) is computed
 * with one v_rcp_f32 + Newton step and one residual correction per quotient instead of the compiler's IEEE division sequence, behind
 * an operand guard that falls back to the latter.  These run that code path on caller-supplied or device-generated operands so a
 * test can compare it with IEEE division (the host's `/`, or the device's) bit for bit.
 *   tsar_selftest_divide: host arrays in / out; ieee = 1 returns the device's IEEE quotients instead, ieee = 2 the fast mode's
 *   X * v_rcp_f32(Z) (with X = 1: the device's reciprocal itself, which the CPU oracle's restatement of the fast arithmetic reads).
 *   tsar_selftest_divide_random: 2^log2_triples (X, Y, Z) triples generated on the device (mode 0: like the tap loop's operands;
 *   1: any mantissa / sign, exponents across the guard range; 2: any bit pattern); guarded = 0 runs the form without the guard (the
 *   clamp-free tap loops; modes 0 and 1).  Returns the number of quotients that differ from `/` and of triples outside the guard. */
/* tsar_selftest_sqrt: the square root of the matching cost's tail (tsar_device_math.h sqrt_rsq_exact: v_rsq_f32 + one fused residual
 * correction instead of the compiler's IEEE sequence) against sqrtf on the device; mode 0 = every mantissa of two adjacent binades
 * (both exponent parities, 2^24 inputs: the enumeration), mode 1 = 2^24 random inputs with exponents across [2^-100, 2^100], mode 2 =
 * the control (the same inputs as mode 0 without the correction step: must report mismatches), mode 3 = 2^24 random mantissas spread over
 * the 67 binades the cost tail's operands can reach ([1e-10, 4.3e9]).  tsar_set_views runs modes 0 and 3 once per context before it
 * accepts 8-bit imagery, and REFUSES the views (TSAR_ERR_HIP) on a mismatch; there is no fallback to sqrtf. */
int tsar_selftest_sqrt(tsar_ctx* ctx, int mode, uint64_t seed, uint64_t* mismatches_out);
int tsar_selftest_divide(tsar_ctx* ctx, const float* X, const float* Y, const float* Z, size_t n, float* u_out, float* v_out, int ieee);
int tsar_selftest_divide_random(tsar_ctx* ctx, int log2_triples, uint64_t seed, int mode, int guarded, uint64_t* mismatches_out,
                                uint64_t* outside_guard_out);
/* Census of the propagation arms of the NEXT half-sweep of `colour` on the current state (nothing is modified): how many multi-view
 * evaluations the wave-uniform hypothesis loop of the sweep kernel runs, against what lane-local candidate queues would run
 * (gipuma.cu:553-555 early-outs; selftest_kernels.hip documents the eight counters). */
int tsar_selftest_sweep_census(tsar_ctx* ctx, int colour, uint64_t* out8);
/* Census behind the propagation memo: for the half-sweep of `colour` about to run, how many alive arms carry the plane the pixel
 * tried in the previous call of this function (memo_dev: w * h * 8 uint64 on the device, zeroed before the first call, updated by
 * each).  out8: [0] alive arms, [1] repeating the same arm's plane, [2] any arm's, [3] (wave, arm) pairs with an alive lane,
 * [4] of those, pairs in which every alive lane repeats, [5] sum over waves of max-over-lanes alive arms, [6] ... of fresh ones,
 * [7] sum over waves of ceil(fresh pairs of the wave / 64) = the propagation trips of the packed form. */
int tsar_selftest_sweep_repeat(tsar_ctx* ctx, int colour, void* memo_dev, uint64_t* out8);
/* One stage of tsar_slic on caller-supplied HOST arrays, so that a test can hold every SLIC kernel to the outputs of the reference's
 * own per-pixel functions (gSLICr_seg_engine_shared.h:7-204, host-compiled from the reference where it lies: tests/golden/slic_ref.npz).
 * Centres are 32-byte records laid out like the reference's spixel_info (gSLICr_spixel_info.h:11-17: center 2 f32, color_info
 * 4 f32, id i32, no_pixels i32).  Uses s->spixel_size, s->coh_weight, s->color_space.
 *   stage 0  Cvt_Img_Space:            in0 = bgra u8[h*w*4]                          inout = float4[h*w]         (out)
 *   stage 1  Init_Cluster_Centers:     in0 = float4[h*w]                             inout = centres[mw*mh]      (out)
 *   stage 2  Find_Center_Association:  in0 = float4[h*w], in1 = centres[mw*mh]       inout = labels i32[h*w]     (in: previous, out)
 *   stage 3  Update_Cluster_Center + Finalize_Reduction_Result (mw = w / S, mh = h / S):
 *                                      in0 = float4[h*w], in1 = labels i32[h*w]      inout = centres[mw*mh]      (out)
 *   stage 4  Enforce_Connectivity, one pass: in0 = labels i32[h*w]                   inout = labels i32[h*w]     (out) */
int tsar_selftest_slic_stage(tsar_ctx* ctx, int stage, int w, int h, int mw, int mh, const tsar_slic_settings* s, const void* in0,
                             const void* in1, void* inout);

#ifdef __cplusplus
}
#endif
#endif /* TSAR_H_ */
