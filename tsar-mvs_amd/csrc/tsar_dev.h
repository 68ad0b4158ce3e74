// tsar_dev.h — device-visible scene description and the host context behind include/tsar.h.
// gfx950 only; no CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/tsar.h"

#define TSAR_LUT_LINES 32         // box <= 63: at most 32 lines of at most 32 taps
#define TSAR_LUT_TAPS 40          // 32 rounded up to whole chunks of 4 / 5 / 6 taps, plus slack
#define TSAR_LUT_MAX_CLASSES 144  // (rows + 1) KiB of LDS: what fits beside the reference window in 160 KiB
#define TSAR_MAX_SELECTED 32   // views scored per hypothesis: the reference's costVector[32] (gipuma.cu:467-468)

// One source view as the kernels read it: pose relative to the reference camera (ref = K[I|0]),
// reference cameraGeometryUtils.h:270-302 / camera.h:9-33.
struct DevView {
    float K[9];
    float R[9];
    float t[3];
    float t_abs_lo;          // min |t[r]|, max |t[r]|: bounds of the nine products t[r] n[c] of the strict homography's operand guard
    float A[9];              // K R K_ref^-1  (fast-mode homography H = A - b m^T, m = K_ref^-T n / d)
    float b[3];              // K t
    float t_abs_hi;
    const float* img;        // [h][w] float gray
    const uint32_t* quad;    // [(h+2)][(w+2)] packed 2x2 texel quads (8-bit images only), see plane_kernels.hip build_quad_kernel
    const uint2* dquad;      // [(h+2)][(w+2)] the same quads as four halfs (t00, t10 - t00, t01 - t00, t11 - t10 - t01 + t00): fast mode's converged sweeps (pm_tap_r5.h MIX), else null
};

// Reference camera block (camera.h:9-33 for cameras[REFERENCE]).
struct DevRef {
    float K[9];
    float Kinv[9];
    float Minv[9];
    float Rorig[9];
    float RorigInv[9];
    float P34[3];
    float C[3];
    float fx, f, alpha, baseline, depthMin, depthMax;
};

// Everything a kernel needs besides the state planes.  Lives in device memory; every field is
// wave-uniform, so reads become scalar loads.
struct DevScene {
    int w, h;
    int n_sel;                 // viewSelectionSubsetNumber
    int hrad, vrad;            // (box-1)/2, gipuma.cu:858-859
    int n_best, cost_comb;
    int refine_steps;          // iterations of the deltaZ loop, gipuma.cu:644
    int quad_pitch;            // w + 2
    int use_quad;              // all views are integral 0..255 -> 1-load bilinear taps
    float min_disp, max_disp;
    uint32_t flags;
    uint32_t seed_lo, seed_hi;
    int k_sparse;              // every view's K is (fx 0 cx; 0 fy cy; 0 0 1) and K_ref^-1 has the same zero / one pattern (no skew):
                               // the strict homography then skips the products with those zeros (plane_homography, tsar_device_math.h)
    DevRef ref;
    int sel[TSAR_MAX_VIEWS];   // view indices in pair.txt order
    DevView view[TSAR_MAX_VIEWS];
    // Shared weight table of the general-window tap loop (pm_core.h view_cost_lut; 8-bit imagery): the bilateral weight
    // exp(-sqrt(i^2 + j^2) / 50 - |r - centre| / 18) of a tap depends on its distance class (the distinct i^2 + j^2 of the
    // window) and on an integer 0..255, so a workgroup keeps one 256-entry row per class in LDS instead of S weights per thread.
    int lut_classes;                               // rows of the table; row lut_classes is all zero (padding taps of a line's last chunk)
    int lut_row_major;                             // 1: lines of the walk are window rows (fast mode), 0: window columns (the oracle's order)
    int lut_chunk;                                 // taps per chunk of a line: 4, 5 or 6 (the fewest padding slots)
    int lut_pad_taps;                              // taps per line rounded up to whole chunks: the stride of tap_row
    int lut_d2[TSAR_LUT_MAX_CLASSES];              // i^2 + j^2 per class
    uint32_t tap_row[TSAR_LUT_LINES * TSAR_LUT_TAPS + 8];   // [line * lut_pad_taps + tap of the line] -> byte offset of the tap's row in the
                                                   // table (padding slots -> the zero row); chunks are consecutive: a walk reads it linearly
};

// State planes of one ping-pong buffer (linestate.h:12-13).
struct PlaneBuf {
    float* c;       // [h][w]
    float4* n4;     // [h][w] (n_x, n_y, n_z, d), n.X + d = 0 in reference-camera coordinates
};

struct KernelTimer {
    std::string name;
    int launches = 0;
    float total_ms = 0.f;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// Device scratch of the operators that need temporaries (weak-texture detection, region RANSAC): one arena per context, grown to
// the largest call seen and kept, so that a worker refining view after view (tsar_gipuma --all --mode=tsar) does not pay ~18
// hipMalloc + hipFree per operator and view (26 of the 59 ms of a RANSAC call at 24 MP were these).
struct ScratchArena {
    char* base = nullptr;
    size_t cap = 0;
    size_t wanted = 0;      // the largest call that did not fit so far: the arena is (re)built the second time one overflows, so a
                            // context that makes each call once (one view per process) never pays for an arena it would not reuse
};

struct tsar_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    tsar_params params{};
    bool have_params = false, have_views = false, have_state = false;
    int w = 0, h = 0, n_views = 0;
    DevScene hscene{};          // host copy
    DevScene* dscene = nullptr;  // device copy
    std::vector<float*> img;     // device images (owned)
    std::vector<uint32_t*> quad;
    std::vector<uint2*> dquad;   // half-float difference textures (fast mode, box-11 loop, buffer gathers), see DevView
    PlaneBuf buf[2]{};           // buf[0] is the canonical state outside tsar_pm_iterate
    float *ratio = nullptr, *depth = nullptr, *scale = nullptr, *lrdiff = nullptr, *confid = nullptr, *fakedepth = nullptr;
    int32_t *beview = nullptr, *canny = nullptr;
    float4* out4 = nullptr;      // result of compute_disp: (n_world, depth)
    bool have_out = false;
    bool cost_consistent = false;   // c[p] is the multi-view score of n4[p] for every pixel (init / sweep produced the state)
    // regions (cannylines)
    int n_regions = 0;
    float *region_text = nullptr, *region_size = nullptr;
    float4* region_n4 = nullptr;
    int exact_sqrt_probe = 0;    // the cost tail's square root (sqrt_rsq_exact) on THIS device: 0 not probed yet, 1 holds, -1 failed
    int exact_div_probe = 0;     // strict mode's short exact division on THIS device: 0 not probed yet, 1 holds, -1 failed (probe_exact_divide)
    int sweeps_done = 0;         // RNG stream counter
    // Propagation memo (pm_sweep_impl.h SweepMemo): per pixel the eight candidates of its previous propagation launch, the number of
    // that launch, and the number of the last launch that changed the pixel's plane.  A candidate that is the same neighbour as last
    // time, with a plane unchanged since, was scored at this pixel then and rejected (or taken and since improved on): the pixel's cost
    // never rises, so it is rejected again and need not be scored — the reference's results, bit for bit, with fewer evaluations.
    // Valid among the launches of ONE tsar_pm_iterate call only (nothing else touches the state in between).
    int32_t* memo_cand = nullptr;     // [h][w][8]
    uint32_t* memo_seq = nullptr;     // [h][w]
    uint32_t* changed_seq = nullptr;  // [h][w]
    uint32_t launch_seq = 0;          // sweep launches of this context so far (never reset)
    uint32_t memo_valid_from = 1;     // memos written before this launch are void
    int call_launch = 0;              // launches since the current tsar_pm_iterate call began
    int memo_mode = 1;                // TSAR_MEMO=0: off
    int compact_from = 6;             // TSAR_COMPACT_FROM=n (-1: never): from launch n of a call on, a wave packs its surviving hypotheses (pm_sweep_impl.h)
    const float* final_text = nullptr;   // device lines->text while tsar_pm_iterate_final runs (the kernels' `final` mode), else null
    // timing
    int variant = 2;             // TSAR_VARIANT=n: code-generation variant of the fast-mode tap loop (pm_core.h view_cost); tsar_create picks 250 (med3/fract + D16 window loads + clamp-free loop for in-image windows + wave priority + SGPR-pinned texture base and line-top weight loads + row-wise window walk in fast mode; strict mode runs it as 122, the oracle's column order) when the D16 probe passes, else 114
    bool mix_gather = true;      // TSAR_MIX_GATHER=0: keep the byte texture for the buffer-load launches too (pm_tap_r5.h MIX off)
    bool buffer_gather = true;   // TSAR_BUFFER_GATHER=0: the fast tap loop's gathers as global loads + a shift instead of structured buffer loads
    int strip_w = -1;            // TSAR_STRIP=n: width in tiles of the strips the sweep walks (pm_core.h strip_tile), 0 = row-major,
                                 // -1 = automatic: one vertical band of the image per XCD (see strip_width)
    // The remaining environment knobs (DESIGN.md §4 lists them all).  Everything is read ONCE, by tsar_create (tsar_api.hip
    // read_knobs): a context never looks at the environment again, and no knob is latched in a function-local static.
    int buffer_from = 1;         // TSAR_BUFFER_FROM=n: fast mode's sweeps use buffer-load gathers on the difference texture from sweep n of a run on (default 1: only the first sweep, whose planes are random, keeps global loads on the byte texture; measured 0 / 1 / 2 -> 37.19 / 35.33 / 35.44 ms mean sweep, profiles/r04); strict mode from max(n, 2)
    int force_block = 0;         // TSAR_BLOCK=128|256: force the sweep's workgroup shape (0: by image size, SWEEP_SMALL_IMAGE_TILES)
    int lut_mode = 1;            // TSAR_LUT=0: one-tap loop for windows other than box 11; 2: box 11 through the general-window loop too
    int ransac_wgs = 8;          // TSAR_RANSAC_WGS: workgroups per region in RANSAC stage 2 (1 = the single-workgroup kernel)
    int ransac_chain = 8;        // TSAR_RANSAC_CHAIN=4|8|16: speculative steps per pass
    int ransac_lookahead = 0;    // TSAR_RANSAC_LOOKAHEAD=1|2|3: the history-tree kernel instead of the chain
    int ransac_poll_limit = 1 << 15;      // TSAR_RANSAC_POLL_LIMIT: polls (~0.3 us each) before a stage-2 workgroup gives up waiting
    bool ransac_cooperative = true;       // TSAR_RANSAC_COOPERATIVE=0: plain launch of the multi-workgroup stage 2
    bool ransac_force_fallback = false;   // TSAR_RANSAC_FORCE_FALLBACK=1: pre-set the give-up flag (tests of that path)
    bool trace_host = false;     // TSAR_TRACE_HOST=1: host-side steps of the refinement operators on stderr
#ifdef TSAR_EXPERIMENTS
    size_t lds_pad = 0;          // TSAR_LDS_PAD=n: unused LDS per sweep workgroup (occupancy experiments)
#endif
    bool timing = false;
    std::vector<KernelTimer> timers;
    ScratchArena scratch;
};

// One operator call's view of the arena: alloc() hands out 256-byte-aligned pieces; what does not fit is a plain hipMalloc for
// this call; the second time a call overflows, the arena is re-sized to the largest total seen, and calls of that size allocate
// nothing from then on.
struct ScratchScope {
    tsar_ctx* ctx;
    size_t used = 0, need = 0;
    std::vector<void*> extra;
    explicit ScratchScope(tsar_ctx* c) : ctx(c) {}
    void* alloc(size_t bytes) {
        bytes = ((bytes ? bytes : 4) + 255) & ~(size_t)255;
        need += bytes;
        if (used + bytes <= ctx->scratch.cap) { void* p = ctx->scratch.base + used; used += bytes; return p; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        extra.push_back(p);
        return p;
    }
    void release() {                       // the stream is idle (callers synchronise first)
        for (void* p : extra) hipFree(p);
        extra.clear();
        if (need > ctx->scratch.cap) {
            const bool again = ctx->scratch.wanted > ctx->scratch.cap;          // an earlier call overflowed this arena too
            if (need > ctx->scratch.wanted) ctx->scratch.wanted = need;
            if (again) {
                if (ctx->scratch.base) hipFree(ctx->scratch.base);
                ctx->scratch.base = nullptr;
                ctx->scratch.cap = 0;
                void* p = nullptr;
                if (hipMalloc(&p, ctx->scratch.wanted) == hipSuccess) { ctx->scratch.base = (char*)p; ctx->scratch.cap = ctx->scratch.wanted; }
            }
        }
        used = need = 0;
    }
    ~ScratchScope() { release(); }
};

#define TSAR_HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return TSAR_ERR_HIP;                                                                      \
        }                                                                                             \
    } while (0)

// RAII bracket that records a hipEvent pair around a launch when timing is enabled.
struct ScopedKernelTimer {
    tsar_ctx* ctx;
    int ti = -1;                // index into ctx->timers (timers may nest: the vector can grow between constructor and destructor)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ScopedKernelTimer(tsar_ctx* c, const char* name);
    ~ScopedKernelTimer();
};

// ---- launchers implemented in the .hip files -----------------------------------------------------
int launch_build_quad(tsar_ctx* ctx, const float* img, uint32_t* quad, int w, int h, int* nonintegral_flag);
int launch_expand_u8(tsar_ctx* ctx, const uint8_t* in, float* out, size_t n);   // plane_kernels.hip
int launch_build_dquad(tsar_ctx* ctx, const uint32_t* quad, uint2* dquad, int w, int h);
int launch_pm_init(tsar_ctx* ctx);
bool probe_d16_hi_zeroes(tsar_ctx* ctx);   // pm_sweep.hip
int launch_pm_sweep(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out,
                    uint32_t stream_id, int do_prop, int do_refine);
int launch_pm_sweep_lut(tsar_ctx* ctx, int need, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out, uint32_t stream_id,
                        int do_prop, int do_refine);       // pm_sweep_lut.hip
int launch_pm_full_lut(tsar_ctx* ctx, int need, bool init, const float4* planes, float* c, float4* n, int32_t* bv, float* rt);   // pm_init_lut.hip
int lut_chunk_taps(int taps_per_line);
// the general-window tap loop serves 8-bit imagery (quad textures), both filter modes, whose window has few enough
// distance classes for the LDS table; TSAR_LUT=0 switches it off (the one-tap-at-a-time loop then runs), TSAR_LUT=2 also sends
// the box-11 / two-best-views configuration through it instead of its own tap loop: A/B measurements
static inline bool lut_path_forced(const tsar_ctx* ctx) { return ctx->lut_mode == 2; }
static inline bool lut_path_applies(const tsar_ctx* ctx) {
    const bool off = ctx->lut_mode == 0;
    const DevScene& hs = ctx->hscene;
    // (its fast-mode loop loads window texels with ds_read_u16_d16_hi: only where tsar_create's probe found the register's
    // other half zeroed — variant bit 3)
    const bool d16_ok = (hs.flags & TSAR_FLAG_STRICT_DIV) || (ctx->variant & 8);
    return !off && d16_ok && hs.use_quad && hs.lut_classes > 0;
}
int launch_pm_sweep_experiment(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out, uint32_t stream_id,
                               int do_prop, int do_refine, int* launched);   // pm_sweep_experiments.hip (TSAR_EXPERIMENTS builds)
int launch_pm_cost_planes(tsar_ctx* ctx, const float4* planes, float* cost, int32_t* beview, float* ratio);
int launch_get_disp(tsar_ctx* ctx, const float* depth_in, const float* normal_world);
int launch_compute_disp(tsar_ctx* ctx);
int launch_compute_disp_final(tsar_ctx* ctx, const float4* resize4, const float* text);
int launch_depth_to_plane(tsar_ctx* ctx);
int launch_getview(tsar_ctx* ctx);
int launch_lrdiff(tsar_ctx* ctx);
int launch_selftest_divide(tsar_ctx* ctx, const float* X, const float* Y, const float* Z, size_t n, float* u, float* v, int ieee);   // selftest_kernels.hip
int launch_sweep_census(tsar_ctx* ctx, int colour, unsigned long long* dout);
int launch_selftest_divide_random(tsar_ctx* ctx, int log2_pairs, uint64_t seed, int mode, int guarded, unsigned long long* dcounts);
int launch_selftest_sqrt(tsar_ctx* ctx, int mode, uint64_t seed, unsigned long long* dcounts);
int launch_update_scale(tsar_ctx* ctx);
int launch_fake_depth(tsar_ctx* ctx);
int launch_split_out4(tsar_ctx* ctx, float* depth, float* normal3);
int launch_label_range(tsar_ctx* ctx, const int32_t* labels, size_t n, int32_t* lo, int32_t* hi);   // min / max of a device label plane
