// tsar_api.hip — the C ABI of include/tsar.h: context, device memory, camera algebra, call order.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "tsar_dev.h"

// ---- kernel timing -------------------------------------------------------------------------------
ScopedKernelTimer::ScopedKernelTimer(tsar_ctx* c, const char* name) : ctx(c) {      // name == nullptr: no record
    if (!ctx->timing || !name) return;
    for (size_t k = 0; k < ctx->timers.size(); k++)
        if (ctx->timers[k].name == name) ti = (int)k;
    if (ti < 0) {
        ctx->timers.emplace_back();
        ctx->timers.back().name = name;
        ti = (int)ctx->timers.size() - 1;
    }
    if (hipEventCreate(&e0) != hipSuccess) { ti = -1; return; }
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); ti = -1; return; }
    hipEventRecord(e0, ctx->stream);
}
ScopedKernelTimer::~ScopedKernelTimer() {
    if (ti < 0) return;
    hipEventRecord(e1, ctx->stream);
    ctx->timers[ti].pending.emplace_back(e0, e1);
}
static void drain_timers(tsar_ctx* ctx) {
    for (auto& k : ctx->timers) {
        for (auto& pr : k.pending) {
            float ms = 0.f;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                k.total_ms += ms;
                k.launches++;
            }
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        k.pending.clear();
    }
}

// ---- helpers -------------------------------------------------------------------------------------
static int fail(tsar_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg;
    return code;
}
template <typename T>
static int dev_alloc(tsar_ctx* ctx, T** p, size_t n) {
    if (*p) { hipFree(*p); *p = nullptr; }
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? TSAR_ERR_NOMEM : TSAR_ERR_HIP; }
    return TSAR_OK;
}
template <typename T>
static void dev_free(T*& p) {
    if (p) hipFree(p);
    p = nullptr;
}
static hipMemcpyKind in_kind(int mem) { return mem == TSAR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice; }
static hipMemcpyKind out_kind(int mem) { return mem == TSAR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost; }
#define CHECK_CTX(ctx) \
    if (!(ctx)) return TSAR_ERR_INVALID; \
    if (hipSetDevice((ctx)->device) != hipSuccess) return fail(ctx, TSAR_ERR_HIP, "hipSetDevice failed")
#define NEED_VIEWS(ctx) if (!(ctx)->have_views) return fail(ctx, TSAR_ERR_STATE, "tsar_set_views has not been called")
// matching scores planes against source views; a context holding the reference view only serves the textureless-refinement
// operators (load_planes, weak-texture detection, region RANSAC, fill)
#define NEED_SOURCES(ctx) if ((ctx)->hscene.n_sel < 1) return fail(ctx, TSAR_ERR_STATE, "no source views: tsar_set_views was given the reference view only")
#define NEED_STATE(ctx) if (!(ctx)->have_state) return fail(ctx, TSAR_ERR_STATE, "no plane state: call tsar_pm_init, tsar_load_planes or tsar_set_plane first")
#define TRY(expr) do { int rc_ = (expr); if (rc_ != TSAR_OK) return rc_; } while (0)

template <typename T>
struct TmpIn {   // device view of a caller buffer (copies host buffers in)
    tsar_ctx* ctx;
    const T* d = nullptr;
    T* owned = nullptr;
    int rc = TSAR_OK;
    // scratch: take the staging buffer from the context's arena (tsar_dev.h ScratchScope) instead of a hipMalloc per call
    TmpIn(tsar_ctx* c, const T* src, size_t n, int mem, ScratchScope* scratch = nullptr) : ctx(c) {
        if (!src) return;
        if (mem == TSAR_MEM_DEVICE) { d = src; return; }
        T* staging = nullptr;
        if (scratch) {
            staging = (T*)scratch->alloc(n * sizeof(T));
            if (!staging) rc = fail(ctx, TSAR_ERR_NOMEM, "hipMalloc failed");
        } else {
            rc = dev_alloc(ctx, &owned, n);
            staging = owned;
        }
        if (rc == TSAR_OK && hipMemcpyAsync(staging, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "H2D failed");
        d = staging;
    }
    ~TmpIn() {
        if (owned) { hipStreamSynchronize(ctx->stream); hipFree(owned); }
    }
};

static void inv3(const double* m, double* o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double s = 1.0 / det;
    o[0] = c00 * s; o[1] = (m[2] * m[7] - m[1] * m[8]) * s; o[2] = (m[1] * m[5] - m[2] * m[4]) * s;
    o[3] = c01 * s; o[4] = (m[0] * m[8] - m[2] * m[6]) * s; o[5] = (m[2] * m[3] - m[0] * m[5]) * s;
    o[6] = c02 * s; o[7] = (m[1] * m[6] - m[0] * m[7]) * s; o[8] = (m[0] * m[4] - m[1] * m[3]) * s;
}
static void mul3(const double* a, const double* b, double* o) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
static void scale_k(const float* K, double s, double* o) {  // scaleK cameraGeometryUtils.h:143-154
    for (int i = 0; i < 9; i++) o[i] = K[i];
    o[0] /= s; o[4] /= s; o[2] /= s; o[5] /= s;
}

// Camera re-origin algebra of getCameraParameters (cameraGeometryUtils.h:270-356), without the
// decomposeProjectionMatrix round trip: K, R, t come straight from the cam files.
static void derive_cameras(tsar_ctx* ctx, const tsar_camera* cams) {
    DevScene& sc = ctx->hscene;
    const double s = ctx->params.cam_scale > 0 ? ctx->params.cam_scale : 1.0;
    double K0[9], R0t[9], t0[3];
    scale_k(cams[0].K, s, K0);
    for (int r = 0; r < 3; r++) {
        t0[r] = cams[0].t[r];
        for (int c = 0; c < 3; c++) R0t[3 * r + c] = cams[0].R[3 * c + r];
    }
    for (int v = 0; v < ctx->n_views; v++) {
        double Kv[9], Rv[9], Rrel[9], trel[3];
        scale_k(cams[v].K, s, Kv);
        for (int i = 0; i < 9; i++) Rv[i] = cams[v].R[i];
        mul3(Rv, R0t, Rrel);                                     // [R|t] [R0|t0]^-1
        for (int r = 0; r < 3; r++) trel[r] = cams[v].t[r] - (Rrel[3 * r] * t0[0] + Rrel[3 * r + 1] * t0[1] + Rrel[3 * r + 2] * t0[2]);
        if (v == 0) {                                            // the reference camera is exactly K[I|0]
            for (int i = 0; i < 9; i++) Rrel[i] = (i % 4 == 0) ? 1.0 : 0.0;
            trel[0] = trel[1] = trel[2] = 0.0;
        }
        DevView& dv = sc.view[v];
        for (int i = 0; i < 9; i++) { dv.K[i] = (float)Kv[i]; dv.R[i] = (float)Rrel[i]; }
        for (int r = 0; r < 3; r++) dv.t[r] = (float)trel[r];
        dv.t_abs_lo = std::min(std::min(fabsf(dv.t[0]), fabsf(dv.t[1])), fabsf(dv.t[2]));
        dv.t_abs_hi = std::max(std::max(fabsf(dv.t[0]), fabsf(dv.t[1])), fabsf(dv.t[2]));
        {                                                        // fast-mode split of the plane homography: A = K R K0^-1, b = K t
            double K0inv[9], KR[9], A[9];
            inv3(K0, K0inv);
            mul3(Kv, Rrel, KR);
            mul3(KR, K0inv, A);
            for (int i = 0; i < 9; i++) dv.A[i] = (float)A[i];
            for (int r = 0; r < 3; r++) dv.b[r] = (float)(Kv[3 * r] * trel[0] + Kv[3 * r + 1] * trel[1] + Kv[3 * r + 2] * trel[2]);
        }
        if (v == 0) {
            DevRef& rf = sc.ref;
            double Kinv[9], M[9], Minv[9];
            inv3(Kv, Kinv);
            mul3(K0, Rrel, M);                                   // P = K_ref [R|t] (cameraGeometryUtils.h:302)
            inv3(M, Minv);
            for (int i = 0; i < 9; i++) {
                rf.K[i] = (float)Kv[i]; rf.Kinv[i] = (float)Kinv[i]; rf.Minv[i] = (float)Minv[i];
                rf.Rorig[i] = cams[0].R[i];
            }
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) rf.RorigInv[3 * r + c] = cams[0].R[3 * c + r];
            for (int r = 0; r < 3; r++) {
                rf.P34[r] = (float)(K0[3 * r] * trel[0] + K0[3 * r + 1] * trel[1] + K0[3 * r + 2] * trel[2]);
                rf.C[r] = (float)(-(Rrel[r] * trel[0] + Rrel[3 + r] * trel[1] + Rrel[6 + r] * trel[2]));
            }
            rf.fx = (float)K0[0];
            rf.f = (float)K0[0];
            rf.alpha = (float)K0[0] / (float)K0[4];
            rf.baseline = 1.0f;                                  // cameraGeometryUtils.h:309
            rf.depthMin = ctx->params.depth_min;
            rf.depthMax = ctx->params.depth_max;
        }
    }
    // zero / one pattern of the intrinsics (cam files without skew, the only kind MVSNet-format scenes carry)
    auto sparse3 = [](const float* K) { return K[1] == 0.f && K[3] == 0.f && K[6] == 0.f && K[7] == 0.f && K[8] == 1.f; };
    sc.k_sparse = sparse3(sc.ref.Kinv) ? 1 : 0;
    for (int v = 0; v < ctx->n_views; v++) sc.k_sparse &= sparse3(sc.view[v].K) ? 1 : 0;
    // main.cpp:1393-1398
    sc.min_disp = sc.ref.f * sc.ref.baseline / ctx->params.depth_max;
    sc.max_disp = sc.ref.f * sc.ref.baseline / ctx->params.depth_min;
    int steps = 0;
    for (float dz = sc.max_disp / 2.0f; dz >= 0.01f; dz = dz / 10.0f) steps++;   // gipuma.cu:643-644
    sc.refine_steps = steps;
}

// gipuma_init_cu2 takes its window radius as box / 2 (gipuma.cu:693-694) where every other kernel takes (box - 1) / 2
// (:858-859, :1065-1066, :1175-1176): an even --blocksize initialises on a window one tap ring larger than the one it sweeps with.
// Reproduced by default; TSAR_FLAG_FIX_INIT_RADIUS initialises on the sweeps' window.
static bool init_window_differs(const tsar_ctx* ctx) {
    const tsar_params& p = ctx->params;
    return !(p.flags & TSAR_FLAG_FIX_INIT_RADIUS) && (p.box_hsize / 2 != (p.box_hsize - 1) / 2 || p.box_vsize / 2 != (p.box_vsize - 1) / 2);
}
static void fill_scene_params(tsar_ctx* ctx, bool for_init = false) {
    DevScene& sc = ctx->hscene;
    const tsar_params& p = ctx->params;
    const bool init_radius = for_init && !(p.flags & TSAR_FLAG_FIX_INIT_RADIUS);
    sc.hrad = init_radius ? p.box_hsize / 2 : (p.box_hsize - 1) / 2;   // gipuma.cu:693-694 : gipuma.cu:858-859
    sc.vrad = init_radius ? p.box_vsize / 2 : (p.box_vsize - 1) / 2;
    sc.n_best = p.n_best;
    sc.cost_comb = p.cost_comb;
    sc.flags = p.flags;
    sc.seed_lo = (uint32_t)p.seed;
    sc.seed_hi = (uint32_t)(p.seed >> 32);
    // weight table of the general-window tap loop (tsar_dev.h DevScene::tap_row): distance classes of the window's taps, and
    // per (line, tap of the line) the row of its class.  Fast mode walks window rows, strict mode the oracle's columns.
    sc.lut_row_major = (p.flags & TSAR_FLAG_STRICT_DIV) ? 0 : 1;
    std::vector<int> d2;
    for (int i = -sc.hrad; i <= sc.hrad; i += 2)
        for (int j = -sc.vrad; j <= sc.vrad; j += 2) d2.push_back(i * i + j * j);
    std::sort(d2.begin(), d2.end());
    d2.erase(std::unique(d2.begin(), d2.end()), d2.end());
    sc.lut_classes = (int)d2.size() <= TSAR_LUT_MAX_CLASSES ? (int)d2.size() : 0;
    for (int k = 0; k < sc.lut_classes; k++) sc.lut_d2[k] = d2[k];
    const int rl = sc.lut_row_major ? sc.vrad : sc.hrad, rt = sc.lut_row_major ? sc.hrad : sc.vrad;   // radius across / along the lines
    sc.lut_chunk = lut_chunk_taps(rt + 1);
    sc.lut_pad_taps = (rt + sc.lut_chunk) / sc.lut_chunk * sc.lut_chunk;
    for (uint32_t& r : sc.tap_row) r = (uint32_t)sc.lut_classes * 1024u;      // the zero row: slots beyond the end of a line
    for (int l = 0; l <= rl && sc.lut_classes; l++)
        for (int t = 0; t <= rt; t++) {
            const int a = 2 * l - rl, b = 2 * t - rt;
            sc.tap_row[l * sc.lut_pad_taps + t] = 1024u * (uint32_t)(std::lower_bound(d2.begin(), d2.end(), a * a + b * b) - d2.begin());
        }
}
static int upload_scene(tsar_ctx* ctx) {
    if (!ctx->dscene) TRY(dev_alloc(ctx, &ctx->dscene, 1));
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->dscene, &ctx->hscene, sizeof(DevScene), hipMemcpyHostToDevice, ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // hscene may be edited again by the caller's next call
    return TSAR_OK;
}

// ---- lifecycle -----------------------------------------------------------------------------------
extern "C" const char* tsar_version(void) {
#ifdef TSAR_EXPERIMENTS
    return "tsar-mvs_amd 0.2.0 (gfx950) +experiments";
#else
    return "tsar-mvs_amd 0.2.0 (gfx950)";
#endif
}

// Every environment knob of the library, read once per context.  All are diagnostics / A-B switches: the defaults
// are the measured best and nothing in the product path depends on one being set.
static void read_knobs(tsar_ctx* ctx) {
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e && e[0] ? atoi(e) : dflt; };
    auto on = [](const char* name, bool dflt) { const char* e = getenv(name); return e && e[0] ? e[0] != '0' : dflt; };
    ctx->variant = num("TSAR_VARIANT", ctx->variant);
    ctx->buffer_gather = on("TSAR_BUFFER_GATHER", true);
    ctx->mix_gather = on("TSAR_MIX_GATHER", true);
    ctx->strip_w = num("TSAR_STRIP", -1);
    ctx->buffer_from = num("TSAR_BUFFER_FROM", 1);
    ctx->force_block = num("TSAR_BLOCK", 0);
    ctx->memo_mode = num("TSAR_MEMO", 1);
    ctx->compact_from = num("TSAR_COMPACT_FROM", 6);
    ctx->lut_mode = num("TSAR_LUT", 1);
    ctx->ransac_wgs = num("TSAR_RANSAC_WGS", 8);
    ctx->ransac_chain = num("TSAR_RANSAC_CHAIN", 8);
    ctx->ransac_lookahead = num("TSAR_RANSAC_LOOKAHEAD", 0);
    ctx->ransac_poll_limit = num("TSAR_RANSAC_POLL_LIMIT", 1 << 15);
    ctx->ransac_cooperative = on("TSAR_RANSAC_COOPERATIVE", true);
    ctx->ransac_force_fallback = on("TSAR_RANSAC_FORCE_FALLBACK", false);
    ctx->trace_host = getenv("TSAR_TRACE_HOST") != nullptr;
#ifdef TSAR_EXPERIMENTS
    ctx->lds_pad = (size_t)num("TSAR_LDS_PAD", 0);
#endif
}

extern "C" int tsar_create(int device, tsar_ctx** out) {
    if (!out) return TSAR_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TSAR_ERR_HIP;
    if (device < 0 || device >= n) return TSAR_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return TSAR_ERR_HIP;
    tsar_ctx* ctx = new (std::nothrow) tsar_ctx();
    if (!ctx) return TSAR_ERR_NOMEM;
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return TSAR_ERR_HIP; }
    tsar_default_params(&ctx->params);
    ctx->variant = probe_d16_hi_zeroes(ctx) ? 250 : 114;
    read_knobs(ctx);
    *out = ctx;
    return TSAR_OK;
}

static void free_views(tsar_ctx* ctx) {
    for (auto& p : ctx->img) dev_free(p);
    for (auto& p : ctx->quad) dev_free(p);
    for (auto& p : ctx->dquad) dev_free(p);
    ctx->img.clear();
    ctx->quad.clear();
    ctx->dquad.clear();
}
static void free_planes(tsar_ctx* ctx) {
    for (int b = 0; b < 2; b++) { dev_free(ctx->buf[b].c); dev_free(ctx->buf[b].n4); }
    dev_free(ctx->ratio); dev_free(ctx->depth); dev_free(ctx->scale); dev_free(ctx->lrdiff); dev_free(ctx->confid);
    dev_free(ctx->fakedepth); dev_free(ctx->beview); dev_free(ctx->canny); dev_free(ctx->out4);
    dev_free(ctx->memo_cand); dev_free(ctx->memo_seq); dev_free(ctx->changed_seq);
}

extern "C" int tsar_destroy(tsar_ctx* ctx) {
    if (!ctx) return TSAR_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    drain_timers(ctx);
    free_views(ctx);
    free_planes(ctx);
    dev_free(ctx->dscene); dev_free(ctx->region_text); dev_free(ctx->region_size); dev_free(ctx->region_n4);
    if (ctx->scratch.base) hipFree(ctx->scratch.base);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return TSAR_OK;
}
extern "C" const char* tsar_last_error(const tsar_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" int tsar_get_stream(tsar_ctx* ctx, void** stream_out) {
    if (!ctx || !stream_out) return TSAR_ERR_INVALID;
    *stream_out = (void*)ctx->stream;
    return TSAR_OK;
}
extern "C" int tsar_synchronize(tsar_ctx* ctx) {
    CHECK_CTX(ctx);
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}

// ---- inputs --------------------------------------------------------------------------------------
extern "C" void tsar_default_params(tsar_params* p) {   // algorithmparameters.h:21-52
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->box_hsize = 19;
    p->box_vsize = 19;
    p->n_best = 2;
    p->cost_comb = TSAR_COMB_BEST_N;
    p->depth_min = 2.0f;     // camera.h:37-44
    p->depth_max = 20.0f;
    p->cam_scale = 1.0f;
    p->flags = 0;
    p->seed = 0;
}

extern "C" int tsar_set_params(tsar_ctx* ctx, const tsar_params* p) {
    CHECK_CTX(ctx);
    if (!p) return fail(ctx, TSAR_ERR_INVALID, "params is NULL");
    if (p->box_hsize < 1 || p->box_vsize < 1 || p->box_hsize > 63 || p->box_vsize > 63) return fail(ctx, TSAR_ERR_INVALID, "box size must be in 1..63");
    if (p->n_best < 1 || p->n_best > TSAR_MAX_SELECTED) return fail(ctx, TSAR_ERR_INVALID, "n_best must be in 1..32 (the reference's costVector holds 32 views, gipuma.cu:467)");
    if (p->cost_comb < TSAR_COMB_ALL || p->cost_comb > TSAR_COMB_GOOD) return fail(ctx, TSAR_ERR_INVALID, "cost_comb must be one of TSAR_COMB_ALL / BEST_N / ANGLE / GOOD");
    if (!(p->depth_min > 0.f) || !(p->depth_max > p->depth_min)) return fail(ctx, TSAR_ERR_INVALID, "need 0 < depth_min < depth_max");
    if (!(p->cam_scale > 0.f)) return fail(ctx, TSAR_ERR_INVALID, "cam_scale must be > 0");
    ctx->params = *p;
    ctx->have_params = true;
    fill_scene_params(ctx);
    if (ctx->have_views) {
        // depth range / scale feed the derived camera block: the views must be set again
        ctx->have_views = false;
        ctx->have_state = false;
    }
    return TSAR_OK;
}

// Strict mode's quotients are bit-exact because persp_divide_exact / div_pair_rcp_exact (tsar_device_math.h) return the correctly
// rounded quotient — a property of THIS GPU's v_rcp_f32, enumerated over all 2^46 mantissa pairs on the device it was measured on
// (profiles/r03/div_exact_all_mantissa_pairs.json).  Like the D16 probe, it is re-checked where it is relied on: once per context,
// before the first strict-mode views are accepted, 2^22 random operand triples in each of the tap loop's two operand
// distributions through the SHIPPED unguarded code path against the compiler's IEEE division (~1 ms).  A device whose reciprocal
// differs fails loudly here instead of drifting from the oracle silently.
static int probe_exact_divide(tsar_ctx* ctx) {
    if (ctx->exact_div_probe != 0) return ctx->exact_div_probe;
    uint64_t bad = 0, total = 0;
    for (int mode = 0; mode < 2; mode++) {
        if (tsar_selftest_divide_random(ctx, 22, 0x5EEDD1F1DE5ull + (uint64_t)mode, mode, /*guarded=*/0, &bad, nullptr) != TSAR_OK) return 0;   // not probed: the caller reports ctx->err
        total += bad;
    }
    ctx->exact_div_probe = total == 0 ? 1 : -1;
    return ctx->exact_div_probe;
}
// The same for the square root of the cost's tail (sqrt_rsq_exact, tsar_device_math.h), which BOTH arithmetic modes run on 8-bit
// imagery: all 2^24 mantissa / exponent-parity cases against sqrtf on the device, once per context (~0.2 ms).
// Second pass (round 5): the enumeration covers two binades; that the result carries to other exponents rests on v_rsq_f32 scaling
// exactly with the exponent, so 2^24 random mantissas spread over the 67 binades the tail's operands can reach (var_ref * var_src in
// [1e-10, 4.3e9]) are checked per context as well.  Run only for contexts whose views are 8-bit imagery: the float-imagery loop and
// the refinement operators keep sqrtf.
static int probe_exact_sqrt(tsar_ctx* ctx) {
    if (ctx->exact_sqrt_probe != 0) return ctx->exact_sqrt_probe;
    uint64_t bad = 0, bad_range = 0;
    if (tsar_selftest_sqrt(ctx, 0, 0, &bad) != TSAR_OK) return 0;
    if (tsar_selftest_sqrt(ctx, 3, 0x5EED5A17ull, &bad_range) != TSAR_OK) return 0;
    ctx->exact_sqrt_probe = (bad | bad_range) == 0 ? 1 : -1;
    return ctx->exact_sqrt_probe;
}

// tsar_set_views (elem = 4: float32 images) and tsar_set_views_u8 (elem = 1: the 8-bit decode itself, widened on the device)
static int set_views_impl(tsar_ctx* ctx, int n_views, int w, int h, const void* const* gray, int elem, int mem, const tsar_camera* cams) {
    CHECK_CTX(ctx);
    if (!ctx->have_params) return fail(ctx, TSAR_ERR_STATE, "tsar_set_params must be called before tsar_set_views");
    if (ctx->params.flags & TSAR_FLAG_STRICT_DIV) {
        const int pr = probe_exact_divide(ctx);
        if (pr == 0) return TSAR_ERR_HIP;
        if (pr < 0) return fail(ctx, TSAR_ERR_HIP, "TSAR_FLAG_STRICT_DIV: this device's v_rcp_f32 does not give correctly rounded quotients through the short division "
                                                   "sequence (tsar_selftest_divide_random found mismatches against IEEE division); strict mode is refused rather than run inexactly");
    }
    if (n_views < 1 || n_views > TSAR_MAX_VIEWS) return fail(ctx, TSAR_ERR_INVALID, "n_views must be in 1..TSAR_MAX_VIEWS");
    if (w < 8 || h < 8 || (int64_t)w * h > (int64_t)1 << 28) return fail(ctx, TSAR_ERR_INVALID, "image size out of range");
    if (w + 2 >= (1 << 23) || h + 2 >= (1 << 23)) return fail(ctx, TSAR_ERR_INVALID, "image side too long for the 24-bit quad addressing (w + 2, h + 2 < 2^23)");
    if (!gray || !cams) return fail(ctx, TSAR_ERR_INVALID, "gray/cams is NULL");
    for (int v = 0; v < n_views; v++)
        if (!gray[v]) return fail(ctx, TSAR_ERR_INVALID, "gray[v] is NULL");
    ctx->have_views = false;
    ctx->have_state = false;
    ctx->have_out = false;
    // The image and quad-texture buffers of the previous views are kept when the size is the same (a worker matching view after
    // view of a scene): 2 x n_views hipMalloc + hipFree of ~100 MB each cost 55 ms per call at ETH3D size, more than the copies.
    // Buffers beyond n_views stay in the pool; a change of size releases everything.
    const size_t np = (size_t)w * h;
    if (w != ctx->w || h != ctx->h) { free_views(ctx); free_planes(ctx); }
    ctx->w = w; ctx->h = h; ctx->n_views = n_views;
    DevScene& sc = ctx->hscene;
    sc.w = w; sc.h = h; sc.quad_pitch = w + 2;
    if ((int)ctx->img.size() < n_views) ctx->img.resize(n_views, nullptr);
    if ((int)ctx->quad.size() < n_views) ctx->quad.resize(n_views, nullptr);
    if ((int)ctx->dquad.size() < n_views) ctx->dquad.resize(n_views, nullptr);
    struct DevInt {   // freed on every exit path
        int* p = nullptr;
        ~DevInt() { if (p) hipFree(p); }
    } dflag_owner;
    TRY(dev_alloc(ctx, &dflag_owner.p, 1));
    int* const dflag = dflag_owner.p;
    hipMemsetAsync(dflag, 0, sizeof(int), ctx->stream);
    for (int v = 0; v < n_views; v++) {
        if (!ctx->img[v]) TRY(dev_alloc(ctx, &ctx->img[v], np));
        if (!ctx->quad[v]) TRY(dev_alloc(ctx, &ctx->quad[v], (size_t)(w + 2) * (h + 2)));
        if (elem == 4) {
            TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->img[v], gray[v], np * sizeof(float), in_kind(mem), ctx->stream));
        } else if (mem == TSAR_MEM_DEVICE) {
            TRY(launch_expand_u8(ctx, (const uint8_t*)gray[v], ctx->img[v], np));
        } else {
            // host bytes: staged in the view's own quad-texture buffer ((w + 2)(h + 2) dwords, written only by build_quad below, which
            // reads img[v]) — no extra allocation, and the copies of the views follow each other on the stream without a sync
            uint8_t* stage = (uint8_t*)ctx->quad[v];
            TSAR_HIP_TRY(ctx, hipMemcpyAsync(stage, gray[v], np, hipMemcpyHostToDevice, ctx->stream));
            TRY(launch_expand_u8(ctx, stage, ctx->img[v], np));
        }
        TRY(launch_build_quad(ctx, ctx->img[v], ctx->quad[v], w, h, dflag));
    }
    int hflag = 0;
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    sc.use_quad = hflag ? 0 : 1;
    if (sc.use_quad) {   // the 8-bit tap loops are the only users of sqrt_rsq_exact: probed for the contexts that will run them
        const int ps = probe_exact_sqrt(ctx);
        if (ps == 0) return TSAR_ERR_HIP;
        if (ps < 0) return fail(ctx, TSAR_ERR_HIP, "this device's v_rsq_f32 does not give correctly rounded square roots through the one-correction sequence of the cost's tail "
                                                   "(tsar_selftest_sqrt found mismatches against sqrtf): refused rather than run with costs that differ from the oracle's");
    }
    for (int v = 0; v < n_views; v++) { sc.view[v].img = ctx->img[v]; sc.view[v].quad = ctx->quad[v]; sc.view[v].dquad = nullptr; }
    if (!sc.use_quad)
        for (auto& q : ctx->quad) dev_free(q);              // (float imagery: the textures are not used; re-allocated if a later call needs them)
    // Fast mode's converged sweeps (buffer gathers, from the third sweep of a run on; the box-11 loop and the general-window loop) read
    // the source views from a second texture with half-float differences (pm_tap_r5.h MIX): 8 bytes per texel quad, source views only.
    const bool want_dquad = sc.use_quad && !(ctx->params.flags & TSAR_FLAG_STRICT_DIV) && ctx->buffer_gather && ctx->mix_gather && ctx->variant == 250;
    for (int v = 1; v < n_views; v++) {
        if (!want_dquad) { dev_free(ctx->dquad[v]); continue; }
        if (!ctx->dquad[v]) TRY(dev_alloc(ctx, &ctx->dquad[v], (size_t)(w + 2) * (h + 2)));
        TRY(launch_build_dquad(ctx, ctx->quad[v], ctx->dquad[v], w, h));
        sc.view[v].dquad = ctx->dquad[v];
    }
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    derive_cameras(ctx, cams);
    // Can the matching kernels run this window?  8-bit imagery shares one weight table per workgroup (pm_core_lut.h): any box whose
    // taps have <= TSAR_LUT_MAX_CLASSES distinct distances (every square box; rectangular ones unless their radii have mixed
    // parity, e.g. 63 x 61 -> 202).  Everything else keeps the hoisted bilateral weights per thread in LDS, (hrad+1)(vrad+1) taps
    // x 256 threads x 4 B beside the reference window, which bounds the box at 23.  Checked for the sweeps' window and, for even
    // boxes, gipuma_init_cu2's own (init_window_differs); a context with the reference view alone never matches (refinement
    // operators only) and takes any box.
    auto window_problem = [&](bool for_init) -> const char* {
        fill_scene_params(ctx, for_init);
        if (lut_path_applies(ctx)) return nullptr;
        const size_t lds = (size_t)(sc.hrad + 1) * (sc.vrad + 1) * 1024 + (size_t)(32 + 2 * sc.hrad) * (16 + 2 * sc.vrad) * 4 + 16;
        if (lds <= 160 * 1024) return nullptr;
        if (!sc.use_quad) return "box too large for images that are not 8-bit: the per-thread weight table does not fit the 160 KiB of LDS per CU (largest square box: 23)";
        if (sc.lut_classes == 0) return "this rectangular box has more than 144 distinct tap distances (radii of mixed parity): the shared weight table cannot hold them and the per-thread table does not fit LDS; use radii of equal parity, or a box of at most 23";
        return "box too large for fast mode on this device: the D16 LDS-load probe failed, so the general-window loop is not available; use TSAR_FLAG_STRICT_DIV, or a box of at most 23";
    };
    const char* problem = n_views > 1 ? window_problem(false) : nullptr;
    if (!problem && n_views > 1 && init_window_differs(ctx)) problem = window_problem(true);
    fill_scene_params(ctx);
    if (problem) return fail(ctx, TSAR_ERR_INVALID, problem);
    sc.n_sel = std::min(n_views - 1, TSAR_MAX_SELECTED);   // default subset: the first 32 source views at most (tsar_set_view_subset picks others)
    for (int i = 0; i < sc.n_sel; i++) sc.sel[i] = i + 1;
    // state planes (LineState::resize linestate.h:71-110)
    for (int b = 0; b < 2; b++) {
        if (!ctx->buf[b].c) TRY(dev_alloc(ctx, &ctx->buf[b].c, np));
        if (!ctx->buf[b].n4) TRY(dev_alloc(ctx, &ctx->buf[b].n4, np));
    }
    float** fplanes[] = {&ctx->ratio, &ctx->depth, &ctx->scale, &ctx->lrdiff, &ctx->confid, &ctx->fakedepth};
    for (float** fp : fplanes) {
        if (!*fp) TRY(dev_alloc(ctx, fp, np));
        TSAR_HIP_TRY(ctx, hipMemsetAsync(*fp, 0, np * sizeof(float), ctx->stream));
    }
    if (!ctx->beview) TRY(dev_alloc(ctx, &ctx->beview, np));
    if (!ctx->canny) TRY(dev_alloc(ctx, &ctx->canny, np));
    if (!ctx->out4) TRY(dev_alloc(ctx, &ctx->out4, np));
    TSAR_HIP_TRY(ctx, hipMemsetAsync(ctx->beview, 0, np * sizeof(int32_t), ctx->stream));
    TSAR_HIP_TRY(ctx, hipMemsetAsync(ctx->canny, 0, np * sizeof(int32_t), ctx->stream));
    TRY(upload_scene(ctx));
    ctx->have_views = true;
    return TSAR_OK;
}

extern "C" int tsar_set_views(tsar_ctx* ctx, int n_views, int w, int h, const float* const* gray, int mem, const tsar_camera* cams) {
    return set_views_impl(ctx, n_views, w, h, (const void* const*)gray, 4, mem, cams);
}
extern "C" int tsar_set_views_u8(tsar_ctx* ctx, int n_views, int w, int h, const uint8_t* const* gray, int mem, const tsar_camera* cams) {
    return set_views_impl(ctx, n_views, w, h, (const void* const*)gray, 1, mem, cams);
}

extern "C" int tsar_set_view_subset(tsar_ctx* ctx, int n, const int32_t* view_idx) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    if (n < 1 || n > TSAR_MAX_SELECTED || !view_idx) return fail(ctx, TSAR_ERR_INVALID, "subset size must be in 1..32 (the reference's viewSelectionSubset / costVector hold 32, gipuma.cu:467)");
    for (int i = 0; i < n; i++)
        if (view_idx[i] < 1 || view_idx[i] >= ctx->n_views) return fail(ctx, TSAR_ERR_INVALID, "view index must be in 1..n_views-1");
    ctx->hscene.n_sel = n;
    for (int i = 0; i < n; i++) ctx->hscene.sel[i] = view_idx[i];
    // stored costs were scored under the previous subset: the sweep may no longer skip a neighbour that carries the
    // pixel's own plane (pm_sweep.hip same_bits shortcut); the reference re-scores it and may accept
    if (ctx->have_state) ctx->cost_consistent = false;
    return upload_scene(ctx);
}

// ---- PatchMatch ----------------------------------------------------------------------------------
extern "C" int tsar_pm_init(tsar_ctx* ctx) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    NEED_SOURCES(ctx);
    const bool own_window = init_window_differs(ctx);
    int rc = TSAR_OK;
    if (own_window) {                      // an even box: the scene block describes the init window for this one launch
        fill_scene_params(ctx, true);
        rc = upload_scene(ctx);
    }
    if (rc == TSAR_OK) rc = launch_pm_init(ctx);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) rc = fail(ctx, TSAR_ERR_HIP, "hipStreamSynchronize failed");
    if (own_window) {
        // back to the sweep window on EVERY path (also when the first upload failed: the host block was already rewritten); if the
        // device block cannot be restored, host and device disagree about the window and the views must be set again
        fill_scene_params(ctx, false);
        const int rc2 = upload_scene(ctx);
        if (rc2 != TSAR_OK) ctx->have_views = false;
        if (rc == TSAR_OK) rc = rc2;
    }
    TRY(rc);
    ctx->have_state = true;
    ctx->have_out = false;
    ctx->sweeps_done = 0;
    // c[p] is the score of n4[p] on the SWEEP window only if init ran on that window: otherwise a neighbour's identical plane may
    // well score lower than the stored cost, and the sweeps must not skip it
    ctx->cost_consistent = !own_window;
    return TSAR_OK;
}

static int pm_sweeps(tsar_ctx* ctx, int n_sweeps, int first_colour, int do_prop, int do_refine) {
    // cur[k]: which ping-pong buffer holds the current values of colour k.  Both start in buf[0].
    int cur[2] = {0, 0};
    // the propagation memo lives within this call: whatever happened to the state before (init, another subset, loaded planes) is
    // out of its reach
    ctx->memo_valid_from = ctx->launch_seq + 1;
    ctx->call_launch = 0;
    if (ctx->memo_mode && n_sweeps > 2 && !ctx->memo_cand) {
        const size_t np = (size_t)ctx->w * ctx->h;
        TRY(dev_alloc(ctx, &ctx->memo_cand, np * 8));
        TRY(dev_alloc(ctx, &ctx->memo_seq, np));
        TRY(dev_alloc(ctx, &ctx->changed_seq, np));
        TSAR_HIP_TRY(ctx, hipMemsetAsync(ctx->memo_seq, 0, np * sizeof(uint32_t), ctx->stream));
        TSAR_HIP_TRY(ctx, hipMemsetAsync(ctx->changed_seq, 0, np * sizeof(uint32_t), ctx->stream));
    }
    for (int s = 0; s < n_sweeps; s++) {
        ctx->launch_seq++;
        const int colour = (first_colour + s) & 1;
        const PlaneBuf& same_in = ctx->buf[cur[colour]];
        const PlaneBuf& other = ctx->buf[cur[colour ^ 1]];
        const PlaneBuf& same_out = ctx->buf[cur[colour] ^ 1];
        TRY(launch_pm_sweep(ctx, colour, same_in, other, same_out, 1u + (uint32_t)ctx->sweeps_done, do_prop, do_refine));
        cur[colour] ^= 1;
        ctx->sweeps_done++;
        ctx->call_launch++;
    }
    // make buf[0] canonical again
    if (cur[0] == 1 && cur[1] == 1) {
        std::swap(ctx->buf[0], ctx->buf[1]);
    } else if (cur[0] != cur[1]) {
        // odd number of sweeps: one colour lives in buf[1]; merge it back (diagnostic path only)
        const size_t np = (size_t)ctx->w * ctx->h;
        const int moved = cur[0] == 1 ? 0 : 1;
        std::vector<float> c0(np), c1(np);
        std::vector<float4> n0(np), n1(np);
        TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        TSAR_HIP_TRY(ctx, hipMemcpy(c0.data(), ctx->buf[0].c, np * 4, hipMemcpyDeviceToHost));
        TSAR_HIP_TRY(ctx, hipMemcpy(c1.data(), ctx->buf[1].c, np * 4, hipMemcpyDeviceToHost));
        TSAR_HIP_TRY(ctx, hipMemcpy(n0.data(), ctx->buf[0].n4, np * 16, hipMemcpyDeviceToHost));
        TSAR_HIP_TRY(ctx, hipMemcpy(n1.data(), ctx->buf[1].n4, np * 16, hipMemcpyDeviceToHost));
        for (int y = 0; y < ctx->h; y++)
            for (int x = 0; x < ctx->w; x++)
                if (((x + y) & 1) == moved) { c0[(size_t)y * ctx->w + x] = c1[(size_t)y * ctx->w + x]; n0[(size_t)y * ctx->w + x] = n1[(size_t)y * ctx->w + x]; }
        TSAR_HIP_TRY(ctx, hipMemcpy(ctx->buf[0].c, c0.data(), np * 4, hipMemcpyHostToDevice));
        TSAR_HIP_TRY(ctx, hipMemcpy(ctx->buf[0].n4, n0.data(), np * 16, hipMemcpyHostToDevice));
    }
    return TSAR_OK;
}

extern "C" int tsar_pm_iterate(tsar_ctx* ctx, int iters) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    NEED_SOURCES(ctx);
    NEED_STATE(ctx);
    if (iters < 0) return fail(ctx, TSAR_ERR_INVALID, "iters must be >= 0");
    TRY(pm_sweeps(ctx, 2 * iters, 0, 1, 1));   // black then red, gipuma.cu:1744-1754
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_out = false;
    return TSAR_OK;
}

// The iteration loop with the kernels' `final` argument true (gipuma.cu:1096-1138 take `bool final`; nothing in the
// snapshot passes true): pixels with text == -1 are not touched, ratio / beview are not written.
extern "C" int tsar_pm_iterate_final(tsar_ctx* ctx, int iters, const float* text, int mem) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    NEED_SOURCES(ctx);
    NEED_STATE(ctx);
    if (iters < 0) return fail(ctx, TSAR_ERR_INVALID, "iters must be >= 0");
    if (!text) return fail(ctx, TSAR_ERR_INVALID, "text is NULL");
    const size_t np = (size_t)ctx->w * ctx->h;
    int rc = TSAR_OK;
    {
        TmpIn<float> t(ctx, text, np, mem);
        TRY(t.rc);
        ctx->final_text = t.d;
        rc = pm_sweeps(ctx, 2 * iters, 0, 1, 1);
        ctx->final_text = nullptr;
        if (rc == TSAR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "sweep failed");
    }
    ctx->have_out = false;
    return rc;
}

// Diagnostics entry (not in the reference): one half-iteration with propagation and/or refinement.
extern "C" int tsar_pm_sweep(tsar_ctx* ctx, int colour, int do_prop, int do_refine) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    NEED_SOURCES(ctx);
    NEED_STATE(ctx);
    TRY(pm_sweeps(ctx, 1, colour & 1, do_prop, do_refine));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_out = false;
    return TSAR_OK;
}
extern "C" int tsar_set_sweep_counter(tsar_ctx* ctx, int n) {
    if (!ctx) return TSAR_ERR_INVALID;
    ctx->sweeps_done = n;
    return TSAR_OK;
}

extern "C" int tsar_pm_cost_planes(tsar_ctx* ctx, const float* planes, int mem, float* cost_out, int32_t* beview_out, float* ratio_out) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    NEED_SOURCES(ctx);
    if (!planes || !cost_out) return fail(ctx, TSAR_ERR_INVALID, "planes/cost_out is NULL");
    const size_t np = (size_t)ctx->w * ctx->h;
    float4* dpl = nullptr;
    float *dc = nullptr, *drt = nullptr;
    int32_t* dbv = nullptr;
    int rc = TSAR_OK;
    if (mem == TSAR_MEM_DEVICE) {
        rc = launch_pm_cost_planes(ctx, (const float4*)planes, cost_out, beview_out, ratio_out);
        if (rc == TSAR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "sync failed");
        return rc;
    }
    if ((rc = dev_alloc(ctx, &dpl, np)) == TSAR_OK && (rc = dev_alloc(ctx, &dc, np)) == TSAR_OK && (rc = dev_alloc(ctx, &drt, np)) == TSAR_OK &&
        (rc = dev_alloc(ctx, &dbv, np)) == TSAR_OK) {
        if (hipMemcpyAsync(dpl, planes, np * 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "H2D failed");
        if (rc == TSAR_OK) rc = launch_pm_cost_planes(ctx, dpl, dc, dbv, drt);
        if (rc == TSAR_OK) {
            hipMemcpyAsync(cost_out, dc, np * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (beview_out) hipMemcpyAsync(beview_out, dbv, np * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (ratio_out) hipMemcpyAsync(ratio_out, drt, np * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "kernel or copy failed");
        }
    }
    dev_free(dpl); dev_free(dc); dev_free(drt); dev_free(dbv);
    return rc;
}

extern "C" int tsar_set_plane(tsar_ctx* ctx, const float* planes, const float* cost, int mem) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    if (!planes || !cost) return fail(ctx, TSAR_ERR_INVALID, "planes/cost is NULL");
    const size_t np = (size_t)ctx->w * ctx->h;
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[0].n4, planes, np * 16, in_kind(mem), ctx->stream));
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->buf[0].c, cost, np * 4, in_kind(mem), ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cost_consistent = false;
    ctx->have_state = true;
    ctx->have_out = false;
    return TSAR_OK;
}
extern "C" int tsar_get_plane(tsar_ctx* ctx, float* planes, float* cost, int32_t* beview, float* ratio, int mem) {
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    const size_t np = (size_t)ctx->w * ctx->h;
    if (planes) TSAR_HIP_TRY(ctx, hipMemcpyAsync(planes, ctx->buf[0].n4, np * 16, out_kind(mem), ctx->stream));
    if (cost) TSAR_HIP_TRY(ctx, hipMemcpyAsync(cost, ctx->buf[0].c, np * 4, out_kind(mem), ctx->stream));
    if (beview) TSAR_HIP_TRY(ctx, hipMemcpyAsync(beview, ctx->beview, np * 4, out_kind(mem), ctx->stream));
    if (ratio) TSAR_HIP_TRY(ctx, hipMemcpyAsync(ratio, ctx->ratio, np * 4, out_kind(mem), ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}

// ---- plane <-> depth -----------------------------------------------------------------------------

extern "C" int tsar_load_planes(tsar_ctx* ctx, const float* depth, const float* normal_world, int mem) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    if (!depth || !normal_world) return fail(ctx, TSAR_ERR_INVALID, "depth/normal_world is NULL");
    const size_t np = (size_t)ctx->w * ctx->h;
    ScratchScope scratch(ctx);             // host maps are staged through the context's scratch arena (released after the sync below)
    TmpIn<float> d(ctx, depth, np, mem, &scratch), n(ctx, normal_world, 3 * np, mem, &scratch);
    TRY(d.rc); TRY(n.rc);
    TRY(launch_get_disp(ctx, d.d, n.d));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cost_consistent = false;
    ctx->have_state = true;
    ctx->have_out = false;
    return TSAR_OK;
}
extern "C" int tsar_compute_disp(tsar_ctx* ctx) {
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    TRY(launch_compute_disp(ctx));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_out = true;
    return TSAR_OK;
}
extern "C" int tsar_compute_disp_final(tsar_ctx* ctx, const float* resize_planes, const float* text, int mem) {
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    if (!resize_planes || !text) return fail(ctx, TSAR_ERR_INVALID, "resize_planes/text is NULL");
    const size_t np = (size_t)ctx->w * ctx->h;
    TmpIn<float> r(ctx, resize_planes, 4 * np, mem), t(ctx, text, np, mem);
    TRY(r.rc); TRY(t.rc);
    ctx->cost_consistent = false;
    TRY(launch_compute_disp_final(ctx, (const float4*)r.d, t.d));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_out = true;
    return TSAR_OK;
}
extern "C" int tsar_depth_to_plane(tsar_ctx* ctx) {
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    ctx->cost_consistent = false;
    TRY(launch_depth_to_plane(ctx));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}
extern "C" int tsar_get_result(tsar_ctx* ctx, float* depth, float* normal_world, float* cost, float* confid, int mem) {
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    if (!ctx->have_out) return fail(ctx, TSAR_ERR_STATE, "call tsar_compute_disp / tsar_fill_textureless first");
    const size_t np = (size_t)ctx->w * ctx->h;
    if (depth || normal_world) {
        if (mem == TSAR_MEM_DEVICE) {
            TRY(launch_split_out4(ctx, depth, normal_world));
        } else {
            ScratchScope scratch(ctx);     // the split maps are staged in the context's scratch arena
            float *dd = nullptr, *dn = nullptr;
            int rc = TSAR_OK;
            if (depth && !(dd = (float*)scratch.alloc(np * 4))) rc = fail(ctx, TSAR_ERR_NOMEM, "hipMalloc failed");
            if (rc == TSAR_OK && normal_world && !(dn = (float*)scratch.alloc(np * 12))) rc = fail(ctx, TSAR_ERR_NOMEM, "hipMalloc failed");
            if (rc == TSAR_OK) rc = launch_split_out4(ctx, dd, dn);
            if (rc == TSAR_OK && depth && hipMemcpyAsync(depth, dd, np * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "D2H failed");
            if (rc == TSAR_OK && normal_world && hipMemcpyAsync(normal_world, dn, np * 12, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "D2H failed");
            hipStreamSynchronize(ctx->stream);
            scratch.release();
            TRY(rc);
        }
    }
    if (cost) TSAR_HIP_TRY(ctx, hipMemcpyAsync(cost, ctx->buf[0].c, np * 4, out_kind(mem), ctx->stream));
    if (confid) TSAR_HIP_TRY(ctx, hipMemcpyAsync(confid, ctx->confid, np * 4, out_kind(mem), ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}

// ---- TSAR refinement -----------------------------------------------------------------------------
extern "C" int tsar_set_reliable_mask(tsar_ctx* ctx, const float* scale, int mem) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    if (!scale) return fail(ctx, TSAR_ERR_INVALID, "scale is NULL");
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->scale, scale, (size_t)ctx->w * ctx->h * 4, in_kind(mem), ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}
extern "C" int tsar_get_reliable_mask(tsar_ctx* ctx, float* scale, int mem) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    if (!scale) return fail(ctx, TSAR_ERR_INVALID, "scale is NULL");
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(scale, ctx->scale, (size_t)ctx->w * ctx->h * 4, out_kind(mem), ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}
extern "C" int tsar_getview(tsar_ctx* ctx) {
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    TRY(launch_getview(ctx));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}
extern "C" int tsar_lrdiff(tsar_ctx* ctx) {
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    NEED_SOURCES(ctx);
    TRY(launch_lrdiff(ctx));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}
extern "C" int tsar_set_regions(tsar_ctx* ctx, const int32_t* labels, int n_regions, const float* region_text, const float* region_size, int mem) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    if (!labels || !region_text || n_regions < 1) return fail(ctx, TSAR_ERR_INVALID, "labels/region_text is NULL or n_regions < 1");
    const size_t np = (size_t)ctx->w * ctx->h;
    {
        // every label indexes the region tables in update_scale / fake_depth / the RANSAC kernels: check the range
        // before anything is installed (one pass; a bad label must give TSAR_ERR_INVALID, not a device fault)
        int32_t lo = 0, hi = 0;
        if (mem == TSAR_MEM_DEVICE) {
            TRY(launch_label_range(ctx, labels, np, &lo, &hi));
        } else {
            lo = hi = labels[0];
            for (size_t i = 1; i < np; i++) { lo = std::min(lo, labels[i]); hi = std::max(hi, labels[i]); }
        }
        if (lo < 0 || hi >= n_regions) return fail(ctx, TSAR_ERR_INVALID, "a label is outside [0, n_regions)");
    }
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->canny, labels, np * 4, in_kind(mem), ctx->stream));
    TRY(dev_alloc(ctx, &ctx->region_text, (size_t)n_regions));
    TRY(dev_alloc(ctx, &ctx->region_size, (size_t)n_regions));
    TRY(dev_alloc(ctx, &ctx->region_n4, (size_t)n_regions));
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->region_text, region_text, (size_t)n_regions * 4, in_kind(mem), ctx->stream));
    if (region_size) TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->region_size, region_size, (size_t)n_regions * 4, in_kind(mem), ctx->stream));
    else TSAR_HIP_TRY(ctx, hipMemsetAsync(ctx->region_size, 0, (size_t)n_regions * 4, ctx->stream));
    TSAR_HIP_TRY(ctx, hipMemsetAsync(ctx->region_n4, 0, (size_t)n_regions * 16, ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_regions = n_regions;
    return TSAR_OK;
}
extern "C" int tsar_set_region_planes(tsar_ctx* ctx, const float* region_planes) {
    CHECK_CTX(ctx);
    if (ctx->n_regions < 1) return fail(ctx, TSAR_ERR_STATE, "tsar_set_regions has not been called");
    if (!region_planes) return fail(ctx, TSAR_ERR_INVALID, "region_planes is NULL");
    TSAR_HIP_TRY(ctx, hipMemcpyAsync(ctx->region_n4, region_planes, (size_t)ctx->n_regions * 16, hipMemcpyHostToDevice, ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}
extern "C" int tsar_fake_depth(tsar_ctx* ctx, float* fakedepth_out, int mem) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    if (ctx->n_regions < 1) return fail(ctx, TSAR_ERR_STATE, "tsar_set_regions has not been called");
    TRY(launch_fake_depth(ctx));
    if (fakedepth_out) TSAR_HIP_TRY(ctx, hipMemcpyAsync(fakedepth_out, ctx->fakedepth, (size_t)ctx->w * ctx->h * 4, out_kind(mem), ctx->stream));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSAR_OK;
}
extern "C" int tsar_fill_textureless(tsar_ctx* ctx) {   // gipuma_fill gipuma.cu:1819-1850
    CHECK_CTX(ctx);
    NEED_STATE(ctx);
    if (ctx->n_regions < 1) return fail(ctx, TSAR_ERR_STATE, "tsar_set_regions has not been called");
    ctx->cost_consistent = false;
    TRY(launch_update_scale(ctx));
    TRY(launch_compute_disp(ctx));
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_out = true;
    return TSAR_OK;
}

// ---- pinned host buffers ---------------------------------------------------------------------------
// Host buffers handed to the library move over PCIe; from pageable memory the runtime stages them through its own
// bounce buffers (tsar_get_result of a 6048 x 4032 view: 51 ms), from page-locked memory the DMA engine reads / writes
// them directly.  The host side (tsar_gipuma, bench.py's host_boundary leg) allocates its image and result buffers here.
extern "C" void* tsar_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void tsar_host_free(void* p) {
    if (p) hipHostFree(p);
}

// ---- device buffers for a multi-GPU host (tsar_gipuma --all --fuse) -----------------------------------------
// One host process drives every GPU of the node (one worker thread + context per device).  Results that are to be fused
// stay on the device that produced them and travel to the fusing device directly over xGMI (peer copy) — the role the
// file system plays in the reference's per-view shell loop (scripts/courtyard.sh:29-48 -> Fusion.exe).  Inside one process
// a peer copy IS the point-to-point transfer a gather is made of; RCCL carries the same gather between the one-rank-per-GPU
// processes of bench.py (driver.gather_results).
extern "C" void* tsar_device_alloc(int device, size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
extern "C" void tsar_device_free(int device, void* p) {
    if (p && hipSetDevice(device) == hipSuccess) hipFree(p);
}
extern "C" int tsar_device_write(int device, void* dst, const void* host_src, size_t bytes) {
    if (!dst || !host_src) return TSAR_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess || hipMemcpy(dst, host_src, bytes, hipMemcpyHostToDevice) != hipSuccess) return TSAR_ERR_HIP;
    return TSAR_OK;
}
extern "C" int tsar_peer_copy(int dst_device, void* dst, int src_device, const void* src, size_t bytes) {
    if (!dst || !src) return TSAR_ERR_INVALID;
    if (hipSetDevice(dst_device) != hipSuccess) return TSAR_ERR_HIP;
    if (dst_device != src_device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst_device, src_device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return TSAR_ERR_HIP;
            (void)hipGetLastError();
        }
    }
    if (hipMemcpyPeer(dst, dst_device, src, src_device, bytes) != hipSuccess) return TSAR_ERR_HIP;   // synchronous: complete on return
    return TSAR_OK;
}

// ---- measurement ---------------------------------------------------------------------------------
extern "C" int tsar_enable_kernel_timing(tsar_ctx* ctx, int enable) {
    if (!ctx) return TSAR_ERR_INVALID;
    ctx->timing = enable != 0;
    return TSAR_OK;
}
extern "C" int tsar_reset_kernel_timing(tsar_ctx* ctx) {
    CHECK_CTX(ctx);
    hipStreamSynchronize(ctx->stream);
    drain_timers(ctx);
    ctx->timers.clear();
    return TSAR_OK;
}
extern "C" int tsar_get_kernel_timing(tsar_ctx* ctx, tsar_kernel_timing* out, int cap, int* n_out) {
    CHECK_CTX(ctx);
    if (!n_out) return fail(ctx, TSAR_ERR_INVALID, "n_out is NULL");
    TSAR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    *n_out = (int)ctx->timers.size();
    for (int i = 0; i < *n_out && i < cap && out; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, ctx->timers[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = ctx->timers[i].launches;
        out[i].total_ms = ctx->timers[i].total_ms;
    }
    return TSAR_OK;
}

// ---- self-tests (selftest_kernels.hip) ---------------------------------------------------------------------------------------
extern "C" int tsar_selftest_divide(tsar_ctx* ctx, const float* X, const float* Y, const float* Z, size_t n, float* u_out, float* v_out, int ieee) {
    CHECK_CTX(ctx);
    if (!X || !Y || !Z || !u_out || !v_out || n == 0 || n > ((size_t)1 << 28)) return fail(ctx, TSAR_ERR_INVALID, "NULL argument or n out of range (1..2^28)");
    ScratchScope scratch(ctx);
    float* d[5];
    for (auto& p : d)
        if (!(p = (float*)scratch.alloc(n * sizeof(float)))) { scratch.release(); return fail(ctx, TSAR_ERR_NOMEM, "device allocation failed"); }
    const float* src[3] = {X, Y, Z};
    int rc = TSAR_OK;
    for (int k = 0; k < 3 && rc == TSAR_OK; k++)
        if (hipMemcpyAsync(d[k], src[k], n * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemcpyAsync failed");
    if (rc == TSAR_OK) rc = launch_selftest_divide(ctx, d[0], d[1], d[2], n, d[3], d[4], ieee);
    if (rc == TSAR_OK && (hipMemcpyAsync(u_out, d[3], n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                          hipMemcpyAsync(v_out, d[4], n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess))
        rc = fail(ctx, TSAR_ERR_HIP, "hipMemcpyAsync failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) rc = fail(ctx, TSAR_ERR_HIP, "hipStreamSynchronize failed");
    scratch.release();
    return rc;
}
extern "C" int tsar_selftest_divide_random(tsar_ctx* ctx, int log2_triples, uint64_t seed, int mode, int guarded, uint64_t* mismatches_out,
                                           uint64_t* outside_guard_out) {
    CHECK_CTX(ctx);
    if (log2_triples < 6 || log2_triples > 36 || mode < 0 || mode > 2 || !mismatches_out) return fail(ctx, TSAR_ERR_INVALID, "log2_triples in 6..36, mode in 0..2");
    if (!guarded && mode == 2) return fail(ctx, TSAR_ERR_INVALID, "the unguarded form is only defined inside the guard (modes 0, 1)");
    ScratchScope scratch(ctx);
    unsigned long long* dc = (unsigned long long*)scratch.alloc(2 * sizeof(unsigned long long));
    if (!dc) { scratch.release(); return fail(ctx, TSAR_ERR_NOMEM, "device allocation failed"); }
    unsigned long long hc[2] = {0, 0};
    int rc = TSAR_OK;
    if (hipMemsetAsync(dc, 0, sizeof hc, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemsetAsync failed");
    // launches of 2^30 triples at most (~0.1 s each)
    for (int done = 0; rc == TSAR_OK && done < (1 << (log2_triples > 30 ? log2_triples - 30 : 0)); done++)
        rc = launch_selftest_divide_random(ctx, log2_triples > 30 ? 30 : log2_triples, seed + 0x9E3779B97F4A7C15ull * (uint64_t)done, mode, guarded, dc);
    if (rc == TSAR_OK && hipMemcpyAsync(hc, dc, sizeof hc, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemcpyAsync failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) rc = fail(ctx, TSAR_ERR_HIP, "hipStreamSynchronize failed");
    scratch.release();
    *mismatches_out = hc[0];
    if (outside_guard_out) *outside_guard_out = hc[1];
    return rc;
}
extern "C" int tsar_selftest_sqrt(tsar_ctx* ctx, int mode, uint64_t seed, uint64_t* mismatches_out) {
    CHECK_CTX(ctx);
    if (mode < 0 || mode > 3 || !mismatches_out) return fail(ctx, TSAR_ERR_INVALID, "mode in 0..3");
    ScratchScope scratch(ctx);
    unsigned long long* dc = (unsigned long long*)scratch.alloc(sizeof(unsigned long long));
    if (!dc) { scratch.release(); return fail(ctx, TSAR_ERR_NOMEM, "device allocation failed"); }
    unsigned long long hc = 0;
    int rc = TSAR_OK;
    if (hipMemsetAsync(dc, 0, sizeof hc, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemsetAsync failed");
    if (rc == TSAR_OK) rc = launch_selftest_sqrt(ctx, mode, seed, dc);
    if (rc == TSAR_OK && hipMemcpyAsync(&hc, dc, sizeof hc, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemcpyAsync failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) rc = fail(ctx, TSAR_ERR_HIP, "hipStreamSynchronize failed");
    scratch.release();
    *mismatches_out = hc;
    return rc;
}
int launch_sweep_repeat(tsar_ctx* ctx, int colour, unsigned long long* memo, unsigned long long* dout);
extern "C" int tsar_selftest_sweep_repeat(tsar_ctx* ctx, int colour, void* memo_dev, uint64_t* out8) {
    CHECK_CTX(ctx);
    if (!memo_dev || !out8) return fail(ctx, TSAR_ERR_INVALID, "memo_dev / out8 is NULL");
    NEED_VIEWS(ctx);
    NEED_STATE(ctx);
    ScratchScope scratch(ctx);
    unsigned long long* dc = (unsigned long long*)scratch.alloc(8 * sizeof(unsigned long long));
    if (!dc) { scratch.release(); return fail(ctx, TSAR_ERR_NOMEM, "device allocation failed"); }
    int rc = TSAR_OK;
    if (hipMemsetAsync(dc, 0, 64, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemsetAsync failed");
    if (rc == TSAR_OK) rc = launch_sweep_repeat(ctx, colour & 1, (unsigned long long*)memo_dev, dc);
    if (rc == TSAR_OK && hipMemcpyAsync(out8, dc, 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemcpyAsync failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) rc = fail(ctx, TSAR_ERR_HIP, "hipStreamSynchronize failed");
    scratch.release();
    return rc;
}
extern "C" int tsar_selftest_sweep_census(tsar_ctx* ctx, int colour, uint64_t* out8) {
    CHECK_CTX(ctx);
    NEED_VIEWS(ctx);
    NEED_STATE(ctx);
    if (!out8) return fail(ctx, TSAR_ERR_INVALID, "out8 is NULL");
    ScratchScope scratch(ctx);
    unsigned long long* dc = (unsigned long long*)scratch.alloc(8 * sizeof(unsigned long long));
    if (!dc) { scratch.release(); return fail(ctx, TSAR_ERR_NOMEM, "device allocation failed"); }
    int rc = TSAR_OK;
    if (hipMemsetAsync(dc, 0, 64, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemsetAsync failed");
    if (rc == TSAR_OK) rc = launch_sweep_census(ctx, colour & 1, dc);
    if (rc == TSAR_OK && hipMemcpyAsync(out8, dc, 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, TSAR_ERR_HIP, "hipMemcpyAsync failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) rc = fail(ctx, TSAR_ERR_HIP, "hipStreamSynchronize failed");
    scratch.release();
    return rc;
}
