// pm_init.hip — random plane initialisation (gipuma_init_cu2, reference gipuma.cu:678-729) and the
// diagnostic "score these planes" kernel (pmCostMultiview_cu over a caller-supplied plane map).
// Both visit every pixel: region 32 x 8 per 256-thread workgroup.
#include "pm_core.h"

#define FULL_RH 8

template <int NB, int HR, bool STRICT, bool QUAD, bool INIT, int V = 0>
__global__ __launch_bounds__(PM_BLOCK) void pm_full_kernel(const DevScene* __restrict__ sc, const float4* __restrict__ planes_in,
                                                           float* __restrict__ c_out, float4* __restrict__ n_out,
                                                           int32_t* __restrict__ beview_out, float* __restrict__ ratio_out, int tiles_x,
                                                           int n_tiles, int strip_w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename TileOf<QUAD>::type TileT;
    const int hr = HR > 0 ? HR : sc->hrad, vr = HR > 0 ? HR : sc->vrad;
    const int tw = PM_RW + 2 * hr, th = FULL_RH + 2 * vr;
    TileT* tile = (TileT*)lds_raw;
    float* wts = (float*)(lds_raw + tile_bytes<QUAD>(tw, th)) + threadIdx.x;
    const int t = xcd_tile(blockIdx.x, n_tiles);
    int tix, tiy;
    strip_tile(t, tiles_x, n_tiles / tiles_x, strip_w, tix, tiy);
    const int ty0 = tiy * FULL_RH, tx0 = tix * PM_RW;
    stage_ref_tile<FULL_RH, TileT>(sc, tile, tx0, ty0, hr, vr);
    __syncthreads();
    const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
    const int x = tx0 + lx, y = ty0 + ly;
    const int w = sc->w, h = sc->h;
    if (x >= w || y >= h) return;
    const int p = y * w + x;
    const int own = (ly + vr) * tw + lx + hr;
    const DevRef& rf = sc->ref;

    float4 n4;
    if (INIT) {
        float vv[3];
        view_vector(rf, x, y, vv);
        Rand4 rn = philox_uniform4((uint32_t)p, 0u, 0u, sc->seed_lo, sc->seed_hi);
        const float disp = between(rn.u[0], sc->min_disp, sc->max_disp);
        // rndUnitVectorSphereMarsaglia_cu gipuma.cu:118-132
        float a = between(rn.u[1], -1.0f, 1.0f), b = between(rn.u[2], -1.0f, 1.0f);
        float sum = fma_(a, a, b * b);
        for (uint32_t call = 1; sum >= 1.0f && call < 16; call++) {
            rn = philox_uniform4((uint32_t)p, 0u, call, sc->seed_lo, sc->seed_hi);
            a = between(rn.u[0], -1.0f, 1.0f); b = between(rn.u[1], -1.0f, 1.0f);
            sum = fma_(a, a, b * b);
            if (sum >= 1.0f) {
                a = between(rn.u[2], -1.0f, 1.0f); b = between(rn.u[3], -1.0f, 1.0f);
                sum = fma_(a, a, b * b);
            }
        }
        if (sum >= 1.0f) { a = 0.f; b = 0.f; sum = 0.f; }
        const float sq = sqrtf(1.0f - sum);
        float n[3] = {2.0f * a * sq, 2.0f * b * sq, 1.0f - 2.0f * sum};
        if (dot3(n, vv) > 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }   // vecOnHemisphere_cu :106-112
        const float depth = rf.f * rf.baseline / disp;
        n4.x = n[0]; n4.y = n[1]; n4.z = n[2];
        n4.w = plane_offset(rf, n, x, y, depth);
        n_out[p] = n4;
    } else {
        n4 = planes_in[p];
    }
    const PixelRef pr = hoist_reference<HR, TileT>(tile, tw, own, wts, hr, vr);
    float cost = TSAR_MAXCOST, rt = 0.f;
    int bv = -1;
    if (pr.textured) cost = multiview_cost<NB, HR, STRICT, QUAD, V>(sc, tile, tw, own, wts, pr, x, y, n4, bv, rt);
    c_out[p] = cost;
    if (!INIT) {
        if (beview_out) beview_out[p] = bv;
        if (ratio_out) ratio_out[p] = rt;
    }
}

template <int NB, int HR, bool STRICT, bool QUAD, bool INIT, int V = 0>
static int launch_full_t(tsar_ctx* ctx, const float4* planes, float* c, float4* n, int32_t* bv, float* rt) {
    const DevScene& hs = ctx->hscene;
    const int tiles_x = (hs.w + PM_RW - 1) / PM_RW, tiles_y = (hs.h + FULL_RH - 1) / FULL_RH;
    const int n_tiles = tiles_x * tiles_y;
    const size_t lds = tile_bytes<QUAD>(PM_RW + 2 * hs.hrad, FULL_RH + 2 * hs.vrad) + sizeof(float) * (size_t)(hs.hrad + 1) * (hs.vrad + 1) * PM_BLOCK;
    auto kern = pm_full_kernel<NB, HR, STRICT, QUAD, INIT, V>;
    if (lds > 64 * 1024) TSAR_HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ScopedKernelTimer tm(ctx, INIT ? "pm_init" : "pm_cost_planes");
        hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(PM_BLOCK), lds, ctx->stream, ctx->dscene, planes, c, n, bv, rt, tiles_x, n_tiles, strip_width(ctx->strip_w, tiles_x));
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

template <int NB, int HR, bool INIT>
static int launch_full_nh(tsar_ctx* ctx, const float4* planes, float* c, float4* n, int32_t* bv, float* rt) {
    const bool strict = ctx->hscene.flags & TSAR_FLAG_STRICT_DIV, quad = ctx->hscene.use_quad;
    const bool production = !(ctx->hscene.flags & TSAR_FLAG_TEX_FILTER_8BIT);   // the 8-bit filter mode runs the generic tap loop
    // the production configuration (8-bit quad textures, box 11, <= 2 best views) runs the sweep's tap loop (pm_core.h view_cost,
    // variant 250 (fast) / 122 (strict) / 114) in both arithmetic modes
    if (production && quad && NB == 2 && HR == 5 && ctx->variant == 250 && !strict) return launch_full_t<2, 5, false, true, INIT, 250>(ctx, planes, c, n, bv, rt);
    if (production && quad && NB == 2 && HR == 5 && (ctx->variant == 250 || ctx->variant == 122 || ctx->variant == 114)) {
        if (strict) return ctx->variant != 114 ? launch_full_t<2, 5, true, true, INIT, 122>(ctx, planes, c, n, bv, rt) : launch_full_t<2, 5, true, true, INIT, 114>(ctx, planes, c, n, bv, rt);
        return ctx->variant == 122 ? launch_full_t<2, 5, false, true, INIT, 122>(ctx, planes, c, n, bv, rt) : launch_full_t<2, 5, false, true, INIT, 114>(ctx, planes, c, n, bv, rt);
    }
    if (strict) return quad ? launch_full_t<NB, HR, true, true, INIT>(ctx, planes, c, n, bv, rt) : launch_full_t<NB, HR, true, false, INIT>(ctx, planes, c, n, bv, rt);
    return quad ? launch_full_t<NB, HR, false, true, INIT>(ctx, planes, c, n, bv, rt) : launch_full_t<NB, HR, false, false, INIT>(ctx, planes, c, n, bv, rt);
}

template <bool INIT>
static int launch_full(tsar_ctx* ctx, const float4* planes, float* c, float4* n, int32_t* bv, float* rt) {
    const DevScene& hs = ctx->hscene;
    const int need = hs.cost_comb == TSAR_COMB_BEST_N ? (hs.n_best < hs.n_sel ? hs.n_best : hs.n_sel) : hs.n_sel;
    const bool r5 = hs.hrad == 5 && hs.vrad == 5;
    if (need <= 2) return r5 ? launch_full_nh<2, 5, INIT>(ctx, planes, c, n, bv, rt) : launch_full_nh<2, 0, INIT>(ctx, planes, c, n, bv, rt);
    return r5 ? launch_full_nh<32, 5, INIT>(ctx, planes, c, n, bv, rt) : launch_full_nh<32, 0, INIT>(ctx, planes, c, n, bv, rt);
}

int launch_pm_init(tsar_ctx* ctx) { return launch_full<true>(ctx, nullptr, ctx->buf[0].c, ctx->buf[0].n4, nullptr, nullptr); }
int launch_pm_cost_planes(tsar_ctx* ctx, const float4* planes, float* cost, int32_t* beview, float* ratio) {
    return launch_full<false>(ctx, planes, cost, nullptr, beview, ratio);
}
