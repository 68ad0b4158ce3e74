// pm_init_impl.h — the every-pixel kernel (random initialisation / scoring of given planes) and its launcher, as templates:
// pm_init.hip instantiates the production (box 11) and float-image configurations, pm_init_lut.hip the general-window ones.
#pragma once
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
    constexpr bool LUTW = (V & 1024) != 0;      // shared weight table first, then the window (pm_core_lut.h)
    const size_t lut_bytes = LUTW ? (size_t)(sc->lut_classes + 1) * 1024 : 0;
    TileT* tile = (TileT*)(lds_raw + lut_bytes);
    float* wts = LUTW ? (float*)lds_raw : (float*)(lds_raw + tile_bytes<QUAD>(tw, th)) + threadIdx.x;
    if constexpr (LUTW) build_weight_lut<PM_BLOCK>(sc, wts);
    const int t = xcd_tile(blockIdx.x, n_tiles);
    int tix, tiy;
    strip_tile(t, tiles_x, n_tiles / tiles_x, strip_w, tix, tiy);
    const int ty0 = tiy * FULL_RH, tx0 = tix * PM_RW;
    stage_ref_tile<FULL_RH, TileT>(sc, tile, tx0, ty0, hr, vr, LUTW ? LUT_TILE_PAD_ROWS : 0);
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
    PixelRef pr;
    if constexpr (LUTW) pr = hoist_reference_lut(sc, tile, tw, own, wts);
    else pr = hoist_reference<HR, TileT>(tile, tw, own, wts, hr, vr);
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
    const size_t lds = tile_bytes<QUAD>(PM_RW + 2 * hs.hrad, FULL_RH + 2 * hs.vrad + ((V & 1024) ? LUT_TILE_PAD_ROWS : 0)) +
                       ((V & 1024) ? (size_t)(hs.lut_classes + 1) * 1024 : sizeof(float) * (size_t)(hs.hrad + 1) * (hs.vrad + 1) * PM_BLOCK);
    auto kern = pm_full_kernel<NB, HR, STRICT, QUAD, INIT, V>;
    if (lds > 64 * 1024) TSAR_HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ScopedKernelTimer tm(ctx, INIT ? "pm_init" : "pm_cost_planes");
        hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(PM_BLOCK), lds, ctx->stream, ctx->dscene, planes, c, n, bv, rt, tiles_x, n_tiles, strip_width(ctx->strip_w, tiles_x));
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

