// pm_sweep_impl.h — the fused propagation + refinement kernel of one red/black half-iteration and its launcher, as templates:
// pm_sweep.hip instantiates the production (box 11) configurations, pm_sweep_lut*.hip the general-window ones.
#pragma once
#include "pm_core.h"

// Workgroup = BLK threads = a region of 32 x BLK/16 pixels, one thread per pixel of the active colour.  BLK = 256 (32 x 16) is
// the production shape; BLK = 128 (32 x 8) is used for small images, where 256-thread tiles number fewer than the ~1000
// workgroup slots of the chip and leave CUs idle or unevenly loaded (640 x 480: 600 tiles of 256, 1200 of 128).
#define SWEEP_SMALL_IMAGE_TILES 3072   // below this many 256-thread tiles the 128-thread shape is launched

// A candidate is the pixel index of the neighbour whose plane is tried, with bit 30 set if that neighbour has the
// active colour (its plane is read from same_in); -1 = arm skipped.
#define CAND_SAME (1 << 30)

// 8-arm adaptive candidate selection, gipuma.cu:874-1042.
DEVFN void select_candidates(const DevScene* __restrict__ sc, const float* __restrict__ c_same, const float* __restrict__ c_other,
                             int x, int y, int cand[8]) {
    const int col = sc->w, row = sc->h;
    const int p = y * col + x;
    const bool fix_seed = sc->flags & TSAR_FLAG_FIX_DOWN_FAR_SEED, fix_cmp = sc->flags & TSAR_FLAG_FIX_RIGHT_FAR_CMP;
    float cmin;
    int cp, cs;
#pragma unroll
    for (int k = 0; k < 8; k++) cand[k] = -1;
    // far arms: offsets 3, 5, ..., 23 along the axis -> always the other colour
    if (y > 2) {
        cp = p - 3 * col; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y > 2 + 2 * i) { const int q = p - (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[0] = cp;
    }
    if (y < row - 3) {
        cp = p + 3 * col;
        cmin = (fix_seed || y <= 2) ? c_other[cp] : c_other[p - 3 * col];   // gipuma.cu:906 seeds with c[up_far]
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y < row - 3 - 2 * i) { const int q = p + (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[1] = cp;
    }
    if (x > 2) {
        cp = p - 3; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (x > 2 + 2 * i) { const int q = p - 3 - 2 * i; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[2] = cp;
    }
    if (x < col - 3) {
        cp = p + 3; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (x < col - 3 - 2 * i) {
                const int q = p + 3 + 2 * i;
                const float v = c_other[q];
                const bool take = fix_cmp ? (v < cmin) : (cmin < v);             // gipuma.cu:943 is inverted
                if (take) { cmin = v; cp = q; }
            }
        cand[3] = cp;
    }
    // near arms: the 4-neighbour (other colour) and three V pairs (same colour)
    if (y > 0) {
        cp = p - col; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y > 1 + i && x > i) { const int q = p - (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (y > 1 + i && x < col - 1 - i) { const int q = p - (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[4] = cs ? (cp | CAND_SAME) : cp;
    }
    if (y < row - 1) {
        cp = p + col; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y < row - 2 - i && x > i) { const int q = p + (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (y < row - 2 - i && x < col - 1 - i) { const int q = p + (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[5] = cs ? (cp | CAND_SAME) : cp;
    }
    if (x > 0) {
        cp = p - 1; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x > 1 + i && y > i) { const int q = p - (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (x > 1 + i && y < row - 1 - i) { const int q = p - (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[6] = cs ? (cp | CAND_SAME) : cp;
    }
    if (x < col - 1) {
        cp = p + 1; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x < col - 2 - i && y > i) { const int q = p + (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (x < col - 2 - i && y < row - 1 - i) { const int q = p + (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[7] = cs ? (cp | CAND_SAME) : cp;
    }
}

DEVFN bool same_bits(const float4& a, const float4& b) {
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
           __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

template <int NB, int HR, bool STRICT, bool QUAD, int V = 0, int BLK = PM_BLOCK>
__global__ __launch_bounds__(BLK) void pm_sweep_kernel(const DevScene* __restrict__ sc, int colour,
                                                            const float* __restrict__ c_same, const float4* __restrict__ n_same,
                                                            const float* __restrict__ c_other, const float4* __restrict__ n_other,
                                                            float* c_out, float4* n_out, float* __restrict__ ratio_out,
                                                            int32_t* __restrict__ beview_out, uint32_t stream_id, int do_prop,
                                                            int do_refine, int tiles_x, int n_tiles, int cost_consistent, int strip_w,
                                                            const float* __restrict__ final_text) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename TileOf<QUAD>::type TileT;
    // region of a workgroup: 32 pixels wide, 2 BLK / 32 high (one thread per pixel of the active colour)
    constexpr int RW = PM_RW;
    constexpr int SWEEP_RH = 2 * BLK / RW;
    const int hr = HR > 0 ? HR : sc->hrad, vr = HR > 0 ? HR : sc->vrad;
    const int tw = RW + 2 * hr, th = SWEEP_RH + 2 * vr;
    // LDS: [reference window][S weights per thread], or (variant bit 10) [shared weight table][reference window]
    constexpr bool LUTW = (V & 1024) != 0;
    const size_t lut_bytes = LUTW ? (size_t)(sc->lut_classes + 1) * 1024 : 0;
    TileT* tile = (TileT*)(lds_raw + lut_bytes);
    float* wts = LUTW ? (float*)lds_raw : (float*)(lds_raw + tile_bytes<QUAD>(tw, th)) + threadIdx.x;
    if constexpr (LUTW) build_weight_lut<BLK>(sc, wts);

    const int t = xcd_tile(blockIdx.x, n_tiles);
    int tix, tiy;
    strip_tile(t, tiles_x, n_tiles / tiles_x, strip_w, tix, tiy);
    const int ty0 = tiy * SWEEP_RH, tx0 = tix * RW;
    stage_ref_tile<SWEEP_RH, TileT, BLK, RW>(sc, tile, tx0, ty0, hr, vr, LUTW ? LUT_TILE_PAD_ROWS : 0);
    __syncthreads();

    // lane -> pixel: a wave covers 4 rows x 16 pixels of the active colour (a 32 x 4 pixel strip).  The other shapes the round-1
    // review asked to have measured lose: 8 x 8 pixels per wave +9.6 %, 16 rows x 4 +61 %, 2 rows x 32 +0.7 % (profiles/r02/ab_lane_maps.json)
    const int ly = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int y = ty0 + ly;
    const int lx = 2 * k + ((colour + y) & 1);            // (x + y) & 1 == colour; gipuma.cu:1099-1103 / :1121-1125
    const int x = tx0 + lx;
    const int w = sc->w, h = sc->h;
    if (x >= w || y >= h) return;
    const int p = y * w + x;
    const int own = (ly + vr) * tw + lx + hr;

    float cost_now = c_same[p];
    float4 n_now = n_same[p];
    // the kernels' `final == true` mode (gipuma.cu:856, :1063): pixels whose lines->text is -1 keep their state
    // (copied across the ping-pong), and no accepted hypothesis writes ratio / beview (:559-562, :669-672)
    if (final_text && final_text[p] == -1.0f) { c_out[p] = cost_now; n_out[p] = n_now; return; }
    PixelRef pr;
    if constexpr (LUTW) pr = hoist_reference_lut(sc, tile, tw, own, wts);
    else pr = hoist_reference<HR, TileT, BLK>(tile, tw, own, wts, hr, vr);
    bool wrote = false;
    float ratio_w = 0.f;
    int beview_w = 0;
    if (pr.textured) {
        const DevRef& rf = sc->ref;
        float depth_now = plane_depth(rf, n_now, x, y);
        const float4 n_first = n_now;
        // One rolled loop over the hypotheses of this pixel: h = 0..7 the propagation arms in the reference's
        // order (gipuma.cu:874-1042), h = 8.. the refinement steps (:1066-1090).  The loop counter is wave-uniform,
        // so the arm/step switch is a scalar branch and the multi-view cost (the whole tap loop) exists once in the
        // binary instead of nine times: ~6 KB of hot code instead of ~45 KB, and fewer live registers.
        int cand[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // neighbour pixel index | same-colour flag << 30, -1 = arm skipped
        if (do_prop) select_candidates(sc, c_same, c_other, x, y, cand);
        float vv[3];
        view_vector(rf, x, y, vv);
        float deltaN = 1.0f;
        float deltaZ = sc->max_disp / 2.0f;
        const float fb = rf.f * rf.baseline;
        const int h_end = do_refine ? 8 + sc->refine_steps : 8;
#pragma unroll 1
        for (int h = do_prop ? 0 : 8; h < h_end; h++) {
            float4 n_t;
            float depth_t;
            if (h < 8) {
                int ci = cand[0];
#pragma unroll
                for (int a = 1; a < 8; a++) ci = (h == a) ? cand[a] : ci;
                if (ci < 0) continue;
                const int idx = ci & 0x3fffffff;
                n_t = (ci >> 30) ? n_same[idx] : n_other[idx];
                // A neighbour often carries the very plane this pixel already holds (or held when the launch
                // started): planes spread by verbatim copies.  While c[p] is the score of norm4[p] (true for
                // every state produced by init / sweeps) re-scoring it returns a cost that is not smaller
                // than cost_now, so the reference's `cost_before < *cost_now` (gipuma.cu:555) rejects it.
                if (cost_consistent && (same_bits(n_t, n_now) || same_bits(n_t, n_first))) continue;
                depth_t = plane_depth(rf, n_t, x, y);
                // spatialPropagation_cu gipuma.cu:524-566; the range test is done first: a
                // hypothesis outside [depthMin, depthMax] is never accepted, so it is not scored.
                if (!(depth_t >= rf.depthMin && depth_t <= rf.depthMax)) continue;
            } else {
                // planeRefinement_cu gipuma.cu:621-676 + getRndDispAndUnitVector_cu :582-619
                const Rand4 rn = philox_uniform4((uint32_t)p, stream_id, (uint32_t)(h - 8), sc->seed_lo, sc->seed_hi);
                const float disp = fb / depth_now;
                const float minDelta = -fminf(deltaZ, sc->min_disp + disp);   // "+" as written, gipuma.cu:601
                const float maxDelta = fminf(deltaZ, sc->max_disp - disp);
                const float dz = between(rn.u[0], minDelta, maxDelta);
                const float dispOut = fminf(fmaxf(disp + dz, sc->min_disp), sc->max_disp);
                depth_t = fb / dispOut;
                float nt[3];
                nt[0] = n_now.x + between(rn.u[1], -deltaN, deltaN);
                nt[1] = n_now.y + between(rn.u[2], -deltaN, deltaN);
                nt[2] = n_now.z + between(rn.u[3], -deltaN, deltaN);
                const float inv = 1.0f / sqrtf(dot3(nt, nt));
                nt[0] *= inv; nt[1] *= inv; nt[2] *= inv;
                if (dot3(nt, vv) > 0.0f) { nt[0] = -nt[0]; nt[1] = -nt[1]; nt[2] = -nt[2]; }
                n_t.x = nt[0]; n_t.y = nt[1]; n_t.z = nt[2];
                n_t.w = plane_offset(rf, nt, x, y, depth_t);
                deltaN = deltaN / 4.0f;
                deltaZ = deltaZ / 10.0f;
            }
            int bv; float rt;
            const float cost_t = multiview_cost<NB, HR, STRICT, QUAD, V, BLK>(sc, tile, tw, own, wts, pr, x, y, n_t, bv, rt);
            if (cost_t < cost_now) {
                cost_now = cost_t; n_now = n_t; depth_now = depth_t;
                ratio_w = rt; beview_w = bv; wrote = true;
            }
        }
    }
    c_out[p] = cost_now;
    n_out[p] = n_now;
    if (wrote && !final_text) { ratio_out[p] = ratio_w; beview_out[p] = beview_w; }
}


template <int NB, int HR, bool STRICT, bool QUAD, int V = 0, int BLK = PM_BLOCK>
static int launch_sweep_t(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out,
                          uint32_t stream_id, int do_prop, int do_refine) {
    const DevScene& hs = ctx->hscene;
    constexpr int RW = PM_RW;
    constexpr int SWEEP_RH = 2 * BLK / RW;
    const int tiles_x = (hs.w + RW - 1) / RW, tiles_y = (hs.h + SWEEP_RH - 1) / SWEEP_RH;
    const int n_tiles = tiles_x * tiles_y;
#ifdef TSAR_EXPERIMENTS
    const size_t lds_pad = ctx->lds_pad;       // occupancy experiments: unused LDS per workgroup
#else
    constexpr size_t lds_pad = 0;
#endif
    const size_t lds = tile_bytes<QUAD>(RW + 2 * hs.hrad, SWEEP_RH + 2 * hs.vrad + ((V & 1024) ? LUT_TILE_PAD_ROWS : 0)) + lds_pad +
                       ((V & 1024) ? (size_t)(hs.lut_classes + 1) * 1024 : sizeof(float) * (size_t)(hs.hrad + 1) * (hs.vrad + 1) * BLK);
    auto kern = pm_sweep_kernel<NB, HR, STRICT, QUAD, V, BLK>;
    if (lds > 64 * 1024) TSAR_HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ScopedKernelTimer tm(ctx, "pm_sweep");
        hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(BLK), lds, ctx->stream, ctx->dscene, colour, same_in.c, same_in.n4, other.c,
                           other.n4, same_out.c, same_out.n4, ctx->ratio, ctx->beview, stream_id, do_prop, do_refine, tiles_x, n_tiles,
                           ctx->cost_consistent ? 1 : 0, strip_width(ctx->strip_w, tiles_x), ctx->final_text);
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

