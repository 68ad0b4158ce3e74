// pm_sweep.hip — one red/black half-iteration of PatchMatch: spatial propagation + plane refinement
// of every pixel of one checkerboard colour, fused into ONE launch.
//
// Replaces gipuma_{black,red}_spatialProp_cu + gipuma_{black,red}_planeRefine_cu
// (reference gipuma.cu:846-1138; host loop :1744-1754 issues 4 launches + 4 device syncs per
// iteration, here 2 launches and no sync).  Refinement touches only the thread's own pixel, so fusing
// it behind propagation does not change any value.
//
// Determinism: neighbours are read from the launch-start state.  Opposite-colour pixels are not
// written by this launch; same-colour pixels (the six "V" taps of each near arm,
// gipuma.cu:958-1034) are read from `same_in` while results go to `same_out` (ping-pong), where the
// reference reads and writes the same array concurrently.
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

// Variant bit 3 of the fast tap loop relies on D16 LDS loads writing the whole destination register (zeros in the
// half that is not loaded), which is how gfx950 behaves with SRAM ECC enabled.  Checked once per context.
__global__ void d16_probe_kernel(uint32_t* out) {
    __shared__ unsigned short t[64];
    t[threadIdx.x] = (unsigned short)(0x4300u + threadIdx.x);
    __syncthreads();
    uint32_t r = 0xffffffffu;
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned short*)(t + threadIdx.x);
    asm volatile("ds_read_u16_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(addr));
    out[threadIdx.x] = r;
}
bool probe_d16_hi_zeroes(tsar_ctx* ctx) {
    uint32_t* d = nullptr;
    uint32_t h[64];
    if (hipMalloc((void**)&d, sizeof h) != hipSuccess) return false;
    hipLaunchKernelGGL(d16_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, d);
    bool ok = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    hipFree(d);
    for (int i = 0; ok && i < 64; i++) ok = h[i] == ((0x4300u + i) << 16);
    return ok;
}

DEVFN bool same_bits(const float4& a, const float4& b) {
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
           __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

template <int NB, int HR, bool STRICT, bool QUAD, int V = 0, int BLK = PM_BLOCK>
__global__ __launch_bounds__(BLK, ((V & 512) ? 1024 / BLK : 1)) void pm_sweep_kernel(const DevScene* __restrict__ sc, int colour,
                                                            const float* __restrict__ c_same, const float4* __restrict__ n_same,
                                                            const float* __restrict__ c_other, const float4* __restrict__ n_other,
                                                            float* c_out, float4* n_out, float* __restrict__ ratio_out,
                                                            int32_t* __restrict__ beview_out, uint32_t stream_id, int do_prop,
                                                            int do_refine, int tiles_x, int n_tiles, int cost_consistent, int strip_w,
                                                            const float* __restrict__ final_text) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename TileOf<QUAD>::type TileT;
    constexpr int SWEEP_RH = BLK / 16;
    const int hr = HR > 0 ? HR : sc->hrad, vr = HR > 0 ? HR : sc->vrad;
    const int tw = PM_RW + 2 * hr, th = SWEEP_RH + 2 * vr;
    TileT* tile = (TileT*)lds_raw;
    float* wts = (float*)(lds_raw + tile_bytes<QUAD>(tw, th)) + threadIdx.x;

    const int t = xcd_tile(blockIdx.x, n_tiles);
    int tix, tiy;
    strip_tile(t, tiles_x, n_tiles / tiles_x, strip_w, tix, tiy);
    const int ty0 = tiy * SWEEP_RH, tx0 = tix * PM_RW;
    stage_ref_tile<SWEEP_RH, TileT, BLK>(sc, tile, tx0, ty0, hr, vr);
    __syncthreads();

    const int ly = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int y = ty0 + ly;
    const int lx = 2 * k + ((colour + y) & 1);     // (x + y) & 1 == colour; gipuma.cu:1099-1103 / :1121-1125
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
    const PixelRef pr = hoist_reference<HR, TileT, BLK>(tile, tw, own, wts, hr, vr);
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
    constexpr int SWEEP_RH = BLK / 16;
    const int tiles_x = (hs.w + PM_RW - 1) / PM_RW, tiles_y = (hs.h + SWEEP_RH - 1) / SWEEP_RH;
    const int n_tiles = tiles_x * tiles_y;
    static const size_t lds_pad = getenv("TSAR_LDS_PAD") ? (size_t)atoi(getenv("TSAR_LDS_PAD")) : 0;   // occupancy experiments: unused LDS per workgroup
    const size_t lds = tile_bytes<QUAD>(PM_RW + 2 * hs.hrad, SWEEP_RH + 2 * hs.vrad) + sizeof(float) * (size_t)(hs.hrad + 1) * (hs.vrad + 1) * BLK + lds_pad;
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

template <int NB, int HR>
static int launch_sweep_nh(tsar_ctx* ctx, int colour, const PlaneBuf& a, const PlaneBuf& b, const PlaneBuf& c, uint32_t sid, int dp, int dr) {
    const bool strict = ctx->hscene.flags & TSAR_FLAG_STRICT_DIV, quad = ctx->hscene.use_quad;
    // The production configuration (8-bit quad textures, box 11, <= 2 best views) runs the hand-scheduled tap loop of
    // pm_core.h view_cost, in both arithmetic modes: variant 250 in fast mode (row-wise walk), 122 in strict mode, 114 where
    // the D16 probe fails.  In strict mode it is
    // the oracle's arithmetic (IEEE divides, min/max, floor) with the same loads, clamp-free loop and priorities: same
    // bits as the generic strict kernel.
    if (quad && NB == 2 && HR == 5 && !(ctx->hscene.flags & TSAR_FLAG_TEX_FILTER_8BIT)) {   // (the 8-bit filter mode runs the generic tap loop)
        // small images: 128-thread workgroups (see SWEEP_SMALL_IMAGE_TILES); TSAR_BLOCK=128|256 forces a shape (A/B runs)
        const int tiles256 = ((ctx->hscene.w + PM_RW - 1) / PM_RW) * ((ctx->hscene.h + 15) / 16);
        bool small = tiles256 < SWEEP_SMALL_IMAGE_TILES;
        if (const char* e = getenv("TSAR_BLOCK")) small = atoi(e) == 128;
        if (small && (ctx->variant == 250 || ctx->variant == 122)) {
            if (strict) return launch_sweep_t<2, 5, true, true, 122, 128>(ctx, colour, a, b, c, sid, dp, dr);
            return ctx->variant == 250 ? launch_sweep_t<2, 5, false, true, 250, 128>(ctx, colour, a, b, c, sid, dp, dr)
                                       : launch_sweep_t<2, 5, false, true, 122, 128>(ctx, colour, a, b, c, sid, dp, dr);
        }
        if (strict) {
            switch (ctx->variant) {
                case 250:       // the row-wise walk is a fast-mode liberty: strict runs the same loop in the oracle's column order
                case 122: return launch_sweep_t<2, 5, true, true, 122>(ctx, colour, a, b, c, sid, dp, dr);
                case 114: return launch_sweep_t<2, 5, true, true, 114>(ctx, colour, a, b, c, sid, dp, dr);
#ifdef TSAR_EXPERIMENTS
                case 58: return launch_sweep_t<2, 5, true, true, 58>(ctx, colour, a, b, c, sid, dp, dr);
                case 50: return launch_sweep_t<2, 5, true, true, 50>(ctx, colour, a, b, c, sid, dp, dr);
#endif
                default: break;
            }
        } else {
            switch (ctx->variant) {
                case 122: return launch_sweep_t<2, 5, false, true, 122>(ctx, colour, a, b, c, sid, dp, dr);
                case 114: return launch_sweep_t<2, 5, false, true, 114>(ctx, colour, a, b, c, sid, dp, dr);
                case 250: return launch_sweep_t<2, 5, false, true, 250>(ctx, colour, a, b, c, sid, dp, dr);
#ifdef TSAR_EXPERIMENTS   // earlier / diagnostic tap-loop variants (make TSAR_EXPERIMENTS=1)
                case 762: return launch_sweep_t<2, 5, false, true, 762>(ctx, colour, a, b, c, sid, dp, dr);
                case 506: return launch_sweep_t<2, 5, false, true, 506>(ctx, colour, a, b, c, sid, dp, dr);
                case 2: return launch_sweep_t<2, 5, false, true, 2>(ctx, colour, a, b, c, sid, dp, dr);
                case 6: return launch_sweep_t<2, 5, false, true, 6>(ctx, colour, a, b, c, sid, dp, dr);
                case 10: return launch_sweep_t<2, 5, false, true, 10>(ctx, colour, a, b, c, sid, dp, dr);
                case 18: return launch_sweep_t<2, 5, false, true, 18>(ctx, colour, a, b, c, sid, dp, dr);
                case 26: return launch_sweep_t<2, 5, false, true, 26>(ctx, colour, a, b, c, sid, dp, dr);
                case 50: return launch_sweep_t<2, 5, false, true, 50>(ctx, colour, a, b, c, sid, dp, dr);
                case 58: return launch_sweep_t<2, 5, false, true, 58>(ctx, colour, a, b, c, sid, dp, dr);
#endif
                default: break;
            }
        }
    }
    if (strict) return quad ? launch_sweep_t<NB, HR, true, true>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<NB, HR, true, false>(ctx, colour, a, b, c, sid, dp, dr);
    return quad ? launch_sweep_t<NB, HR, false, true>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<NB, HR, false, false>(ctx, colour, a, b, c, sid, dp, dr);
}

int launch_pm_sweep(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out, uint32_t stream_id,
                    int do_prop, int do_refine) {
    const DevScene& hs = ctx->hscene;
#ifdef TSAR_EXPERIMENTS
    if (ctx->lds_sweep && !ctx->final_text) {   // opt-in LDS-patch form for 8-bit imagery, box 11, n_best <= 2, <= 10 views (pm_sweep_lds.hip)
        int launched = 0;
        const int rc = launch_pm_sweep_lds(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine, &launched);
        if (rc != TSAR_OK || launched) return rc;
    }
#endif
    const int need = hs.cost_comb == TSAR_COMB_BEST_N ? (hs.n_best < hs.n_sel ? hs.n_best : hs.n_sel) : hs.n_sel;
    const bool r5 = hs.hrad == 5 && hs.vrad == 5;
    if (need <= 2) return r5 ? launch_sweep_nh<2, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
                             : launch_sweep_nh<2, 0>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    return r5 ? launch_sweep_nh<32, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
              : launch_sweep_nh<32, 0>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
}
