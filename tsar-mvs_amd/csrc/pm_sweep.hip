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
#include "pm_sweep_impl.h"

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

template <int NB, int HR>
static int launch_sweep_nh(tsar_ctx* ctx, int colour, const PlaneBuf& a, const PlaneBuf& b, const PlaneBuf& c, uint32_t sid, int dp, int dr) {
    const bool strict = ctx->hscene.flags & TSAR_FLAG_STRICT_DIV, quad = ctx->hscene.use_quad;
    // structured buffer loads for the gathers (pm_core.h variant bit 17) from the third sweep of a run on: measured per launch
    // (tools/launch_series.sh) they take 0.6 ms off a converged launch (37.6 -> 37.0) and add 4 ms to the first sweep after the
    // random initialisation (55.1 -> 59.1), where neighbouring lanes' footprints are unrelated, and 0.3 ms to the second (41.0 -> 41.3)
    const bool buffer_gather = ctx->buffer_gather && ctx->sweeps_done >= 2;
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
            if (strict) return buffer_gather ? launch_sweep_t<2, 5, true, true, 131194, 128>(ctx, colour, a, b, c, sid, dp, dr)
                                             : launch_sweep_t<2, 5, true, true, 122, 128>(ctx, colour, a, b, c, sid, dp, dr);
            if (ctx->variant == 250 && buffer_gather) return launch_sweep_t<2, 5, false, true, 131322, 128>(ctx, colour, a, b, c, sid, dp, dr);
            return ctx->variant == 250 ? launch_sweep_t<2, 5, false, true, 250, 128>(ctx, colour, a, b, c, sid, dp, dr)
                                       : launch_sweep_t<2, 5, false, true, 122, 128>(ctx, colour, a, b, c, sid, dp, dr);
        }
        if (strict) {
            switch (ctx->variant) {
                case 250:       // the row-wise walk is a fast-mode liberty: strict runs the same loop in the oracle's column order
                case 122: return buffer_gather ? launch_sweep_t<2, 5, true, true, 131194>(ctx, colour, a, b, c, sid, dp, dr)    // 122 + bit 17
                                                : launch_sweep_t<2, 5, true, true, 122>(ctx, colour, a, b, c, sid, dp, dr);
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
                case 250:       // (+ bit 17: the gathers as structured buffer loads, -0.65 %; TSAR_BUFFER_GATHER=0 keeps global loads)
                    return buffer_gather ? launch_sweep_t<2, 5, false, true, 131322>(ctx, colour, a, b, c, sid, dp, dr)
                                              : launch_sweep_t<2, 5, false, true, 250>(ctx, colour, a, b, c, sid, dp, dr);
#ifdef TSAR_EXPERIMENTS   // earlier / diagnostic tap-loop variants (make TSAR_EXPERIMENTS=1)
                case 762: return launch_sweep_t<2, 5, false, true, 762>(ctx, colour, a, b, c, sid, dp, dr);
                case 655610: return launch_sweep_t<2, 5, false, true, 655610>(ctx, colour, a, b, c, sid, dp, dr);   // buffer loads + division-free corner test
                case 131290: return launch_sweep_t<2, 5, false, true, 131290>(ctx, colour, a, b, c, sid, dp, dr);   // buffer loads, no wave priority
                case 393466: return launch_sweep_t<2, 5, false, true, 393466>(ctx, colour, a, b, c, sid, dp, dr);   // 250 + buffer loads, issued back to back
                case 131322: return launch_sweep_t<2, 5, false, true, 131322>(ctx, colour, a, b, c, sid, dp, dr);   // 250 + buffer loads in every launch
                case 65786: return launch_sweep_t<2, 5, false, true, 65786>(ctx, colour, a, b, c, sid, dp, dr);   // 250 + 64 x 8 region: 2 x 32 lanes per wave
                case 16634: return launch_sweep_t<2, 5, false, true, 16634>(ctx, colour, a, b, c, sid, dp, dr);   // 250 + 8 x 8 lanes per wave
                case 33018: return launch_sweep_t<2, 5, false, true, 33018>(ctx, colour, a, b, c, sid, dp, dr);   // 250 + 16 x 4 lanes per wave
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
    // box 11 with three or four best views: the same tap loop, four-register selection (256-thread workgroups only)
    if (quad && NB == 4 && HR == 5 && !(ctx->hscene.flags & TSAR_FLAG_TEX_FILTER_8BIT) && (ctx->variant == 250 || ctx->variant == 122)) {
        if (strict) return buffer_gather ? launch_sweep_t<4, 5, true, true, 131194>(ctx, colour, a, b, c, sid, dp, dr)
                                         : launch_sweep_t<4, 5, true, true, 122>(ctx, colour, a, b, c, sid, dp, dr);
        if (ctx->variant == 250 && buffer_gather) return launch_sweep_t<4, 5, false, true, 131322>(ctx, colour, a, b, c, sid, dp, dr);
        return ctx->variant == 250 ? launch_sweep_t<4, 5, false, true, 250>(ctx, colour, a, b, c, sid, dp, dr)
                                   : launch_sweep_t<4, 5, false, true, 122>(ctx, colour, a, b, c, sid, dp, dr);
    }
    constexpr int NBG = NB == 4 ? 32 : NB;      // the one-tap-at-a-time kernels exist for 2 and 32 best views
    if (strict) return quad ? launch_sweep_t<NBG, HR, true, true>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<NBG, HR, true, false>(ctx, colour, a, b, c, sid, dp, dr);
    return quad ? launch_sweep_t<NBG, HR, false, true>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<NBG, HR, false, false>(ctx, colour, a, b, c, sid, dp, dr);
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
    // 8-bit imagery, any window but the box-11 / two-best-views configuration (which has its own tap loop): shared weight
    // table, chunked lines (pm_sweep_lut.hip)
    // (the box-11 loop filters with exact fp32 weights only: the 8-bit filter mode takes the general-window loop at box 11 too)
    const bool own_loop = r5 && need <= 4 && (need <= 2 || ctx->variant == 250 || ctx->variant == 122) && !(hs.flags & TSAR_FLAG_TEX_FILTER_8BIT);
    if (lut_path_applies(ctx) && (!own_loop || lut_path_forced())) return launch_pm_sweep_lut(ctx, need, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    if (need <= 4 && need > 2 && r5) return launch_sweep_nh<4, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    if (need <= 2) return r5 ? launch_sweep_nh<2, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
                             : launch_sweep_nh<2, 0>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    return r5 ? launch_sweep_nh<32, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
              : launch_sweep_nh<32, 0>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
}
