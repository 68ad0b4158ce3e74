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
    // structured buffer loads for the gathers (pm_tap_r5.h BUF) from the third sweep of a run on: measured per launch
    // (tools/launch_series.sh) they take 0.6 ms off a converged launch (37.6 -> 37.0) and add 4 ms to the first sweep after the
    // random initialisation (55.1 -> 59.1), where neighbouring lanes' footprints are unrelated, and 0.3 ms to the second (41.0 -> 41.3).
    // Fast mode on the difference texture (round 4: its cheaper blend outweighs the doubled footprint already on the SECOND
    // sweep, -1.7 ms per view; on the first it costs 28 ms): from sweep ctx->buffer_from = 1 on.
    const bool buf = ctx->buffer_gather && ctx->sweeps_done >= (strict ? (ctx->buffer_from > 2 ? ctx->buffer_from : 2) : ctx->buffer_from);
    // The production configuration (8-bit quad textures, box 11, <= 4 best views) runs the hand-scheduled tap loop of pm_tap_r5.h in
    // both arithmetic modes: variant 250 in fast mode (row-wise walk), 122 in strict mode (the oracle's column order; also the
    // column-order fast loop, TSAR_VARIANT=122), 114 where the D16 probe fails; + 131072 with buffer loads.
    const int v = ctx->variant;
    if constexpr ((NB == 2 || NB == 4) && HR == 5)
    if (quad && !(ctx->hscene.flags & TSAR_FLAG_TEX_FILTER_8BIT) && (v == 250 || v == 122 || (v == 114 && NB == 2))) {   // (the 8-bit filter mode runs the general-window loop)
#define SWEEP_R5(S, V, B) (buf ? launch_sweep_t<NB, 5, S, true, (V) | 131072, B>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<NB, 5, S, true, V, B>(ctx, colour, a, b, c, sid, dp, dr))
        // fast mode, buffer-load launches: the half-float difference texture (pm_tap_r5.h MIX) when tsar_set_views built it
        const bool mix = buf && !strict && v == 250 && ctx->hscene.n_sel > 0 && ctx->hscene.view[ctx->hscene.sel[0]].dquad != nullptr;
#define SWEEP_R5_FAST250(B) (mix ? launch_sweep_t<NB, 5, false, true, 250 | 131072 | 2097152, B>(ctx, colour, a, b, c, sid, dp, dr) : SWEEP_R5(false, 250, B))
        if constexpr (NB == 2) {
            // small images: 128-thread workgroups (see SWEEP_SMALL_IMAGE_TILES); TSAR_BLOCK=128|256 forces a shape (A/B runs)
            const int tiles256 = ((ctx->hscene.w + PM_RW - 1) / PM_RW) * ((ctx->hscene.h + 15) / 16);
            bool small = tiles256 < SWEEP_SMALL_IMAGE_TILES;
            if (ctx->force_block) small = ctx->force_block == 128;
            if (small && v != 114) {
                if (strict) return SWEEP_R5(true, 122, 128);
                return v == 250 ? SWEEP_R5_FAST250(128) : launch_sweep_t<2, 5, false, true, 122, 128>(ctx, colour, a, b, c, sid, dp, dr);
            }
            if (v == 114) return strict ? launch_sweep_t<2, 5, true, true, 114>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<2, 5, false, true, 114>(ctx, colour, a, b, c, sid, dp, dr);
        }
        if (strict) return SWEEP_R5(true, 122, PM_BLOCK);
        return v == 250 ? SWEEP_R5_FAST250(PM_BLOCK) : launch_sweep_t<NB, 5, false, true, 122>(ctx, colour, a, b, c, sid, dp, dr);
#undef SWEEP_R5_FAST250
#undef SWEEP_R5
    }
    constexpr int NBG = NB == 4 ? 32 : NB;      // the one-tap-at-a-time kernels exist for 2 and 32 best views
    if (strict) return quad ? launch_sweep_t<NBG, HR, true, true>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<NBG, HR, true, false>(ctx, colour, a, b, c, sid, dp, dr);
    return quad ? launch_sweep_t<NBG, HR, false, true>(ctx, colour, a, b, c, sid, dp, dr) : launch_sweep_t<NBG, HR, false, false>(ctx, colour, a, b, c, sid, dp, dr);
}

int launch_pm_sweep(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out, uint32_t stream_id,
                    int do_prop, int do_refine) {
    const DevScene& hs = ctx->hscene;
#ifdef TSAR_EXPERIMENTS
    {   // measured-and-rejected / diagnostic forms (pm_sweep_experiments.hip: TSAR_VARIANT / TSAR_VARIANT_NOW)
        int launched = 0;
        const int rc = launch_pm_sweep_experiment(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine, &launched);
        if (rc != TSAR_OK || launched) return rc;
    }
#endif
    const int need = hs.cost_comb == TSAR_COMB_BEST_N ? (hs.n_best < hs.n_sel ? hs.n_best : hs.n_sel) : hs.n_sel;
    const bool r5 = hs.hrad == 5 && hs.vrad == 5;
    // 8-bit imagery, any window but the box-11 / two-best-views configuration (which has its own tap loop): shared weight
    // table, chunked lines (pm_sweep_lut.hip)
    // (the box-11 loop filters with exact fp32 weights only: the 8-bit filter mode takes the general-window loop at box 11 too)
    const bool own_loop = r5 && need <= 4 && (need <= 2 || ctx->variant == 250 || ctx->variant == 122) && !(hs.flags & TSAR_FLAG_TEX_FILTER_8BIT);   // (TSAR_VARIANT=0 or an experiment: the generic loop below)
    if (lut_path_applies(ctx) && (!own_loop || lut_path_forced(ctx))) return launch_pm_sweep_lut(ctx, need, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    if (need <= 4 && need > 2 && r5) return launch_sweep_nh<4, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    if (need <= 2) return r5 ? launch_sweep_nh<2, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
                             : launch_sweep_nh<2, 0>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    return r5 ? launch_sweep_nh<32, 5>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
              : launch_sweep_nh<32, 0>(ctx, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
}
