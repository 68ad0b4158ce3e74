// pm_sweep_lut.hip — the sweep kernel (pm_sweep_impl.h) for every window other than the scripts' box 11, on 8-bit imagery:
// runtime radius, weights from the shared table, lines walked in chunks of 4 / 5 / 6 taps (pm_core_lut.h).
#include "pm_sweep_impl.h"

// Taps per chunk for a line of T taps: the fewest padding slots, the longer chunk on a tie (more gathers in flight per wave).
int lut_chunk_taps(int T) {
    int best = 6, waste = (6 - T % 6) % 6;
    for (int ch = 5; ch >= 4; ch--) {
        const int wst = (ch - T % ch) % ch;
        if (wst < waste) { waste = wst; best = ch; }
    }
    return best;
}

#define LUT_V(ch) (1024 | ((ch) << 11))

template <int NB, bool STRICT>
static int launch_sweep_lut_ns(tsar_ctx* ctx, int ch, int colour, const PlaneBuf& a, const PlaneBuf& b, const PlaneBuf& c, uint32_t sid, int dp, int dr) {
    // fast mode, from the third sweep of a run on: gathers as structured buffer loads (variant bit 17, see pm_sweep.hip)
    if (!STRICT && ctx->buffer_gather && ctx->sweeps_done >= ctx->buffer_from && ctx->hscene.n_sel > 0 && ctx->hscene.view[ctx->hscene.sel[0]].dquad != nullptr) {
        switch (ch) {      // + bit 21: the half-float difference texture (pm_tap_r5.h MIX) when tsar_set_views built it
            case 4: return launch_sweep_t<NB, 0, false, true, LUT_V(4) | 131072 | 2097152>(ctx, colour, a, b, c, sid, dp, dr);
            case 5: return launch_sweep_t<NB, 0, false, true, LUT_V(5) | 131072 | 2097152>(ctx, colour, a, b, c, sid, dp, dr);
            default: return launch_sweep_t<NB, 0, false, true, LUT_V(6) | 131072 | 2097152>(ctx, colour, a, b, c, sid, dp, dr);
        }
    }
    if (!STRICT && ctx->buffer_gather && ctx->sweeps_done >= ctx->buffer_from) {
        switch (ch) {
            case 4: return launch_sweep_t<NB, 0, false, true, LUT_V(4) | 131072>(ctx, colour, a, b, c, sid, dp, dr);
            case 5: return launch_sweep_t<NB, 0, false, true, LUT_V(5) | 131072>(ctx, colour, a, b, c, sid, dp, dr);
            default: return launch_sweep_t<NB, 0, false, true, LUT_V(6) | 131072>(ctx, colour, a, b, c, sid, dp, dr);
        }
    }
    switch (ch) {
        case 4: return launch_sweep_t<NB, 0, STRICT, true, LUT_V(4)>(ctx, colour, a, b, c, sid, dp, dr);
        case 5: return launch_sweep_t<NB, 0, STRICT, true, LUT_V(5)>(ctx, colour, a, b, c, sid, dp, dr);
        default: return launch_sweep_t<NB, 0, STRICT, true, LUT_V(6)>(ctx, colour, a, b, c, sid, dp, dr);
    }
}

// need: how many best views enter the cost (<= 2 / <= 4: selection in two / four registers, else the general one)
int launch_pm_sweep_lut(tsar_ctx* ctx, int need, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out, uint32_t stream_id,
                        int do_prop, int do_refine) {
    const DevScene& hs = ctx->hscene;
    const bool strict = hs.flags & TSAR_FLAG_STRICT_DIV;
    const int ch = hs.lut_chunk;
    if (need <= 2)
        return strict ? launch_sweep_lut_ns<2, true>(ctx, ch, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
                      : launch_sweep_lut_ns<2, false>(ctx, ch, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    if (need <= 4)
        return strict ? launch_sweep_lut_ns<4, true>(ctx, ch, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
                      : launch_sweep_lut_ns<4, false>(ctx, ch, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
    return strict ? launch_sweep_lut_ns<32, true>(ctx, ch, colour, same_in, other, same_out, stream_id, do_prop, do_refine)
                  : launch_sweep_lut_ns<32, false>(ctx, ch, colour, same_in, other, same_out, stream_id, do_prop, do_refine);
}
