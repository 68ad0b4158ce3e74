// pm_sweep_experiments.hip — dispatch of the measured-and-rejected / diagnostic tap-loop variants (pm_core_experiments.h, pm_tap_r5.h
// DIAG).  Built only into libtsar_hip_exp.so (`make TSAR_EXPERIMENTS=1`); selected with TSAR_VARIANT / TSAR_VARIANT_NOW.
#include "pm_sweep_impl.h"

int launch_pm_sweep_experiment(tsar_ctx* ctx, int colour, const PlaneBuf& a, const PlaneBuf& b, const PlaneBuf& c, uint32_t sid, int dp, int dr, int* launched) {
    *launched = 0;
    const DevScene& hs = ctx->hscene;
    const int need = hs.cost_comb == TSAR_COMB_BEST_N ? (hs.n_best < hs.n_sel ? hs.n_best : hs.n_sel) : hs.n_sel;
    const bool strict = hs.flags & TSAR_FLAG_STRICT_DIV;
    if (!(hs.use_quad && hs.hrad == 5 && hs.vrad == 5 && need <= 2) || (hs.flags & TSAR_FLAG_TEX_FILTER_8BIT)) return TSAR_OK;
#define EXP(S, V) case V: *launched = 1; return launch_sweep_t<2, 5, S, true, V>(ctx, colour, a, b, c, sid, dp, dr)
    // TSAR_VARIANT_NOW (read per launch): time a variant on a state the production kernels converged (tools/ab_converged.py) —
    // the wrong-result variants never converge on their own and would be measured in the random-plane regime
    int variant = getenv("TSAR_VARIANT_NOW") ? atoi(getenv("TSAR_VARIANT_NOW")) : ctx->variant;
    // the difference-texture loops read the views' dquad textures: without them (strict-built views, TSAR_MIX_GATHER=0) refuse
    if ((variant & 2097152) && !(hs.n_sel > 0 && hs.view[hs.sel[0]].dquad != nullptr)) {
        ctx->err = "TSAR_VARIANT: the difference-texture tap loop needs the views' dquad textures (fast mode, TSAR_MIX_GATHER on)";
        return TSAR_ERR_STATE;
    }
    if (strict) {
        switch (variant) {
            EXP(true, 58);
            EXP(true, 50);
            default: return TSAR_OK;
        }
    }
    switch (variant) {
        EXP(false, 762);       // 250 + gathers of line t+1 issued before line t is blended
        EXP(false, 655610);    // buffer loads + division-free corner test
        EXP(false, 131290);    // buffer loads, no wave priority
        EXP(false, 393466);    // 250 + buffer loads, issued back to back
        EXP(false, 65786);     // 250 + 64 x 8 region: 2 x 32 lanes per wave
        EXP(false, 16634);     // 250 + 8 x 8 lanes per wave
        EXP(false, 33018);     // 250 + 16 x 4 lanes per wave
        EXP(false, 506);       // WRONG RESULTS: the instruction mix of pairing two taps into one 16-byte gather
        EXP(false, 254);       // WRONG RESULTS: 250 without gathers (texel bits synthesised from the address): the VALU floor
        EXP(false, 131322);    // the production buffer-load loop in EVERY launch (the library uses it from the third sweep on)
        EXP(false, 2228474);   // the production difference-texture loop (pm_tap_r5.h MIX) in EVERY launch (needs the context's dquad textures)
        EXP(false, 6422778);   // WRONG RESULTS: the difference-texture loop without gathers: the VALU floor of the shipping body (pm_tap_r5.h DIAG 1)
        EXP(false, 10617082);  // WRONG RESULTS: the difference-texture loop with every gather replaced by an 8-byte LDS read (DIAG 2)
        EXP(false, 1048826);   // WRONG RESULTS: 250 with every gather replaced by an LDS read: the ceiling of an LDS-staged source patch
        EXP(false, 2);
        EXP(false, 6);
        EXP(false, 10);
        EXP(false, 18);
        EXP(false, 26);
        EXP(false, 50);
        EXP(false, 58);
        default: return TSAR_OK;
    }
#undef EXP
}
