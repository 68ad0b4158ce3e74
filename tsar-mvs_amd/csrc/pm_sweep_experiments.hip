// pm_sweep_experiments.hip — dispatch of the diagnostic tap-loop variants (pm_tap_r5.h DIAG: the ceilings of profiles/r04) and of the
// production variants forced into every launch.  Built only into libtsar_hip_exp.so (`make TSAR_EXPERIMENTS=1`); selected with
// TSAR_VARIANT / TSAR_VARIANT_NOW.  (The measured-and-rejected tap loops of rounds 1-3 — lane maps, paired gathers, pipelined lines,
// the LDS-patch sweep — were removed in round 4; their numbers are in profiles/r01-r03 and DESIGN.md section 4.)
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
    if (strict) return TSAR_OK;
    switch (variant) {
        EXP(false, 131322);    // the production buffer-load loop on the byte texture in EVERY launch (the library uses the difference texture from the second sweep on)
        EXP(false, 2228474);   // the production difference-texture loop (pm_tap_r5.h MIX) in EVERY launch (needs the context's dquad textures)
        EXP(false, 6422778);   // WRONG RESULTS: the difference-texture loop without gathers: what the shipping body costs without its gather path (pm_tap_r5.h DIAG 1)
        EXP(false, 10617082);  // WRONG RESULTS: the difference-texture loop with every gather replaced by an 8-byte LDS read (DIAG 2): the ceiling of LDS-staged source patches
        default: return TSAR_OK;
    }
#undef EXP
}
