// pm_init.hip — random plane initialisation (gipuma_init_cu2, reference gipuma.cu:678-729) and the
// diagnostic "score these planes" kernel (pmCostMultiview_cu over a caller-supplied plane map).
// Both visit every pixel: region 32 x 8 per 256-thread workgroup.
#include "pm_init_impl.h"

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
    if (lut_path_applies(ctx) && (!(r5 && need <= 2) || (hs.flags & TSAR_FLAG_TEX_FILTER_8BIT) || lut_path_forced(ctx))) return launch_pm_full_lut(ctx, need, INIT, planes, c, n, bv, rt);   // pm_init_lut.hip
    if (need <= 2) return r5 ? launch_full_nh<2, 5, INIT>(ctx, planes, c, n, bv, rt) : launch_full_nh<2, 0, INIT>(ctx, planes, c, n, bv, rt);
    return r5 ? launch_full_nh<32, 5, INIT>(ctx, planes, c, n, bv, rt) : launch_full_nh<32, 0, INIT>(ctx, planes, c, n, bv, rt);
}

int launch_pm_init(tsar_ctx* ctx) { return launch_full<true>(ctx, nullptr, ctx->buf[0].c, ctx->buf[0].n4, nullptr, nullptr); }
int launch_pm_cost_planes(tsar_ctx* ctx, const float4* planes, float* cost, int32_t* beview, float* ratio) {
    return launch_full<false>(ctx, planes, cost, nullptr, beview, ratio);
}
