// pm_init_lut.hip — the every-pixel kernel (pm_init_impl.h) for every window other than the scripts' box 11, on 8-bit imagery
// (pm_core_lut.h): random initialisation and the scoring of caller-supplied planes.
#include "pm_init_impl.h"

#define LUT_V(ch) (1024 | ((ch) << 11))

template <int NB, bool STRICT, bool INIT>
static int launch_full_lut_nsi(tsar_ctx* ctx, int ch, const float4* planes, float* c, float4* n, int32_t* bv, float* rt) {
    switch (ch) {
        case 4: return launch_full_t<NB, 0, STRICT, true, INIT, LUT_V(4)>(ctx, planes, c, n, bv, rt);
        case 5: return launch_full_t<NB, 0, STRICT, true, INIT, LUT_V(5)>(ctx, planes, c, n, bv, rt);
        default: return launch_full_t<NB, 0, STRICT, true, INIT, LUT_V(6)>(ctx, planes, c, n, bv, rt);
    }
}
template <int NB, bool STRICT>
static int launch_full_lut_ns(tsar_ctx* ctx, int ch, bool init, const float4* planes, float* c, float4* n, int32_t* bv, float* rt) {
    return init ? launch_full_lut_nsi<NB, STRICT, true>(ctx, ch, planes, c, n, bv, rt) : launch_full_lut_nsi<NB, STRICT, false>(ctx, ch, planes, c, n, bv, rt);
}

int launch_pm_full_lut(tsar_ctx* ctx, int need, bool init, const float4* planes, float* c, float4* n, int32_t* bv, float* rt) {
    const DevScene& hs = ctx->hscene;
    const bool strict = hs.flags & TSAR_FLAG_STRICT_DIV;
    const int ch = hs.lut_chunk;
    if (need <= 2) return strict ? launch_full_lut_ns<2, true>(ctx, ch, init, planes, c, n, bv, rt) : launch_full_lut_ns<2, false>(ctx, ch, init, planes, c, n, bv, rt);
    if (need <= 4) return strict ? launch_full_lut_ns<4, true>(ctx, ch, init, planes, c, n, bv, rt) : launch_full_lut_ns<4, false>(ctx, ch, init, planes, c, n, bv, rt);
    return strict ? launch_full_lut_ns<32, true>(ctx, ch, init, planes, c, n, bv, rt) : launch_full_lut_ns<32, false>(ctx, ch, init, planes, c, n, bv, rt);
}
