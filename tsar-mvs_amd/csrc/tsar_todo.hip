// entry points whose kernels land later this round (weighted median filter, region RANSAC, SLIC)
#include "tsar_dev.h"
static int nyi(tsar_ctx* ctx, const char* what) {
    if (ctx) ctx->err = std::string(what) + ": not implemented yet";
    return TSAR_ERR_STATE;
}
extern "C" int tsar_wmf(tsar_ctx* ctx, int, int) { return nyi(ctx, "tsar_wmf"); }
extern "C" int tsar_ransac_regions(tsar_ctx* ctx, float*, float*) { return nyi(ctx, "tsar_ransac_regions"); }
extern "C" void tsar_default_slic_settings(tsar_slic_settings* s) {
    if (!s) return;
    s->spixel_size = 20; s->no_iters = 5; s->coh_weight = 5.0f; s->do_enforce_connectivity = 0; s->color_space = 0;
}
extern "C" int tsar_slic(tsar_ctx* ctx, const uint8_t*, int, int, const tsar_slic_settings*, int32_t*, int) { return nyi(ctx, "tsar_slic"); }
