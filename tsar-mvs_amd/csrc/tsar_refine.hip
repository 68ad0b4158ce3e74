// tsar_refine.hip — the TSAR confidence kernels that evaluate a matching cost:
// reverse-direction NCC (rlCost, reference gipuma.cu:300-392) and gipuma_getlrdiff (:1160-1186).
#include "pm_core.h"

#define EW_BLOCK 256

template <bool STRICT, bool QUAD>
DEVFN float reverse_cost(const DevScene* __restrict__ sc, const DevView& vw, int x, int y, const float4& n4) {
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    const int hr = sc->hrad, vr = sc->vrad;
    const DevView& rv = sc->view[0];
    float H[9], V[9];
    plane_homography(sc->ref, vw, n4, H, sc->k_sparse != 0);
    // inverse homography by the adjugate, gipuma.cu:316-337
    const float det = H[0] * H[4] * H[8] + H[1] * H[5] * H[6] + H[2] * H[3] * H[7] - H[2] * H[4] * H[6] - H[1] * H[3] * H[8] - H[0] * H[5] * H[7];
    V[0] = (H[4] * H[8] - H[5] * H[7]) / det;
    V[1] = -(H[1] * H[8] - H[2] * H[7]) / det;
    V[2] = (H[1] * H[5] - H[2] * H[4]) / det;
    V[3] = -(H[3] * H[8] - H[5] * H[6]) / det;
    V[4] = (H[0] * H[8] - H[2] * H[6]) / det;
    V[5] = -(H[0] * H[5] - H[2] * H[3]) / det;
    V[6] = (H[3] * H[7] - H[4] * H[6]) / det;
    V[7] = -(H[0] * H[7] - H[1] * H[6]) / det;
    V[8] = (H[0] * H[4] - H[1] * H[3]) / det;
    const float xf = (float)x, yf = (float)y;
    // getCorrespondingPoint_cu gipuma.cu:161-171 (matvecmul4noz: the two products first, the constant last; oracle S4: mul, fma, add)
    const float Zc = fma_(H[7], yf, H[6] * xf) + H[8];
    const float pcx = (fma_(H[1], yf, H[0] * xf) + H[2]) / Zc, pcy = (fma_(H[4], yf, H[3] * xf) + H[5]) / Zc;
    const bool q8 = (sc->flags & TSAR_FLAG_TEX_FILTER_8BIT) != 0;
    const float cen = sample_bilinear<QUAD>(vw, w, h, qp, pcx, pcy, q8);
    float sum_ref = 0.f, sum_ref_ref = 0.f, sum_src = 0.f, sum_src_src = 0.f, sum_ref_src = 0.f, wsum = 0.f;
    for (int i = -hr; i <= hr; i += 2)
        for (int j = -vr; j <= vr; j += 2) {
            // make_int2(pt_c.x + i, pt_c.y + j): float -> int truncation, gipuma.cu:355
            const float fx_ = fminf(fmaxf(pcx + (float)i, -2.0e9f), 2.0e9f), fy_ = fminf(fmaxf(pcy + (float)j, -2.0e9f), 2.0e9f);
            const int plx = (int)fx_, ply = (int)fy_;
            const float ref_pix = vw.img[(size_t)min(max(ply, 0), h - 1) * w + min(max(plx, 0), w - 1)];
            const float qx = (float)plx, qy = (float)ply;
            const float Z = fma_(V[7], qy, V[6] * qx) + V[8];
            const float X = fma_(V[1], qy, V[0] * qx) + V[2], Y = fma_(V[4], qy, V[3] * qx) + V[5];
            float u, v;
            if (STRICT) persp_divide_exact<true>(X, Y, Z, u, v);   // = X / Z, Y / Z bit for bit
            else { const float rz = __builtin_amdgcn_rcpf(Z); u = X * rz; v = Y * rz; }
            const float src_pix = sample_bilinear<QUAD>(rv, w, h, qp, u, v, q8);
            const float sd = sqrtf((float)(i * i + j * j));
            const float cd = fabsf(ref_pix - cen);
            const float wt = tsar_expf(-sd / 50.0f - cd / 18.0f);
            const float wr = wt * ref_pix, ws = wt * src_pix;
            sum_ref += wr;
            sum_ref_ref = fma_(wr, ref_pix, sum_ref_ref);
            sum_src += ws;
            sum_src_src = fma_(ws, src_pix, sum_src_src);
            sum_ref_src = fma_(wr, src_pix, sum_ref_src);
            wsum += wt;
        }
    const float inv = 1.0f / wsum;
    sum_ref *= inv; sum_ref_ref *= inv; sum_src *= inv; sum_src_src *= inv; sum_ref_src *= inv;
    const float var_ref = sum_ref_ref - sum_ref * sum_ref;
    const float var_src = sum_src_src - sum_src * sum_src;
    if (var_ref < 1e-5f || var_src < 1e-5f) return TSAR_MAXCOST;
    const float covar = sum_ref_src - sum_ref * sum_src;
    return fmaxf(0.0f, fminf(TSAR_MAXCOST, 1.0f - covar / sqrtf(var_ref * var_src)));
}

template <bool STRICT, bool QUAD>
__global__ __launch_bounds__(EW_BLOCK) void lrdiff_kernel(const DevScene* __restrict__ sc, const float* __restrict__ c,
                                                          const float4* __restrict__ n4, const int32_t* __restrict__ beview,
                                                          float* __restrict__ lrdiff) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (p >= w * h) return;
    const int y = p / w, x = p - y * w;
    const int v = beview[p];
    if (v < 1 || v >= TSAR_MAX_VIEWS || sc->view[v].img == nullptr) return;   // no best view recorded: lrdiff keeps its value
    const float rc = reverse_cost<STRICT, QUAD>(sc, sc->view[v], x, y, n4[p]);
    const float d = fabsf(c[p] - rc);
    lrdiff[p] = d > 1.0f ? 1.0f : d;
}

int launch_lrdiff(tsar_ctx* ctx) {
    const bool strict = ctx->hscene.flags & TSAR_FLAG_STRICT_DIV, quad = ctx->hscene.use_quad;
    const dim3 grid((ctx->w * ctx->h + EW_BLOCK - 1) / EW_BLOCK), block(EW_BLOCK);
    {
        ScopedKernelTimer tm(ctx, "lrdiff");
#define LR(S, Q) hipLaunchKernelGGL((lrdiff_kernel<S, Q>), grid, block, 0, ctx->stream, ctx->dscene, ctx->buf[0].c, ctx->buf[0].n4, ctx->beview, ctx->lrdiff)
        if (strict) { if (quad) LR(true, true); else LR(true, false); }
        else { if (quad) LR(false, true); else LR(false, false); }
#undef LR
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}
