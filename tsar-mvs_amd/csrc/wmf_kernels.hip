// wmf_kernels.hip — multi-scale weighted median filter of the TSAR refinement:
// gipuma_WMF (reliability detection, reference gipuma.cu:1499-1698) and gipuma_WMF_Final
// (fill of unreliable pixels in textured regions, :1294-1497).
//
// The reference bubble-sorts four (value, weight) lists of up to 121 taps per pixel in 5 kB of
// per-thread local memory, O(n^2) data-moving swaps.  Here the same sorted order is obtained without
// moving data: each tap's rank is counted (stable: ties keep tap order) from an LDS copy of the list, and the
// lists are walked in rank order.  The reference's sort also touches slot `num` (a zero entry joins and the largest entry
// drops out); that is reproduced by ranking num+1 entries and walking the first num.
// Neighbours are read from launch-start copies of scale / depth / planes (the reference reads what
// other threads of the same launch are writing).
#include "tsar_device_math.h"

#define WMF_BLOCK 64
#define WMF_CAP 146

#define WMF_ROWS 123   // 11 x 11 taps + the zero slot + 1

struct WmfTaps {
    float w[WMF_CAP];
    float d[WMF_CAP], x[WMF_CAP], y[WMF_CAP], z[WMF_CAP];
    int n[WMF_CAP];
    int num;
};

// Per-workgroup staging for the O(n^2) ranking: the list being ranked, [entry][thread], and the resulting order.
// The tap arrays above live in scratch (written once, read O(n) times); ranking them from scratch cost ~120 k scratch
// loads per pixel (1.04 s per launch at 24 Mpixel), from LDS it is a conflict-free ds_read per comparison.
struct WmfLds {
    float v[WMF_ROWS * WMF_BLOCK];
    unsigned char pos[WMF_ROWS * WMF_BLOCK];
};

// pos[r] = index of the entry with stable rank r among entries 0..num (entry num is the zero slot).
// Four entries are ranked per pass over the list: one LDS read feeds four independent compare/add chains.
DEVFN void rank_order(const float* v, int num, WmfLds& l) {
    const int tid = threadIdx.x;
    for (int k = 0; k <= num; k++) l.v[k * WMF_BLOCK + tid] = v[k];
    for (int k0 = 0; k0 <= num; k0 += 4) {
        float vk[4];
        int r[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; c++) vk[c] = l.v[min(k0 + c, num) * WMF_BLOCK + tid];
        for (int j = 0; j < k0; j++) {                               // entries before all four: ties sort first
            const float vj = l.v[j * WMF_BLOCK + tid];
#pragma unroll
            for (int c = 0; c < 4; c++) r[c] += vj <= vk[c];
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {                             // the four themselves
            const int j = k0 + jj;
            if (j > num) break;
            const float vj = l.v[j * WMF_BLOCK + tid];
#pragma unroll
            for (int c = 0; c < 4; c++) r[c] += (jj < c) ? (vj <= vk[c]) : ((jj > c) ? (vj < vk[c]) : 0);
        }
        for (int j = k0 + 4; j <= num; j++) {                        // entries after all four
            const float vj = l.v[j * WMF_BLOCK + tid];
#pragma unroll
            for (int c = 0; c < 4; c++) r[c] += vj < vk[c];
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (k0 + c <= num) l.pos[r[c] * WMF_BLOCK + tid] = (unsigned char)(k0 + c);
    }
}
DEVFN int pos_at(const WmfLds& l, int i) { return l.pos[i * WMF_BLOCK + threadIdx.x]; }
DEVFN float weighted_median(const float* v, const float* w, const WmfLds& l, int num, float half) {
    float acc = 0.f;
    for (int i = 0; i < num; i++) {
        const int k = pos_at(l, i);
        acc += w[k];
        if (acc >= half) return v[k];
    }
    return v[pos_at(l, num - 1)];
}

DEVFN int collect_taps(const DevScene* __restrict__ sc, const float* __restrict__ scale_in, const float* __restrict__ depth_in,
                       const float4* __restrict__ n_in, int x, int y, int radius, int gap, float sdiv, WmfTaps& t) {
    const float* __restrict__ img = sc->view[0].img;
    const int w = sc->w, h = sc->h;
    const float cen = img[(size_t)y * w + x];
    int num = 0;
    for (int i = -radius; i <= radius; i += gap)
        for (int j = -radius; j <= radius; j += gap) {
            const int px = x + i, py = y + j;
            if (px < 0 || px >= w || py < 0 || py >= h) continue;
            const size_t q = (size_t)py * w + px;
            if (scale_in[q] != 1.0f) continue;
            const float cd = fabsf(img[q] - cen);
            const float sd = sqrtf((float)(i * i + j * j)) / sdiv;
            t.w[num] = tsar_expf(-sd / 4.0f) * tsar_expf(-cd / 9.0f);   // sigma_spatial 2, sigma_color 3 (gipuma.cu:1537-1550)
            t.d[num] = depth_in[q];
            t.n[num] = (int)q;
            const float4 nn = n_in[q];
            t.x[num] = nn.x; t.y[num] = nn.y; t.z[num] = nn.z;
            num++;
        }
    // the zero slot the reference's sort drags in
    t.w[num] = 0.f; t.d[num] = 0.f; t.x[num] = 0.f; t.y[num] = 0.f; t.z[num] = 0.f; t.n[num] = 0;
    t.num = num;
    return num;
}

// plane through the weighted-median-depth tap with the per-component weighted-median normal
DEVFN bool median_plane(const DevScene* __restrict__ sc, const float* __restrict__ depth_in, WmfTaps& t, WmfLds& l, float4& out) {
    const DevRef& rf = sc->ref;
    const int num = t.num;
    rank_order(t.d, num, l);
    float wsum = 0.f;
    for (int i = 0; i < num; i++) wsum += t.w[pos_at(l, i)];
    const float half = wsum / 2.f;
    int weimid = -1;
    {
        float acc = 0.f;
        for (int i = 0; i < num; i++) {
            const int k = pos_at(l, i);
            acc += t.w[k];
            if (acc >= half) { weimid = t.n[k]; break; }
        }
    }
    float nm[3];
    rank_order(t.x, num, l);
    nm[0] = weighted_median(t.x, t.w, l, num, half);
    rank_order(t.y, num, l);
    nm[1] = weighted_median(t.y, t.w, l, num, half);
    rank_order(t.z, num, l);
    nm[2] = weighted_median(t.z, t.w, l, num, half);
    if (weimid < 0) return false;
    const float depth_mid = rf.f * rf.baseline / depth_in[weimid];
    const double nrm = (double)sqrtf(dot3(nm, nm));   // `double xyzsqr = sqrtf(..)`, gipuma.cu:1663-1666
    nm[0] = (float)((double)nm[0] / nrm);
    nm[1] = (float)((double)nm[1] / nrm);
    nm[2] = (float)((double)nm[2] / nrm);
    out.x = nm[0]; out.y = nm[1]; out.z = nm[2];
    out.w = plane_offset(rf, nm, weimid % sc->w, weimid / sc->w, depth_mid);
    return true;
}

__global__ __launch_bounds__(WMF_BLOCK) void wmf_detect_kernel(const DevScene* __restrict__ sc, const float* __restrict__ scale_in,
                                                               const float* __restrict__ depth, const float4* __restrict__ n4,
                                                               float* __restrict__ scale_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_BLOCK + threadIdx.x;
    if (p >= w * h) return;
    const int y = p / w, x = p - y * w;
    const int po = 1 << iter, repo = 1 << (3 - iter);
    const int radius = 80 / po, gap = 16 / po, ths = 24 / po;
    __shared__ WmfLds lds;
    WmfTaps t;
    float4 nm;
    float s = 0.0f;
    if (collect_taps(sc, scale_in, depth, n4, x, y, radius, gap, (float)repo, t) > 0 && median_plane(sc, depth, t, lds, nm)) {
        const DevRef& rf = sc->ref;
        const float fb = rf.f * rf.baseline;
        const float disp_now = fb / plane_depth(rf, nm, x, y);
        const float disp_org = fb / plane_depth(rf, n4[p], x, y);
        s = fabsf(disp_now - disp_org) > (float)ths ? 0.0f : 1.0f;      // DEPTH_THS_MIN/MAX are 0 (gipuma.cu:38-39)
    }
    scale_out[p] = s;
}

__global__ __launch_bounds__(WMF_BLOCK) void wmf_fill_kernel(const DevScene* __restrict__ sc, const int32_t* __restrict__ canny,
                                                             const float* __restrict__ region_text, const float* __restrict__ scale_in,
                                                             const float* __restrict__ depth_in, const float4* __restrict__ n_in,
                                                             float* __restrict__ scale_out, float* __restrict__ depth_out,
                                                             float4* __restrict__ n_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_BLOCK + threadIdx.x;
    if (p >= w * h) return;
    if (!(region_text[canny[p]] == 1.0f && scale_in[p] == 0.0f)) return;
    const int y = p / w, x = p - y * w;
    const int po = 1 << iter;
    const int radius = 5 * po, gap = po, ths = 32 / po;
    __shared__ WmfLds lds;
    WmfTaps t;
    float4 nm;
    const int num = collect_taps(sc, scale_in, depth_in, n_in, x, y, radius, gap, (float)po, t);
    if (num < ths || num == 0) return;
    if (!median_plane(sc, depth_in, t, lds, nm)) return;
    const DevRef& rf = sc->ref;
    n_out[p] = nm;
    const float disp = rf.f * rf.baseline / plane_depth(rf, nm, x, y);
    if (disp <= sc->min_disp || disp >= sc->max_disp) { scale_out[p] = 0.0f; depth_out[p] = sc->min_disp; }
    else { scale_out[p] = 1.0f; depth_out[p] = disp; }
}

// iters launches of gipuma_WMF (final_pass = 0; the reference's loop runs 4, gipuma.cu:1809-1812) or of
// gipuma_WMF_Final (final_pass = 1; 6 in the reference, :1844-1847)
extern "C" int tsar_wmf(tsar_ctx* ctx, int iters, int final_pass) {
    if (!ctx) return TSAR_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return TSAR_ERR_HIP; }
    if (!ctx->have_state) { ctx->err = "no plane state"; return TSAR_ERR_STATE; }
    if (iters < 1 || iters > (final_pass ? 6 : 4)) { ctx->err = "tsar_wmf: iters must be 1..4 (detect) or 1..6 (final)"; return TSAR_ERR_INVALID; }
    if (final_pass && ctx->n_regions < 1) { ctx->err = "tsar_set_regions has not been called"; return TSAR_ERR_STATE; }
    const size_t np = (size_t)ctx->w * ctx->h;
    float *scale_snap = nullptr, *depth_snap = nullptr;
    if (hipMalloc((void**)&scale_snap, np * 4) != hipSuccess || (final_pass && hipMalloc((void**)&depth_snap, np * 4) != hipSuccess)) {
        hipFree(scale_snap);
        ctx->err = "hipMalloc failed";
        return TSAR_ERR_NOMEM;
    }
    const dim3 grid((unsigned)((np + WMF_BLOCK - 1) / WMF_BLOCK)), block(WMF_BLOCK);
    int rc = TSAR_OK;
    for (int it = 0; it < iters && rc == TSAR_OK; it++) {
        hipMemcpyAsync(scale_snap, ctx->scale, np * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (final_pass) {
            hipMemcpyAsync(depth_snap, ctx->depth, np * 4, hipMemcpyDeviceToDevice, ctx->stream);
            hipMemcpyAsync(ctx->buf[1].n4, ctx->buf[0].n4, np * 16, hipMemcpyDeviceToDevice, ctx->stream);
            ScopedKernelTimer tm(ctx, "wmf_fill");
            hipLaunchKernelGGL(wmf_fill_kernel, grid, block, 0, ctx->stream, ctx->dscene, ctx->canny, ctx->region_text, scale_snap, depth_snap,
                               ctx->buf[1].n4, ctx->scale, ctx->depth, ctx->buf[0].n4, it);
        } else {
            ScopedKernelTimer tm(ctx, "wmf_detect");
            hipLaunchKernelGGL(wmf_detect_kernel, grid, block, 0, ctx->stream, ctx->dscene, scale_snap, ctx->depth, ctx->buf[0].n4, ctx->scale, it);
        }
        if (hipGetLastError() != hipSuccess) { ctx->err = "wmf launch failed"; rc = TSAR_ERR_HIP; }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) { ctx->err = "wmf kernel failed"; rc = TSAR_ERR_HIP; }
    hipFree(scale_snap);
    hipFree(depth_snap);
    ctx->have_out = false;
    if (final_pass) ctx->cost_consistent = false;
    return rc;
}
