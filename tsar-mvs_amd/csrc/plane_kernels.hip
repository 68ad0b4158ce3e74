// plane_kernels.hip — the streaming (HBM-bound) per-pixel kernels: texel-quad packing, plane <-> depth
// conversions and the textureless plane fill.  One thread per pixel, 16-byte plane accesses,
// row-major order, so every wave reads/writes whole 1 KiB / 256 B segments.
#include "tsar_device_math.h"

#define EW_BLOCK 256

// ---- 2x2 texel quads -------------------------------------------------------------------------
// quad[(j0+1)][(i0+1)], i0 in [-1, w], j0 in [-1, h], packs the four texels a bilinear tap with
// floor(u) = i0, floor(v) = j0 needs, clamp addressing baked in: byte0 T(i0,j0), byte1 T(i0+1,j0),
// byte2 T(i0,j0+1), byte3 T(i0+1,j0+1).  This is what stands in for the texture unit the reference
// relies on (main.cpp:1190-1228): one dword gather per tap instead of four.
__global__ __launch_bounds__(EW_BLOCK) void build_quad_kernel(const float* __restrict__ img, uint32_t* __restrict__ quad, int w, int h,
                                                              int* __restrict__ nonintegral) {
    const int qw = w + 2, qh = h + 2;
    const int64_t n = (int64_t)qw * qh;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * EW_BLOCK) {
        const int qy = (int)(k / qw), qx = (int)(k - (int64_t)qy * qw);
        const int x0 = min(max(qx - 1, 0), w - 1), x1 = min(max(qx, 0), w - 1);
        const int y0 = min(max(qy - 1, 0), h - 1), y1 = min(max(qy, 0), h - 1);
        const float t00 = img[(size_t)y0 * w + x0], t10 = img[(size_t)y0 * w + x1];
        const float t01 = img[(size_t)y1 * w + x0], t11 = img[(size_t)y1 * w + x1];
        bad |= !(t00 >= 0.f && t00 <= 255.f && t00 == floorf(t00));
        const uint32_t b0 = (uint32_t)t00 & 0xffu, b1 = (uint32_t)t10 & 0xffu, b2 = (uint32_t)t01 & 0xffu, b3 = (uint32_t)t11 & 0xffu;
        quad[k] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(nonintegral, 1);
}

int launch_build_quad(tsar_ctx* ctx, const float* img, uint32_t* quad, int w, int h, int* nonintegral_flag) {
    const int64_t n = (int64_t)(w + 2) * (h + 2);
    const int grid = (int)((n + EW_BLOCK - 1) / EW_BLOCK < 4096 ? (n + EW_BLOCK - 1) / EW_BLOCK : 4096);
    {
        ScopedKernelTimer tm(ctx, "build_quad");
        hipLaunchKernelGGL(build_quad_kernel, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, img, quad, w, h, nonintegral_flag);
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

// tsar_set_views_u8: the 8-bit decode widened on the device (what the reference does on the host, convertTo(CV_32F), main.cpp:1423):
// a quarter of the bytes cross PCIe and the caller never holds 4-byte copies of its images.  16 pixels per thread.
__global__ __launch_bounds__(EW_BLOCK) void expand_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * EW_BLOCK + threadIdx.x) * 16;
    if (i + 16 <= n && (((uintptr_t)(in + i)) & 15) == 0) {
        const uint4 v = *(const uint4*)(in + i);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            *(float4*)(out + i + 4 * k) = make_float4((float)(wds[k] & 0xffu), (float)((wds[k] >> 8) & 0xffu), (float)((wds[k] >> 16) & 0xffu), (float)(wds[k] >> 24));
    } else {
        for (size_t k = i; k < n && k < i + 16; k++) out[k] = (float)in[k];
    }
}
int launch_expand_u8(tsar_ctx* ctx, const uint8_t* in, float* out, size_t n) {
    const size_t threads = (n + 15) / 16;
    {
        ScopedKernelTimer tm(ctx, "expand_u8");
        hipLaunchKernelGGL(expand_u8_kernel, dim3((unsigned)((threads + EW_BLOCK - 1) / EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream, in, out, n);
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

// The same quads as four halfs per entry, (t00, d1 = t10 - t00, d2 = t01 - t00, d3 = t11 - t10 - t01 + t00): what the fast
// arithmetic's blend (t00 + ax d1) + ay (d2 + ax d3) consumes (oracle S7 (6)).  All four are integers of magnitude <= 510, exact in
// fp16, so three v_fma_mix_f32 read them straight out of the gathered 8 bytes — no byte converts, no subtractions (pm_tap_r5.h MIX).
__global__ __launch_bounds__(EW_BLOCK) void build_dquad_kernel(const uint32_t* __restrict__ quad, uint2* __restrict__ dquad, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * EW_BLOCK) {
        const uint32_t q = quad[k];
        const int t00 = q & 0xff, t10 = (q >> 8) & 0xff, t01 = (q >> 16) & 0xff, t11 = q >> 24;
        const _Float16 h0 = (_Float16)(float)t00, h1 = (_Float16)(float)(t10 - t00), h2 = (_Float16)(float)(t01 - t00), h3 = (_Float16)(float)(t11 - t10 - t01 + t00);
        uint2 o;
        o.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
        o.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
        dquad[k] = o;
    }
}
int launch_build_dquad(tsar_ctx* ctx, const uint32_t* quad, uint2* dquad, int w, int h) {
    const int64_t n = (int64_t)(w + 2) * (h + 2);
    const int grid = (int)((n + EW_BLOCK - 1) / EW_BLOCK < 4096 ? (n + EW_BLOCK - 1) / EW_BLOCK : 4096);
    {
        ScopedKernelTimer tm(ctx, "build_quad");
        hipLaunchKernelGGL(build_dquad_kernel, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, quad, dquad, n);
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

// ---- plane <-> depth -------------------------------------------------------------------------
#define PIXEL_LOOP_BEGIN                                                                             \
    const int w = sc->w, h = sc->h;                                                                  \
    const int p = blockIdx.x * EW_BLOCK + threadIdx.x;                                               \
    if (p >= w * h) return;                                                                          \
    const int y = p / w, x = p - y * w;

// host fill main.cpp:1479-1490 + gipuma_get_disp gipuma.cu:731-755
__global__ __launch_bounds__(EW_BLOCK) void get_disp_kernel(const DevScene* __restrict__ sc, const float* __restrict__ depth_in,
                                                            const float* __restrict__ normal_world, float* __restrict__ c,
                                                            float4* __restrict__ n4, float* __restrict__ depth_plane) {
    PIXEL_LOOP_BEGIN
    const DevRef& rf = sc->ref;
    const float nw[3] = {normal_world[3 * (size_t)p], normal_world[3 * (size_t)p + 1], normal_world[3 * (size_t)p + 2]};
    float n[3];
    mat3vec(rf.Rorig, nw, n);
    const float fb = rf.f * rf.baseline;
    const float disp = fb / depth_in[p];       // lines->depth = f*b/depth (main.cpp:1488)
    depth_plane[p] = disp;
    const float depth = fb / disp;             // gipuma.cu:751-752
    float4 o;
    o.x = n[0]; o.y = n[1]; o.z = n[2];
    o.w = plane_offset(rf, n, x, y, depth);
    n4[p] = o;
    c[p] = 1.0f;
}

// gipuma_compute_disp gipuma.cu:810-844: out4 = (R_orig^-1 n, depth or 0 where c == MAXCOST)
__global__ __launch_bounds__(EW_BLOCK) void compute_disp_kernel(const DevScene* __restrict__ sc, const float* __restrict__ c,
                                                                const float4* __restrict__ n4, float4* __restrict__ out4) {
    PIXEL_LOOP_BEGIN
    const DevRef& rf = sc->ref;
    const float4 nn = n4[p];
    const float n[3] = {nn.x, nn.y, nn.z};
    float o[3];
    mat3vec(rf.RorigInv, n, o);
    float4 r;
    r.x = o[0]; r.y = o[1]; r.z = o[2];
    r.w = (c[p] != TSAR_MAXCOST) ? plane_depth(rf, nn, x, y) : 0.0f;
    out4[p] = r;
}

// gipuma_compute_disp_final gipuma.cu:757-808
__global__ __launch_bounds__(EW_BLOCK) void compute_disp_final_kernel(const DevScene* __restrict__ sc, const float* __restrict__ c,
                                                                      float4* __restrict__ n4, const float4* __restrict__ resize4,
                                                                      const float* __restrict__ text, float* __restrict__ depth,
                                                                      float4* __restrict__ out4) {
    PIXEL_LOOP_BEGIN
    const DevRef& rf = sc->ref;
    float4 nn = n4[p];
    const float4 rs = resize4[p];
    const float fb = rf.f * rf.baseline;
    const float disp_now = fb / plane_depth(rf, nn, x, y);
    const float disp_org = fb / plane_depth(rf, rs, x, y);
    const float tx = text[p];
    if ((fabsf(disp_now - disp_org) > 6.0f && tx == 1.0f) || tx == -1.0f) nn = rs;
    const float d = plane_depth(rf, nn, x, y);
    const float n[3] = {nn.x, nn.y, nn.z};
    if (d > rf.depthMax) nn.w = plane_offset(rf, n, x, y, rf.depthMax);
    if (d < rf.depthMin) nn.w = plane_offset(rf, n, x, y, rf.depthMin);
    const float dd = plane_depth(rf, nn, x, y);
    depth[p] = dd;
    n4[p] = nn;
    float o[3];
    mat3vec(rf.RorigInv, n, o);
    float4 r;
    r.x = o[0]; r.y = o[1]; r.z = o[2];
    r.w = (c[p] != TSAR_MAXCOST) ? dd : 0.0f;
    out4[p] = r;
}

// gipuma_dptow gipuma.cu:1140-1158
__global__ __launch_bounds__(EW_BLOCK) void depth_to_plane_kernel(const DevScene* __restrict__ sc, const float* __restrict__ depth,
                                                                  float4* __restrict__ n4) {
    PIXEL_LOOP_BEGIN
    const DevRef& rf = sc->ref;
    float4 nn = n4[p];
    const float n[3] = {nn.x, nn.y, nn.z};
    const float disp = rf.f * rf.baseline / depth[p];
    nn.w = plane_offset(rf, n, x, y, disp);
    n4[p] = nn;
}

// gipuma_getview gipuma.cu:1188-1213
__global__ __launch_bounds__(EW_BLOCK) void getview_kernel(const DevScene* __restrict__ sc, const float* __restrict__ c,
                                                           const float4* __restrict__ n4, const float* __restrict__ lrdiff,
                                                           float* __restrict__ confid, float* __restrict__ depth) {
    PIXEL_LOOP_BEGIN
    const DevRef& rf = sc->ref;
    confid[p] = ((2.0f - c[p]) / 2.0f + (1.0f - lrdiff[p])) / 2.0f;
    const float d = plane_depth(rf, n4[p], x, y);
    depth[p] = rf.f * rf.baseline / d;
}

DEVFN float4 region_plane(const DevRef& rf, const float4* __restrict__ region_n4, int rg, int x, int y) {
    float vv[3];
    view_vector(rf, x, y, vv);
    float4 nn = region_n4[rg];
    const float dp = nn.x * vv[0] + nn.y * vv[1] + nn.z * vv[2];
    if (dp > 0.0f) { nn.x *= -1; nn.y *= -1; nn.z *= -1; nn.w *= -1; }
    return nn;
}

// gipuma_update_scale gipuma.cu:1215-1259
__global__ __launch_bounds__(EW_BLOCK) void update_scale_kernel(const DevScene* __restrict__ sc, const int32_t* __restrict__ canny,
                                                                const float* __restrict__ region_text, const float4* __restrict__ region_n4,
                                                                float* __restrict__ c, float4* __restrict__ n4, float* __restrict__ scale,
                                                                float* __restrict__ depth) {
    PIXEL_LOOP_BEGIN
    const DevRef& rf = sc->ref;
    const int rg = canny[p];
    float4 nn;
    if (region_text[rg] == -1.0f) {
        c[p] = 0.0f;
        scale[p] = 1.0f;
        nn = region_plane(rf, region_n4, rg, x, y);
        n4[p] = nn;
    } else {
        nn = n4[p];
    }
    depth[p] = rf.f * rf.baseline / plane_depth(rf, nn, x, y);
}

// gipuma_update_scale_2 gipuma.cu:1261-1292
__global__ __launch_bounds__(EW_BLOCK) void fake_depth_kernel(const DevScene* __restrict__ sc, const int32_t* __restrict__ canny,
                                                              const float* __restrict__ region_text, const float4* __restrict__ region_n4,
                                                              float* __restrict__ fakedepth) {
    PIXEL_LOOP_BEGIN
    const DevRef& rf = sc->ref;
    const int rg = canny[p];
    if (region_text[rg] == -1.0f) fakedepth[p] = plane_depth(rf, region_plane(rf, region_n4, rg, x, y), x, y);
}

// copy-out main.cpp:1785-1795: (n_world, depth) -> separate depth / normal maps
__global__ __launch_bounds__(EW_BLOCK) void split_out4_kernel(const float4* __restrict__ out4, float* __restrict__ depth,
                                                              float* __restrict__ normal3, int n) {
    const int p = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (p >= n) return;
    const float4 v = out4[p];
    if (depth) depth[p] = v.w;
    if (normal3) { normal3[3 * (size_t)p] = v.x; normal3[3 * (size_t)p + 1] = v.y; normal3[3 * (size_t)p + 2] = v.z; }
}

#define EW_GRID(ctx) dim3(((ctx)->w * (ctx)->h + EW_BLOCK - 1) / EW_BLOCK)
#define EW_LAUNCH(ctx, name, kern, ...)                                                          \
    do {                                                                                         \
        {                                                                                        \
            ScopedKernelTimer tm(ctx, name);                                                     \
            hipLaunchKernelGGL(kern, EW_GRID(ctx), dim3(EW_BLOCK), 0, (ctx)->stream, __VA_ARGS__); \
        }                                                                                        \
        TSAR_HIP_TRY(ctx, hipGetLastError());                                                    \
        return TSAR_OK;                                                                          \
    } while (0)

int launch_get_disp(tsar_ctx* ctx, const float* depth_in, const float* normal_world) {
    EW_LAUNCH(ctx, "get_disp", get_disp_kernel, ctx->dscene, depth_in, normal_world, ctx->buf[0].c, ctx->buf[0].n4, ctx->depth);
}
int launch_compute_disp(tsar_ctx* ctx) { EW_LAUNCH(ctx, "compute_disp", compute_disp_kernel, ctx->dscene, ctx->buf[0].c, ctx->buf[0].n4, ctx->out4); }
int launch_compute_disp_final(tsar_ctx* ctx, const float4* resize4, const float* text) {
    EW_LAUNCH(ctx, "compute_disp_final", compute_disp_final_kernel, ctx->dscene, ctx->buf[0].c, ctx->buf[0].n4, resize4, text, ctx->depth, ctx->out4);
}
int launch_depth_to_plane(tsar_ctx* ctx) { EW_LAUNCH(ctx, "depth_to_plane", depth_to_plane_kernel, ctx->dscene, ctx->depth, ctx->buf[0].n4); }
int launch_getview(tsar_ctx* ctx) {
    EW_LAUNCH(ctx, "getview", getview_kernel, ctx->dscene, ctx->buf[0].c, ctx->buf[0].n4, ctx->lrdiff, ctx->confid, ctx->depth);
}
int launch_update_scale(tsar_ctx* ctx) {
    EW_LAUNCH(ctx, "update_scale", update_scale_kernel, ctx->dscene, ctx->canny, ctx->region_text, ctx->region_n4, ctx->buf[0].c, ctx->buf[0].n4,
              ctx->scale, ctx->depth);
}
int launch_fake_depth(tsar_ctx* ctx) {
    EW_LAUNCH(ctx, "fake_depth", fake_depth_kernel, ctx->dscene, ctx->canny, ctx->region_text, ctx->region_n4, ctx->fakedepth);
}
int launch_split_out4(tsar_ctx* ctx, float* depth, float* normal3) {
    EW_LAUNCH(ctx, "split_out4", split_out4_kernel, ctx->out4, depth, normal3, ctx->w * ctx->h);
}

// ---- label range (tsar_set_regions validation) ---------------------------------------------------
__global__ __launch_bounds__(EW_BLOCK) void label_range_kernel(const int32_t* __restrict__ labels, size_t n, int32_t* __restrict__ lohi) {
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    for (size_t k = (size_t)blockIdx.x * EW_BLOCK + threadIdx.x; k < n; k += (size_t)gridDim.x * EW_BLOCK) {
        const int32_t v = labels[k];
        lo = min(lo, v);
        hi = max(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&lohi[0], lo); atomicMax(&lohi[1], hi); }
}
int launch_label_range(tsar_ctx* ctx, const int32_t* labels, size_t n, int32_t* lo, int32_t* hi) {
    int32_t* d = nullptr;
    if (hipMalloc((void**)&d, 2 * sizeof(int32_t)) != hipSuccess) { ctx->err = "hipMalloc failed"; return TSAR_ERR_NOMEM; }
    const int32_t init[2] = {INT32_MAX, INT32_MIN};
    int32_t out[2] = {0, 0};
    const size_t blocks = (n + EW_BLOCK - 1) / EW_BLOCK;
    hipError_t e = hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(label_range_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(EW_BLOCK), 0, ctx->stream, labels, n, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof out, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e != hipSuccess) { ctx->err = std::string("label range check: ") + hipGetErrorString(e); return TSAR_ERR_HIP; }
    *lo = out[0];
    *hi = out[1];
    return TSAR_OK;
}
