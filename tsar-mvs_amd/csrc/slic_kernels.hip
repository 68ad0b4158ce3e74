// slic_kernels.hip — gSLICr superpixel segmentation as the reference configures it
// (reference gSLICr_Lib/engines/gSLICr_seg_engine_GPU.cu, gSLICr_seg_engine_shared.h,
// gSLICr_seg_engine.cpp:30-44; settings main.cpp:608-615).
//
// MI355X shape: the reference launches (superpixels x 15) workgroups of 256 threads for the centre
// update, most of them idle, plus a second kernel to add the 15 partials (GPU.cu:148-172).  Here one
// 256-thread workgroup owns one superpixel, walks the 15 sub-blocks of its 3S x 3S window itself and
// adds the partials in the same order, so the float sums are bit-identical to the reference's
// reduction tree while launching 15x fewer workgroups and no finalize kernel.  The 64-lane tail of the
// tree runs on wave shuffles instead of LDS.
#include "tsar_dev.h"

#define DEVFN __device__ __forceinline__
#define SL_BLOCK 256

struct Spixel {   // spixel_info, gSLICr_spixel_info.h:11-17
    float cx, cy;
    float col[4];
    int id, n;
};

// pow(x, 1.0f / 3.0f) of rgb2CIELab (shared.h:41-46): the CORRECTLY ROUNDED fp32 value of x^(0.3333333432674407958984375) (the
// exponent the reference passes is the float nearest 1/3), from IEEE operations only — the cube root in double-double (division-free
// Newton on the inverse cube root, then one step on the exact residual c^3 - x formed with fma) times x^delta, delta = (double)(1.0f / 3.0f) - 1/3, for which ln x is
// needed to ~1e-10 only.  Enumerated against powl on every argument an 8-bit colour can produce: 50 329 213 evaluations, 0
// mismatches (oracle/tsar_oracle_slic.c orc_pow_third_check, the same operation sequence; tests/test_slic_reference_golden.py).
// The reference compiled on a host calls glibc's powf, which differs from this on 0.07 % of them by one ulp; the Newton cube root
// of rounds 1-4 differed on 15 %.  Two fp64 divisions per evaluation (seven with the plain Newton of the first version), three evaluations per pixel of a quarter-resolution image.
DEVFN float pow_third(float xf) {
    const double x = (double)xf;
    double y = (double)(1.0f / __uint_as_float(__float_as_uint(xf) / 3u + 0x2a5137a0u));      // seed of x^(-1/3) from a 5 % cube-root seed
#pragma unroll
    for (int i = 0; i < 4; i++) y = y * ((4.0 - x * (y * y * y)) * (1.0 / 3.0));             // Newton on y^-3 = x: no division
    const double c = x * (y * y);
    const double c2 = c * c, e2 = fma(c, c, -c2);
    const double c3 = c2 * c, e3 = fma(c2, c, -c3);
    const double r = (c3 - x) + fma(e2, c, e3);
    const double lo = -r / (3.0 * c2);
    const unsigned long long xb = (unsigned long long)__double_as_longlong(x);
    const int k = (int)((xb >> 52) & 0x7ff) - 1023;
    const double m = __longlong_as_double((long long)((xb & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double p = 1.0 / 13.0;
    p = fma(p, t2, 1.0 / 11.0);
    p = fma(p, t2, 1.0 / 9.0);
    p = fma(p, t2, 1.0 / 7.0);
    p = fma(p, t2, 1.0 / 5.0);
    p = fma(p, t2, 1.0 / 3.0);
    p = fma(p, t2, 1.0);
    const double lnx = fma((double)k, 0.6931471805599453, 2.0 * t * p);
    const double delta = (double)(1.0f / 3.0f) - 1.0 / 3.0;
    const double u = delta * lnx;
    const double q = fma(0.5 * u, u, u);
    return (float)(c + fma(c, q, lo));
}

__global__ __launch_bounds__(SL_BLOCK) void slic_cvt_kernel(const uchar4* __restrict__ in, float4* __restrict__ out, int n, int color_space) {
    const int p = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (p >= n) return;
    const uchar4 px = in[p];   // b, g, r, a
    float4 o;
    if (color_space == 2) {    // RGB: raw channel values (shared.h:59-63)
        o = make_float4((float)px.x, (float)px.y, (float)px.z, 0.f);
    } else {
        const float _b = (float)px.x * 0.0039216f, _g = (float)px.y * 0.0039216f, _r = (float)px.z * 0.0039216f;
        const float x = _r * 0.412453f + _g * 0.357580f + _b * 0.180423f;
        const float y = _r * 0.212671f + _g * 0.715160f + _b * 0.072169f;
        const float z = _r * 0.019334f + _g * 0.119193f + _b * 0.950227f;
        if (color_space == 1) {
            o = make_float4(x, y, z, 0.f);
        } else {               // CIELAB shared.h:19-51
            const float epsilon = 0.008856f, kappa = 903.3f;
            const float xr = x / 0.950456f, yr = y / 1.0f, zr = z / 1.088754f;
            const float fx = xr > epsilon ? pow_third(xr) : (kappa * xr + 16.0f) / 116.0f;
            const float fy = yr > epsilon ? pow_third(yr) : (kappa * yr + 16.0f) / 116.0f;
            const float fz = zr > epsilon ? pow_third(zr) : (kappa * zr + 16.0f) / 116.0f;
            o = make_float4(116.0f * fy - 16.0f, 500.0f * (fx - fy), 200.0f * (fy - fz), 0.f);
        }
    }
    out[p] = o;
}

// init_cluster_centers_shared shared.h:73-90
__global__ void slic_init_kernel(const float4* __restrict__ lab, Spixel* __restrict__ sp, int w, int h, int mw, int mh, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mw * mh) return;
    const int x = i % mw, y = i / mw;
    int ix = x * S + S / 2, iy = y * S + S / 2;
    ix = ix >= w ? (x * S + w) / 2 : ix;
    iy = iy >= h ? (y * S + h) / 2 : iy;
    const float4 c = lab[(size_t)iy * w + ix];
    Spixel s;
    s.cx = (float)ix; s.cy = (float)iy;
    s.col[0] = c.x; s.col[1] = c.y; s.col[2] = c.z; s.col[3] = c.w;
    s.id = i; s.n = 0;
    sp[i] = s;
}

// find_center_association_shared + compute_slic_distance shared.h:92-134
__global__ __launch_bounds__(SL_BLOCK) void slic_assoc_kernel(const float4* __restrict__ lab, const Spixel* __restrict__ sp,
                                                              int32_t* __restrict__ idx, int w, int h, int mw, int mh, int S, float weight,
                                                              float norm_xy) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const float4 pix = lab[(size_t)y * w + x];
    const int cx = x / S, cy = y / S;
    int minidx = -1;
    float dist = 999999.9999f;
#pragma unroll
    for (int i = -1; i <= 1; i++)
#pragma unroll
        for (int j = -1; j <= 1; j++) {
            const int xx = cx + j, yy = cy + i;
            if (xx >= 0 && yy >= 0 && xx < mw && yy < mh) {
                const Spixel c = sp[yy * mw + xx];
                const float d0 = pix.x - c.col[0], d1 = pix.y - c.col[1], d2 = pix.z - c.col[2];
                const float dcolor = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
                const float ex = (float)x - c.cx, ey = (float)y - c.cy;
                const float dxy = sqrtf(ex * ex + ey * ey);
                const float t = dxy * norm_xy * weight;
                const float cd = sqrtf(dcolor * dcolor + t * t);
                if (cd < dist) { dist = cd; minidx = c.id; }
            }
        }
    if (minidx >= 0) idx[(size_t)y * w + x] = minidx;
}

struct Acc7 {   // colour (4) + xy (2) + count
    float c0, c1, c2, c3, px, py;
    int n;
};
DEVFN Acc7 acc_add(const Acc7& a, const Acc7& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2, a.c3 + b.c3, a.px + b.px, a.py + b.py, a.n + b.n}; }
DEVFN Acc7 acc_shfl_down(const Acc7& a, int d) {
    return {__shfl_down(a.c0, d), __shfl_down(a.c1, d), __shfl_down(a.c2, d), __shfl_down(a.c3, d), __shfl_down(a.px, d), __shfl_down(a.py, d), __shfl_down(a.n, d)};
}

// Update_Cluster_Center_device (GPU.cu:260-357) + finalize_reduction_result_shared (shared.h:151-173)
__global__ __launch_bounds__(SL_BLOCK) void slic_update_kernel(const float4* __restrict__ lab, const int32_t* __restrict__ idx,
                                                               Spixel* __restrict__ sp, int w, int h, int mw, int S, int nblk, int bpl) {
    __shared__ Acc7 sh[128];
    __shared__ int any_flag[2];
    const int id = blockIdx.x;
    const int sx = id % mw, sy = id / mw;
    const int l = threadIdx.x, tx = l & 15, ty = l >> 4;
    Acc7 total = {0, 0, 0, 0, 0, 0, 0};
    for (int bz = 0; bz < nblk; bz++) {
        if (l == 0) any_flag[bz & 1] = 0;
        __syncthreads();
        const int bx = bz % bpl, by = bz / bpl;
        const int xo = bx * 16 + tx, yo = by * 16 + ty;
        Acc7 v = {0, 0, 0, 0, 0, 0, 0};
        if (xo < S * 3 && yo < S * 3) {   // bpl = 3S/16 truncates: columns >= 16*bpl of the window are never visited (GPU.cu:160)
            const int xi = sx * S - S + xo, yi = sy * S - S + yo;
            if (xi >= 0 && xi < w && yi >= 0 && yi < h && idx[(size_t)yi * w + xi] == id) {
                const float4 c = lab[(size_t)yi * w + xi];
                v = {c.x, c.y, c.z, c.w, (float)xi, (float)yi, 1};
                any_flag[bz & 1] = 1;
            }
        }
        if (l >= 128) sh[l - 128] = v;
        __syncthreads();
        const bool any = any_flag[bz & 1] != 0;      // should_add
        if (any) {                                   // uniform across the workgroup
            if (l < 128) v = acc_add(v, sh[l]);      // s[l] += s[l + 128]
            __syncthreads();
            if (l >= 64 && l < 128) sh[l - 64] = v;
            __syncthreads();
            if (l < 64) {
                v = acc_add(v, sh[l]);               // s[l] += s[l + 64]
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v = acc_add(v, acc_shfl_down(v, d));   // lanes 0..31 in lockstep
            }
            if (l == 0) total = acc_add(total, v);   // finalize: partials added in block order
        }
        // no-add blocks contribute zeros: x + 0 == x exactly for the finite sums here
    }
    if (l == 0) {
        Spixel s;
        s.id = id;
        s.n = total.n;
        s.cx = total.px; s.cy = total.py;
        s.col[0] = total.c0; s.col[1] = total.c1; s.col[2] = total.c2; s.col[3] = total.c3;
        if (total.n != 0) {
            const float fn = (float)total.n;
            s.cx /= fn; s.cy /= fn;
            s.col[0] /= fn; s.col[1] /= fn; s.col[2] /= fn; s.col[3] /= fn;
        }
        sp[id] = s;
    }
}

// supress_local_lable shared.h:175-204
__global__ __launch_bounds__(SL_BLOCK) void slic_connect_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int w, int h) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const int cl = in[(size_t)y * w + x];
    if (x <= 1 || y <= 1 || x >= w - 2 || y >= h - 2) { out[(size_t)y * w + x] = cl; return; }
    int dc = 0, dl = -1;
#pragma unroll
    for (int j = -2; j <= 2; j++)
#pragma unroll
        for (int i = -2; i <= 2; i++) {
            const int nl = in[(size_t)(y + j) * w + (x + i)];
            if (nl != cl) { dl = nl; dc++; }
        }
    out[(size_t)y * w + x] = dc >= 16 ? dl : cl;
}

static int fail(tsar_ctx* ctx, int code, const char* msg) { ctx->err = msg; return code; }

extern "C" void tsar_default_slic_settings(tsar_slic_settings* s) {   // main.cpp:608-615
    if (!s) return;
    s->spixel_size = 20;
    s->no_iters = 5;
    s->coh_weight = 5.0f;
    s->do_enforce_connectivity = 0;
    s->color_space = 0;
}

extern "C" int tsar_slic(tsar_ctx* ctx, const uint8_t* bgra, int w, int h, const tsar_slic_settings* st, int32_t* labels_out, int mem) {
    if (!ctx) return TSAR_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, TSAR_ERR_HIP, "hipSetDevice failed");
    if (!bgra || !st || !labels_out) return fail(ctx, TSAR_ERR_INVALID, "bgra/settings/labels_out is NULL");
    const int S = st->spixel_size;
    if (S < 4 || S > 256 || w < S || h < S || st->no_iters < 0 || st->color_space < 0 || st->color_space > 2)
        return fail(ctx, TSAR_ERR_INVALID, "bad SLIC settings or image smaller than one superpixel");
    const size_t np = (size_t)w * h;
    const int mw = w / S, mh = h / S;                                  // (int)ceil(int / int), GPU.cu:70-71
    const int nblk = (int)ceilf((float)(S * S * 9) / 256.0f);          // no_grid_per_center GPU.cu:77-79
    const int bpl = S * 3 / 16 < 1 ? 1 : S * 3 / 16;                   // no_blocks_per_line GPU.cu:160
    // temporaries from the context's scratch arena (tsar_dev.h ScratchScope): five hipMalloc + hipFree per call cost 22 ms of
    // wall time around 1.3 ms of kernels at 1512 x 1008 (profiles/r02)
    ScratchScope scratch(ctx);
    int rc = TSAR_OK;
    auto cleanup = [&]() { hipStreamSynchronize(ctx->stream); scratch.release(); };
#define SL_TRY(e) do { if ((e) != hipSuccess) { ctx->err = #e " failed"; cleanup(); return TSAR_ERR_HIP; } } while (0)
    uchar4* d_in = (uchar4*)scratch.alloc(np * 4);
    float4* d_lab = (float4*)scratch.alloc(np * 16);
    int32_t* d_idx = (int32_t*)scratch.alloc(np * 4);
    int32_t* d_tmp = st->do_enforce_connectivity ? (int32_t*)scratch.alloc(np * 4) : nullptr;
    Spixel* d_sp = (Spixel*)scratch.alloc((size_t)mw * mh * sizeof(Spixel));
    if (!d_in || !d_lab || !d_idx || !d_sp || (st->do_enforce_connectivity && !d_tmp)) { cleanup(); return fail(ctx, TSAR_ERR_NOMEM, "device allocation failed"); }
    SL_TRY(hipMemcpyAsync(d_in, bgra, np * 4, mem == TSAR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    SL_TRY(hipMemsetAsync(d_idx, 0, np * 4, ctx->stream));
    const dim3 g1((unsigned)((np + SL_BLOCK - 1) / SL_BLOCK)), g2((w + 31) / 32, (h + 7) / 8), b(SL_BLOCK);
    { ScopedKernelTimer tm(ctx, "slic_cvt"); hipLaunchKernelGGL(slic_cvt_kernel, g1, b, 0, ctx->stream, d_in, d_lab, (int)np, st->color_space); }
    { ScopedKernelTimer tm(ctx, "slic_init"); hipLaunchKernelGGL(slic_init_kernel, dim3((mw * mh + 255) / 256), dim3(256), 0, ctx->stream, d_lab, d_sp, w, h, mw, mh, S); }
    const float norm_xy = 1.0f / (float)S;
    { ScopedKernelTimer tm(ctx, "slic_assoc"); hipLaunchKernelGGL(slic_assoc_kernel, g2, b, 0, ctx->stream, d_lab, d_sp, d_idx, w, h, mw, mh, S, st->coh_weight, norm_xy); }
    for (int it = 0; it < st->no_iters; it++) {
        { ScopedKernelTimer tm(ctx, "slic_update"); hipLaunchKernelGGL(slic_update_kernel, dim3(mw * mh), b, 0, ctx->stream, d_lab, d_idx, d_sp, w, h, mw, S, nblk, bpl); }
        { ScopedKernelTimer tm(ctx, "slic_assoc"); hipLaunchKernelGGL(slic_assoc_kernel, g2, b, 0, ctx->stream, d_lab, d_sp, d_idx, w, h, mw, mh, S, st->coh_weight, norm_xy); }
    }
    if (st->do_enforce_connectivity) {
        { ScopedKernelTimer tm(ctx, "slic_connect"); hipLaunchKernelGGL(slic_connect_kernel, g2, b, 0, ctx->stream, d_idx, d_tmp, w, h); }
        { ScopedKernelTimer tm(ctx, "slic_connect"); hipLaunchKernelGGL(slic_connect_kernel, g2, b, 0, ctx->stream, d_tmp, d_idx, w, h); }
    }
    SL_TRY(hipGetLastError());
    SL_TRY(hipMemcpyAsync(labels_out, d_idx, np * 4, mem == TSAR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    SL_TRY(hipStreamSynchronize(ctx->stream));
    cleanup();
    return rc;
}

// The stages of tsar_slic one at a time on caller-supplied HOST inputs (include/tsar.h "self-tests"): lets a test hold each kernel
// to the outputs of the reference's own functions compiled on a host (tests/golden/slic_ref.npz).  Centres are 32-byte records
// laid out like the reference's spixel_info (gSLICr_spixel_info.h:11-17) = Spixel above.
extern "C" int tsar_selftest_slic_stage(tsar_ctx* ctx, int stage, int w, int h, int mw, int mh, const tsar_slic_settings* st, const void* in0,
                                        const void* in1, void* inout) {
    if (!ctx) return TSAR_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, TSAR_ERR_HIP, "hipSetDevice failed");
    if (!st || !in0 || !inout || stage < 0 || stage > 4 || w < 1 || h < 1) return fail(ctx, TSAR_ERR_INVALID, "bad stage arguments");
    const int S = st->spixel_size;
    if (S < 4 || S > 256) return fail(ctx, TSAR_ERR_INVALID, "bad superpixel size");
    if ((stage >= 1 && stage <= 3) && (mw < 1 || mh < 1 || (size_t)mw * mh > (1u << 24))) return fail(ctx, TSAR_ERR_INVALID, "bad centre map size");
    if (stage == 2 && !in1) return fail(ctx, TSAR_ERR_INVALID, "stage 2 needs centres");
    if (stage == 3 && (!in1 || mw != w / S || mh != h / S)) return fail(ctx, TSAR_ERR_INVALID, "stage 3 needs labels and the engine's own map size");
    const size_t np = (size_t)w * h, nc = (size_t)mw * mh;
    ScratchScope scratch(ctx);
    auto cleanup = [&]() { hipStreamSynchronize(ctx->stream); scratch.release(); };
    const dim3 g1((unsigned)((np + SL_BLOCK - 1) / SL_BLOCK)), g2((w + 31) / 32, (h + 7) / 8), b(SL_BLOCK);
    const size_t in0_bytes = stage == 0 ? np * 4 : stage == 4 ? np * 4 : np * 16;
    const size_t in1_bytes = stage == 2 ? nc * sizeof(Spixel) : stage == 3 ? np * 4 : 0;
    const size_t io_bytes = stage == 0 ? np * 16 : (stage == 1 || stage == 3) ? nc * sizeof(Spixel) : np * 4;
    void* d0 = scratch.alloc(in0_bytes);
    void* d1 = in1_bytes ? scratch.alloc(in1_bytes) : nullptr;
    void* dio = scratch.alloc(io_bytes);
    if (!d0 || !dio || (in1_bytes && !d1)) { cleanup(); return fail(ctx, TSAR_ERR_NOMEM, "device allocation failed"); }
    SL_TRY(hipMemcpyAsync(d0, in0, in0_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (in1_bytes) SL_TRY(hipMemcpyAsync(d1, in1, in1_bytes, hipMemcpyHostToDevice, ctx->stream));
    SL_TRY(hipMemcpyAsync(dio, inout, io_bytes, hipMemcpyHostToDevice, ctx->stream));   // stage 2 keeps labels no centre claims
    switch (stage) {
    case 0: hipLaunchKernelGGL(slic_cvt_kernel, g1, b, 0, ctx->stream, (const uchar4*)d0, (float4*)dio, (int)np, st->color_space); break;
    case 1: hipLaunchKernelGGL(slic_init_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, (const float4*)d0, (Spixel*)dio, w, h, mw, mh, S); break;
    case 2: hipLaunchKernelGGL(slic_assoc_kernel, g2, b, 0, ctx->stream, (const float4*)d0, (const Spixel*)d1, (int32_t*)dio, w, h, mw, mh, S, st->coh_weight, 1.0f / (float)S); break;
    case 3: {
        const int nblk = (int)ceilf((float)(S * S * 9) / 256.0f), bpl = S * 3 / 16 < 1 ? 1 : S * 3 / 16;
        hipLaunchKernelGGL(slic_update_kernel, dim3((unsigned)nc), b, 0, ctx->stream, (const float4*)d0, (const int32_t*)d1, (Spixel*)dio, w, h, mw, S, nblk, bpl);
        break;
    }
    default: hipLaunchKernelGGL(slic_connect_kernel, g2, b, 0, ctx->stream, (const int32_t*)d0, (int32_t*)dio, w, h); break;
    }
    SL_TRY(hipGetLastError());
    SL_TRY(hipMemcpyAsync(inout, dio, io_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SL_TRY(hipStreamSynchronize(ctx->stream));
    cleanup();
    return TSAR_OK;
}
