// tsar_device_math.h — per-hypothesis geometry and RNG for the gfx950 kernels.
// Compiled with -ffp-contract=off: a fused multiply-add exists only where __builtin_fmaf is
// written, every other operation is a single IEEE-754 fp32 operation (hipcc keeps '/' and sqrtf
// correctly rounded without fast-math).  That makes all per-hypothesis quantities (planes, depths,
// random perturbations) reproducible bit for bit on any IEEE machine.
#pragma once
#include "tsar_dev.h"

#define DEVFN __device__ __forceinline__

DEVFN float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEVFN float dot3(const float* a, const float* b) { return fma_(a[2], b[2], fma_(a[1], b[1], a[0] * b[0])); }
DEVFN void mat3vec(const float* m, const float* v, float* o) {
    o[0] = dot3(m, v);
    o[1] = dot3(m + 3, v);
    o[2] = dot3(m + 6, v);
}
DEVFN void mat3mul(const float* a, const float* b, float* o) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) o[r * 3 + c] = fma_(a[r * 3 + 2], b[6 + c], fma_(a[r * 3 + 1], b[3 + c], a[r * 3] * b[c]));
}

// exp(x), x <= 0: bilateral weight (reference gipuma.cu:268).  Cody-Waite + degree-5 polynomial,
// ~1 ulp; v_rndne + 9 fma, no v_exp_f32 so the value does not depend on the hardware's
// transcendental unit.
DEVFN float tsar_expf(float x) {
    x = fmaxf(x, -87.0f);
    float k = rintf(x * 1.44269504f);
    float r = fma_(k, -0.693145752f, x);
    r = fma_(k, -1.42860677e-6f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float y = fma_(p, r * r, r) + 1.0f;
    return y * __uint_as_float((uint32_t)((int)k + 127) << 23);
}

// The perspective divide of a tap, u = X / Z and v = Y / Z (getCorrespondingPoint_cu gipuma.cu:161-171, vecdiv4), correctly rounded
// like the oracle's two IEEE divisions but without the compiler's v_div_scale / v_div_fmas / v_div_fixup sequence (12 VALU + 1
// transcendental per quotient): one v_rcp_f32 and one Newton step shared by both quotients, then per quotient q = n r,
// s = n - d q (exact), q' = q + s r: 8 VALU + 1 transcendental per tap.  That q' is the correctly rounded quotient is not argued but
// enumerated: whether it is depends on the two 23-bit mantissas only while every intermediate stays normal (scaling an operand by a
// power of two scales every step exactly), and tools/div_exact.hip compares q' with `/` for all 2^46 mantissa pairs on the GPU this
// runs on — 0 mismatches (profiles/r03/div_exact_all_mantissa_pairs.json; without the Newton step 47 045 pairs differ).
// Operand guard: |X|, |Y|, |Z| in [2^-20, 2^39).  Then q in (2^-60, 2^60), r in (2^-40, 2^21), the residual s is a multiple of
// 2^(-20 - 23 - 60 - 23) = 2^-126 (zero or normal, exactly representable) and e = 1 - d r0 a multiple of 2^-47: nothing underflows
// or overflows.  Outside the guard (a tap on the image's zero column, a plane seen edge-on, NaN) the wave takes the IEEE division.
// The guard is wave-uniform and compares magnitudes, not exponents: v_min3 / v_max3 over the three operands, two compares.
#define TSAR_DIV_GUARD_LO 9.5367431640625e-07f      // 2^-20
#define TSAR_DIV_GUARD_HI 274877906944.0f           // 2^38 (bound on the magnitudes: everything below 2^39 would do)
DEVFN bool div_guard_ok(float X, float Y, float Z) {
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(X), __builtin_fabsf(Y)), __builtin_fabsf(Z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(X), __builtin_fabsf(Y)), __builtin_fabsf(Z));
    // NaN operands: v_min / v_max return the other operand, so a NaN may pass — harmless, both forms then return NaN
    return lo >= TSAR_DIV_GUARD_LO && hi <= TSAR_DIV_GUARD_HI;
}
DEVFN void div_pair_rcp_exact(float X, float Y, float Z, float& u, float& v) {    // operands inside the guard
    const float r0 = __builtin_amdgcn_rcpf(Z);
    const float r = fma_(fma_(-Z, r0, 1.0f), r0, r0);
    const float qu = X * r, qv = Y * r;
    u = fma_(fma_(-qu, Z, X), r, qu);
    v = fma_(fma_(-qv, Z, Y), r, qv);
}
// GUARD = false: the caller has shown that every tap of the wave is inside the guard (the clamp-free tap loops: all four window
// corners project inside the source image with Z in range, see view_cost)
template <bool GUARD = true>
DEVFN void persp_divide_exact(float X, float Y, float Z, float& u, float& v) {
    div_pair_rcp_exact(X, Y, Z, u, v);
    if (GUARD) {
        if (__builtin_expect(__any(!div_guard_ok(X, Y, Z)), 0)) {
            u = X / Z;
            v = Y / Z;
        }
    }
}

// sqrtf of the matching cost's tail (pmCost gipuma.cu:289-297: 1 - covar / sqrtf(var_ref * var_src)), correctly rounded like the oracle's
// sqrtf but without the compiler's sequence (v_sqrt_f32 + two residual tests + denormal scaling: ~17 instructions, six of them
// v_cndmask): s = x y and h = y / 2 from y = v_rsq_f32(x), one fused residual correction s' = s + (x - s s) h — Markstein's square
// root.  That s' is the correctly rounded root is not argued but ENUMERATED, like the divide's: for x in two adjacent binades (every
// mantissa, both exponent parities: 2^24 inputs) and sampled across the range, tsar_selftest_sqrt compares it with sqrtf on the device
// (0 mismatches; tests/test_gpu_divide.py, and tsar_set_views re-runs the enumeration once per context).  Operand range [2^-100,
// 2^100] (no intermediate leaves the normal range); the 8-bit tap loops need no guard: both variances are >= 1e-5 where this is
// reached and at most 255^2.  (Measured: +0.1 % on the bench, like the mask-free bookkeeping of multiview_cost +0.4 %; a guarded
// form with the short exact division on top LOST 1.1 % to its two wave-uniform branches: profiles/r04/README.md section 5.)
DEVFN float sqrt_rsq_exact(float x) {                             // x inside [2^-100, 2^100]
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    return fma_(fma_(-s, s, x), h, s);
}

// Philox4x32-10 counter-based generator: stateless (0 B/pixel; the reference keeps 48 B/pixel of
// XORWOW state and re-seeds it from clock64() in every kernel, gipuma.cu:700,1077,1714).
struct Rand4 {
    float u[4];
};
DEVFN Rand4 philox_uniform4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1) {
    uint32_t c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Rand4 o;
    const float s = 5.9604644775390625e-8f;  // 2^-24 -> (0, 1] like curand_uniform
    o.u[0] = (float)((c0 >> 8) + 1u) * s;
    o.u[1] = (float)((c1 >> 8) + 1u) * s;
    o.u[2] = (float)((c2 >> 8) + 1u) * s;
    o.u[3] = (float)((c3 >> 8) + 1u) * s;
    return o;
}
DEVFN float between(float u, float lo, float hi) { return fma_(u, hi - lo, lo); }  // curand_between gipuma.cu:113-116

// getD_cu gipuma.cu:71-86: plane offset d such that the plane with normal n passes through the
// point of pixel (x, y) at `depth`.
DEVFN float plane_offset(const DevRef& rf, const float* n, int x, int y, float depth) {
    float pt[3], X[3];
    pt[0] = depth * (float)x - rf.P34[0];
    pt[1] = depth * (float)y - rf.P34[1];
    pt[2] = depth - rf.P34[2];
    mat3vec(rf.Minv, pt, X);
    return -dot3(n, X);
}
// getDisparity_cu / getDepthFromPlane3_cu gipuma.cu:436-453 (the value is a depth, the reference's
// naming notwithstanding)
DEVFN float plane_depth(const DevRef& rf, float nx, float ny, float nz, float d, int x, int y) {
    if (d != d) return 1000.0f;
    float den = fma_(nz, rf.fx, fma_(ny * ((float)y - rf.K[5]), rf.alpha, nx * ((float)x - rf.K[2])));
    return (-d * rf.fx) / den;
}
DEVFN float plane_depth(const DevRef& rf, const float4& n4, int x, int y) { return plane_depth(rf, n4.x, n4.y, n4.z, n4.w, x, y); }
// getViewVector_cu gipuma.cu:97-105
DEVFN void view_vector(const DevRef& rf, int x, int y, float* v) {
    float pt[3], X[3];
    pt[0] = (float)x - rf.P34[0];
    pt[1] = (float)y - rf.P34[1];
    pt[2] = 1.0f - rf.P34[2];
    mat3vec(rf.Minv, pt, X);
    X[0] -= rf.C[0];
    X[1] -= rf.C[1];
    X[2] -= rf.C[2];
    float inv = 1.0f / sqrtf(dot3(X, X));
    v[0] = X[0] * inv;
    v[1] = X[1] * inv;
    v[2] = X[2] * inv;
}
// getHomography_cu gipuma.cu:207-224: H = K_src (R - t n^T / d) K_ref^-1
// k_sparse (wave-uniform, DevScene): K_src = (fx 0 cx; 0 fy cy; 0 0 1) and K_ref^-1 of the same pattern.  Inside the operand guard
// every element of M and T is finite, so a product with one of those zeros is (+-)0 and adding it returns the other addend: the two
// 3x3 products of matmul_cu (config.h:205-230) then need 15 + 12 operations instead of 27 + 27, with the oracle's values — up to
// the SIGN of an element that is exactly zero (R[e] equal to its quotient to the last bit), which no later step can tell apart: a
// zero H element only ever meets finite factors, and tap fractions / indices come out of floor and x - floor(x).
DEVFN void plane_homography(const DevRef& rf, const DevView& vw, const float4& n4, float* H, bool k_sparse = false) {
    const float n[3] = {n4.x, n4.y, n4.z};
    float M[9], T[9];
    // outer product, then every element divided by d (matdivide, config.h:139-148): nine correctly rounded quotients over one
    // denominator — one reciprocal + Newton step, then three operations per quotient (div_pair_rcp_exact above, same enumeration);
    // the wave takes the IEEE divisions when an operand of any lane is outside the guard (a zero component of t or n, mostly)
    float P[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) P[r * 3 + c] = vw.t[r] * n[c];
    const float d = n4.w;
    // Guard over the nine |P[e]| and |d|.  Rounding is monotonic, so min |fl(t[r] n[c])| = fl(min|t| min|n|) and likewise the
    // maximum: two products bound all nine exactly (the view's min / max |t[r]| come with the view, the plane's are two
    // three-operand instructions that do not depend on the view) — the same decision as nine v_min + nine v_max, for six instructions
    const float nlo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(n[0]), __builtin_fabsf(n[1])), __builtin_fabsf(n[2]));
    const float nhi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n[0]), __builtin_fabsf(n[1])), __builtin_fabsf(n[2]));
    const float lo = __builtin_fminf(vw.t_abs_lo * nlo, __builtin_fabsf(d)), hi = __builtin_fmaxf(vw.t_abs_hi * nhi, __builtin_fabsf(d));
    if (__builtin_expect(__any(!(lo >= TSAR_DIV_GUARD_LO && hi <= TSAR_DIV_GUARD_HI)), 0)) {
#pragma unroll
        for (int e = 0; e < 9; e++) M[e] = vw.R[e] - P[e] / d;
        mat3mul(M, rf.Kinv, T);
        mat3mul(vw.K, T, H);
        return;
    }
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float rc = fma_(fma_(-d, r0, 1.0f), r0, r0);
#pragma unroll
    for (int e = 0; e < 9; e++) {
        const float q = P[e] * rc;
        M[e] = vw.R[e] - fma_(fma_(-q, d, P[e]), rc, q);
    }
    if (k_sparse) {
        const float* Ki = rf.Kinv;
        const float* K = vw.K;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            T[r * 3 + 0] = M[r * 3] * Ki[0];                                                  // + M1 * 0 + M2 * 0
            T[r * 3 + 1] = M[r * 3 + 1] * Ki[4];                                              // M0 * 0 + . + M2 * 0
            T[r * 3 + 2] = fma_(M[r * 3 + 2], Ki[8], fma_(M[r * 3 + 1], Ki[5], M[r * 3] * Ki[2]));
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            H[c] = fma_(K[2], T[6 + c], K[0] * T[c]);                                         // K01 = 0
            H[3 + c] = fma_(K[5], T[6 + c], K[4] * T[3 + c]);                                 // K10 = 0: fma(fy, T1c, 0) = fy T1c
            H[6 + c] = T[6 + c];                                                              // (0 0 1)
        }
        return;
    }
    mat3mul(M, rf.Kinv, T);
    mat3mul(vw.K, T, H);
}

// Fast mode: the same homography as A - b m^T with A = K R K_ref^-1 and b = K t folded per view on the host
// (tsar_api.hip derive_cameras) and m = K_ref^-T n / d, which depends on the plane only, so the compiler hoists it
// out of the view loop: 9 FMAs per view instead of two 3x3 products.  Rounding differs from plane_homography.
DEVFN void plane_homography_fast(const DevRef& rf, const DevView& vw, const float4& n4, float* H) {
    const float inv_d = __builtin_amdgcn_rcpf(n4.w);
    float m[3];
#pragma unroll
    for (int c = 0; c < 3; c++) m[c] = fma_(n4.z, rf.Kinv[6 + c], fma_(n4.y, rf.Kinv[3 + c], n4.x * rf.Kinv[c])) * inv_d;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) H[r * 3 + c] = fma_(-vw.b[r], m[c], vw.A[r * 3 + c]);
}
