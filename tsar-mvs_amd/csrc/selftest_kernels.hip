// selftest_kernels.hip — device self-tests behind the C ABI (include/tsar.h "self-tests"): the strict mode's perspective divide
// (tsar_device_math.h persp_divide_exact) against the IEEE division the oracle uses (getCorrespondingPoint_cu gipuma.cu:161-171).
// tools/div_exact.hip is the exhaustive proof over all mantissa pairs; these entry points let the GPU test suite re-check the
// shipped code path — operands drawn like the tap loop's, across the whole guard range, and across all of fp32 — in milliseconds.
#include "tsar_device_math.h"

#define ST_BLOCK 256

__global__ __launch_bounds__(ST_BLOCK) void divide_arrays_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ Z,
                                                                 size_t n, float* __restrict__ u, float* __restrict__ v, int ieee) {
    const size_t i = (size_t)blockIdx.x * ST_BLOCK + threadIdx.x;
    if (i >= n) return;
    float a, b;
    if (ieee == 2) { const float rz = __builtin_amdgcn_rcpf(Z[i]); a = X[i] * rz; b = Y[i] * rz; }      // the fast mode's form (pm_tap_r5.h)
    else if (ieee) { a = X[i] / Z[i]; b = Y[i] / Z[i]; }
    else persp_divide_exact<true>(X[i], Y[i], Z[i], a, b);
    u[i] = a;
    v[i] = b;
}

DEVFN bool same_quotient(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

// mode 0: operands like the tap loop's — Z in [2^-4, 2^4), u in [-64, 8256), v likewise, X = u Z, Y = v Z
// mode 1: random mantissas and signs, exponents uniform over the guard range [2^-20, 2^38)
// mode 2: any bit pattern (denormals, zeros, infinities, NaN, huge and tiny): the guard must catch what the short form cannot do
// guarded = 0 (modes 0 and 1 only): the unguarded form the clamp-free loops run
__global__ __launch_bounds__(ST_BLOCK) void divide_random_kernel(uint32_t per_thread, uint32_t seed_lo, uint32_t seed_hi, int mode, int guarded,
                                                                 unsigned long long* out) {
    const uint32_t tid = blockIdx.x * ST_BLOCK + threadIdx.x;
    uint32_t bad = 0, fell_back = 0;
    for (uint32_t k = 0; k < per_thread; k++) {
        const Rand4 rn = philox_uniform4(tid, k, 0x5eedu, seed_lo, seed_hi);
        const Rand4 rm = philox_uniform4(tid, k, 0x5eeeu, seed_lo, seed_hi);
        float X, Y, Z;
        if (mode == 0) {
            Z = __builtin_ldexpf(1.0f + rn.u[0], (int)(rn.u[1] * 8.0f) - 4);
            X = between(rn.u[2], -64.0f, 8256.0f) * Z;
            Y = between(rn.u[3], -64.0f, 8256.0f) * Z;
        } else {
            // 24 random bits per uniform: mantissa from one, exponent / sign from another
            const uint32_t b0 = (uint32_t)(rn.u[0] * 16777216.0f) - 1u, b1 = (uint32_t)(rn.u[1] * 16777216.0f) - 1u, b2 = (uint32_t)(rn.u[2] * 16777216.0f) - 1u;
            const uint32_t e0 = (uint32_t)(rm.u[0] * 16777216.0f) - 1u, e1 = (uint32_t)(rm.u[1] * 16777216.0f) - 1u, e2 = (uint32_t)(rm.u[2] * 16777216.0f) - 1u;
            auto make = [&](uint32_t m, uint32_t e) {
                const uint32_t ex = mode == 1 ? 107u + (e >> 1) % 58u : (e >> 1) & 0xffu;     // 2^-20 .. 2^37, or anything
                return __uint_as_float(((e & 1u) << 31) | (ex << 23) | (m & 0x7fffffu));
            };
            X = make(b0, e0); Y = make(b1, e1); Z = make(b2, e2);
        }
        float a, b;
        if (guarded) persp_divide_exact<true>(X, Y, Z, a, b);
        else div_pair_rcp_exact(X, Y, Z, a, b);
        bad += !same_quotient(a, X / Z);
        bad += !same_quotient(b, Y / Z);
        fell_back += !div_guard_ok(X, Y, Z);
    }
    for (int o = 32; o; o >>= 1) { bad += __shfl_down(bad, o); fell_back += __shfl_down(fell_back, o); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], (unsigned long long)bad);
        atomicAdd(&out[1], (unsigned long long)fell_back);
    }
}

// sqrt_rsq_exact against sqrtf.  mode 0: EVERY mantissa in the binades [1, 2) and [2, 4) — both exponent parities, 2^24 inputs: the
// rounding of Markstein's correction depends on the mantissa and the parity only while every intermediate stays normal; mode 1:
// 2^24 random inputs with exponents across the guard range [2^-100, 2^100] (the scaling argument, sampled); mode 3: 2^24 random
// mantissas spread evenly over the 67 binades the production operands can fall into (what the per-context probe adds to mode 0).
__global__ __launch_bounds__(ST_BLOCK) void sqrt_enumerate_kernel(int mode, uint32_t seed_lo, uint32_t seed_hi, unsigned long long* out) {
    const uint32_t i = blockIdx.x * ST_BLOCK + threadIdx.x;            // 2^24 threads
    float x;
    if (mode == 0 || mode == 2) x = __uint_as_float(0x3f800000u + i);   // 1.0 .. 4.0 - ulp
    else if (mode == 3) {
        // the PRODUCTION operand range: var_ref * var_src with both variances in [1e-5, 255^2], i.e. [1e-10, 4.3e9] — exponents
        // 2^-34 .. 2^32 (67 binades), every one of them with 2^24 / 67 random mantissas
        const Rand4 rn = philox_uniform4(i, 0u, 0x5a18u, seed_lo, seed_hi);
        const uint32_t m = (uint32_t)(rn.u[0] * 16777216.0f) - 1u, e = 93u + i % 67u;
        x = __uint_as_float((e << 23) | (m & 0x7fffffu));
    } else {
        const Rand4 rn = philox_uniform4(i, 0u, 0x5a17u, seed_lo, seed_hi);
        const uint32_t m = (uint32_t)(rn.u[0] * 16777216.0f) - 1u, e = 27u + (uint32_t)(rn.u[1] * 200.0f) % 200u;   // 2^-100 .. 2^99
        x = __uint_as_float((e << 23) | (m & 0x7fffffu));
    }
    // mode 2: the control — x v_rsq_f32(x) WITHOUT the correction must differ somewhere, or the comparison discriminates nothing
    const float got = mode == 2 ? x * __builtin_amdgcn_rsqf(x) : sqrt_rsq_exact(x);
    uint32_t bad = __float_as_uint(got) != __float_as_uint(sqrtf(x));
    for (int o = 32; o; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&out[0], (unsigned long long)bad);
}
int launch_selftest_sqrt(tsar_ctx* ctx, int mode, uint64_t seed, unsigned long long* dcounts) {
    hipLaunchKernelGGL(sqrt_enumerate_kernel, dim3((1u << 24) / ST_BLOCK), dim3(ST_BLOCK), 0, ctx->stream, mode, (uint32_t)seed, (uint32_t)(seed >> 32), dcounts);
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

int launch_selftest_divide(tsar_ctx* ctx, const float* X, const float* Y, const float* Z, size_t n, float* u, float* v, int ieee) {
    hipLaunchKernelGGL(divide_arrays_kernel, dim3((unsigned)((n + ST_BLOCK - 1) / ST_BLOCK)), dim3(ST_BLOCK), 0, ctx->stream, X, Y, Z, n, u, v, ieee);
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

int launch_selftest_divide_random(tsar_ctx* ctx, int log2_pairs, uint64_t seed, int mode, int guarded, unsigned long long* dcounts) {
    // 2^log2_pairs triples (two quotients each) over 2^20 threads at most
    const int lt = log2_pairs < 20 ? log2_pairs : 20;
    const uint32_t threads = 1u << lt, per_thread = 1u << (log2_pairs - lt);
    hipLaunchKernelGGL(divide_random_kernel, dim3((threads + ST_BLOCK - 1) / ST_BLOCK), dim3(threads < ST_BLOCK ? threads : ST_BLOCK), 0, ctx->stream, per_thread,
                       (uint32_t)seed, (uint32_t)(seed >> 32), mode, guarded, dcounts);
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

// ---- census of the sweep's hypothesis loop -----------------------------------------------------------------------------------
// pm_sweep_kernel runs ONE rolled loop per wave over the 8 propagation arms and the R refinement steps: an arm costs a whole
// multi-view evaluation for the wave whenever ANY of its 64 lanes has a candidate that survives the early-outs (gipuma.cu:553-555,
// pm_sweep_impl.h: arm skipped, plane already held, depth out of range).  A lane-local queue (each lane walks its own surviving
// candidates) would cost max-over-lanes(survivors) evaluations instead.  This kernel counts both on the context's current state with
// the sweep's own lane -> pixel map and candidate selection, so the gain can be known before the loop is rewritten:
//   out[0] waves with at least one pixel        out[1] sum over waves of arms with any surviving lane   (what the loop runs now)
//   out[2] sum over waves of max-over-lanes survivors            out[3] the same with duplicate planes among a lane's own arms removed
//   out[4] sum over lanes of survivors          out[5] the same without duplicates          out[6] pixels        out[7] arms not skipped (ci >= 0)
#include "pm_sweep_impl.h"
__global__ __launch_bounds__(256) void sweep_census_kernel(const DevScene* __restrict__ sc, int colour, const float* __restrict__ c, const float4* __restrict__ n4,
                                                           int tiles_x, int cost_consistent, unsigned long long* out) {
    const int tix = blockIdx.x % tiles_x, tiy = blockIdx.x / tiles_x;
    const int ly = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int y = tiy * 16 + ly;
    const int x = tix * 32 + 2 * k + ((colour + y) & 1);
    const int w = sc->w, h = sc->h;
    const bool live = x < w && y < h;
    uint32_t alive = 0, alive_nodup = 0, present = 0;
    if (live) {
        const DevRef& rf = sc->ref;
        int cand[8];
        select_candidates(sc, c, c, x, y, cand);
        const float4 n_first = n4[y * w + x];
        float4 pl[8];
#pragma unroll
        for (int a = 0; a < 8; a++) {
            if (cand[a] < 0) continue;
            present |= 1u << a;
            pl[a] = n4[cand[a] & 0x3fffffff];
            if (cost_consistent && same_bits(pl[a], n_first)) continue;
            const float d = plane_depth(rf, pl[a], x, y);
            if (!(d >= rf.depthMin && d <= rf.depthMax)) continue;
            alive |= 1u << a;
            bool dup = false;
#pragma unroll
            for (int b = 0; b < 8; b++)
                if (b < a && ((alive_nodup >> b) & 1u) && same_bits(pl[a], pl[b])) dup = true;
            if (!dup) alive_nodup |= 1u << a;
        }
    }
    // per wave
    int arms_any = 0;
#pragma unroll
    for (int a = 0; a < 8; a++) arms_any += __any((alive >> a) & 1u) ? 1 : 0;
    int mx = __popc(alive), mxn = __popc(alive_nodup), sm = mx, smn = mxn, px = live ? 1 : 0, pr = __popc(present);
    for (int o = 32; o; o >>= 1) {
        mx = max(mx, __shfl_xor(mx, o)); mxn = max(mxn, __shfl_xor(mxn, o));
        sm += __shfl_xor(sm, o); smn += __shfl_xor(smn, o); px += __shfl_xor(px, o); pr += __shfl_xor(pr, o);
    }
    if ((threadIdx.x & 63) == 0 && px > 0) {
        atomicAdd(&out[0], 1ull); atomicAdd(&out[1], (unsigned long long)arms_any); atomicAdd(&out[2], (unsigned long long)mx);
        atomicAdd(&out[3], (unsigned long long)mxn); atomicAdd(&out[4], (unsigned long long)sm); atomicAdd(&out[5], (unsigned long long)smn);
        atomicAdd(&out[6], (unsigned long long)px); atomicAdd(&out[7], (unsigned long long)pr);
    }
}

// Census behind the propagation memo (pm_sweep_impl.h SweepMemo): how often does an arm's candidate carry a plane this pixel already
// tried in its previous propagation launch?  A plane rejected once stays rejected (the cost of a (pixel, plane) pair is fixed and the
// pixel's cost never rises), so such a hypothesis can be skipped bit-exactly.  memo_dev: 8 hashes per pixel, written for the next call.
//   out[0] alive arms   out[1] alive arms repeating the same arm's previous plane   out[2] ... any previous arm's plane
//   out[3] (wave, arm) pairs with an alive lane   out[4] ... in which EVERY alive lane repeats (what the rolled loop could skip)
//   out[5] sum over waves of max-over-lanes alive arms   out[6] ... of max-over-lanes non-repeating alive arms (lane-local queues)
//   out[7] sum over waves of ceil(non-repeating alive arms of the wave / 64): the trips of the packed form
__global__ __launch_bounds__(256) void sweep_repeat_kernel(const DevScene* __restrict__ sc, int colour, const float* __restrict__ c, const float4* __restrict__ n4,
                                                           int tiles_x, int cost_consistent, unsigned long long* __restrict__ memo, unsigned long long* out) {
    const int tix = blockIdx.x % tiles_x, tiy = blockIdx.x / tiles_x;
    const int ly = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int y = tiy * 16 + ly;
    const int x = tix * 32 + 2 * k + ((colour + y) & 1);
    const int w = sc->w, h = sc->h;
    const bool live = x < w && y < h;
    uint32_t alive = 0, rep_same = 0, rep_any = 0;
    if (live) {
        const DevRef& rf = sc->ref;
        int cand[8];
        select_candidates(sc, c, c, x, y, cand);
        const size_t p = (size_t)y * w + x;
        const float4 n_first = n4[p];
        unsigned long long prev[8], now[8];
#pragma unroll
        for (int a = 0; a < 8; a++) { prev[a] = memo[p * 8 + a]; now[a] = 0; }
#pragma unroll
        for (int a = 0; a < 8; a++) {
            if (cand[a] < 0) continue;
            const float4 pl = n4[cand[a] & 0x3fffffff];
            if (cost_consistent && same_bits(pl, n_first)) continue;
            const float d = plane_depth(rf, pl, x, y);
            if (!(d >= rf.depthMin && d <= rf.depthMax)) continue;
            alive |= 1u << a;
            unsigned long long hsh = ((unsigned long long)__float_as_uint(pl.x) | ((unsigned long long)__float_as_uint(pl.y) << 32)) * 0x9E3779B97F4A7C15ull;
            hsh ^= ((unsigned long long)__float_as_uint(pl.z) | ((unsigned long long)__float_as_uint(pl.w) << 32)) * 0xC2B2AE3D27D4EB4Full;
            hsh |= 1ull;
            now[a] = hsh;
            if (prev[a] == hsh) rep_same |= 1u << a;
#pragma unroll
            for (int b = 0; b < 8; b++)
                if (prev[b] == hsh) rep_any |= 1u << a;
        }
#pragma unroll
        for (int a = 0; a < 8; a++) memo[p * 8 + a] = now[a];
    }
    int pairs = 0, pairs_skippable = 0;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const bool al = (alive >> a) & 1u, fresh = al && !((rep_any >> a) & 1u);
        if (__any(al)) { pairs++; if (!__any(fresh)) pairs_skippable++; }
    }
    int sa = __popc(alive), ss = __popc(rep_same), sy = __popc(rep_any), mx = sa, mf = __popc(alive & ~rep_any), sf = mf;
    for (int o = 32; o; o >>= 1) {
        sa += __shfl_xor(sa, o); ss += __shfl_xor(ss, o); sy += __shfl_xor(sy, o); sf += __shfl_xor(sf, o);
        mx = max(mx, __shfl_xor(mx, o)); mf = max(mf, __shfl_xor(mf, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], (unsigned long long)sa); atomicAdd(&out[1], (unsigned long long)ss); atomicAdd(&out[2], (unsigned long long)sy);
        atomicAdd(&out[3], (unsigned long long)pairs); atomicAdd(&out[4], (unsigned long long)pairs_skippable);
        atomicAdd(&out[5], (unsigned long long)mx); atomicAdd(&out[6], (unsigned long long)mf);
        atomicAdd(&out[7], (unsigned long long)((sf + 63) >> 6));       // trips of the packed form: the wave's fresh pairs, 64 at a time
    }
}
int launch_sweep_repeat(tsar_ctx* ctx, int colour, unsigned long long* memo, unsigned long long* dout) {
    const DevScene& hs = ctx->hscene;
    const int tiles_x = (hs.w + 31) / 32, tiles_y = (hs.h + 15) / 16;
    hipLaunchKernelGGL(sweep_repeat_kernel, dim3(tiles_x * tiles_y), dim3(256), 0, ctx->stream, ctx->dscene, colour, ctx->buf[0].c, ctx->buf[0].n4, tiles_x,
                       ctx->cost_consistent ? 1 : 0, memo, dout);
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}

int launch_sweep_census(tsar_ctx* ctx, int colour, unsigned long long* dout) {
    const DevScene& hs = ctx->hscene;
    const int tiles_x = (hs.w + 31) / 32, tiles_y = (hs.h + 15) / 16;
    hipLaunchKernelGGL(sweep_census_kernel, dim3(tiles_x * tiles_y), dim3(256), 0, ctx->stream, ctx->dscene, colour, ctx->buf[0].c, ctx->buf[0].n4, tiles_x,
                       ctx->cost_consistent ? 1 : 0, dout);
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}
