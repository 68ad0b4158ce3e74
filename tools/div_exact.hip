// tools/div_exact.hip — proof run for the strict mode's perspective divide (pm_core.h div_pair_exact).
//
// The oracle divides with IEEE `/` (oracle/tsar_oracle.c, getCorrespondingPoint_cu gipuma.cu:161-171).  hipcc lowers an fp32 `/` to
//   v_div_scale x2, v_rcp_f32, fma e = 1 - d r, fma r' = r + e r, mul q = n r', fma s = n - d q, fma q' = q + s r', fma s' = n - d q',
//   v_div_fmas q'' = q' + s' r', v_div_fixup                                                              (~12 VALU + 1 transcendental)
// Scale and fixup only act on denormal / huge / zero / inf / nan operands.  With all operands and the quotient well inside the normal
// range they are the identity, every fma above is exact in its residual, and scaling both operands by powers of two commutes with
// every step (no underflow, no overflow) — so whether a shorter sequence returns the same bits depends on the two 23-bit MANTISSAS
// alone.  That space has 2^46 points and this GPU walks it in about a minute, so the question "is one correction enough after one
// Newton step on v_rcp_f32?" is answered by exhaustion rather than by argument:
//   A: r' (shared by u and v), q = n r', s = n - d q, q' = q + s r'                 (3 VALU per quotient + 2 + 1 transcendental per tap)
//   B: A + s' = n - d q', q'' = q' + s' r'                                         (the compiler's sequence minus scale / fixup)
//   C: A without the Newton step (q' from the raw v_rcp_f32)
// Usage: div_exact [log2 numerator mantissas per denominator mantissa = 23] [ez = 0] [ex = 0]   (all 2^23 denominators always run)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_exact div_exact.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Counts { unsigned long long bad_a, bad_b, bad_c, pairs; uint32_t first[8][4]; uint32_t n_first; };

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// one launch: every denominator mantissa (thread) against numerator mantissas [mx0, mx0 + n) in steps of `stride`
__global__ __launch_bounds__(256) void walk(Counts* out, int ez, int ex, uint32_t sz, uint32_t sx, uint32_t mx0, uint32_t n, uint32_t stride) {
    const uint32_t mz = blockIdx.x * 256u + threadIdx.x;
    const float Z = __uint_as_float(sz | ((uint32_t)(ez + 127) << 23) | mz);
    const float r0 = __builtin_amdgcn_rcpf(Z);
    const float e = fma_(-Z, r0, 1.0f);
    const float r = fma_(e, r0, r0);
    uint32_t bad_a = 0, bad_b = 0, bad_c = 0;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t mx = (mx0 + k * stride) & 0x7fffffu;
        const float X = __uint_as_float(sx | ((uint32_t)(ex + 127) << 23) | mx);
        const float ref = X / Z;
        const float q = X * r;
        const float qa = fma_(fma_(-q, Z, X), r, q);
        const float qb = fma_(fma_(-qa, Z, X), r, qa);
        const float qc0 = X * r0;
        const float qc = fma_(fma_(-qc0, Z, X), r0, qc0);
        const bool fa = __float_as_uint(qa) != __float_as_uint(ref);
        bad_a += fa;
        bad_b += __float_as_uint(qb) != __float_as_uint(ref);
        bad_c += __float_as_uint(qc) != __float_as_uint(ref);
        if (fa) {
            const uint32_t slot = atomicAdd(&out->n_first, 1u);
            if (slot < 8) { out->first[slot][0] = __float_as_uint(X); out->first[slot][1] = __float_as_uint(Z); out->first[slot][2] = __float_as_uint(qa); out->first[slot][3] = __float_as_uint(ref); }
        }
    }
    // wave totals, one atomic per wave
    for (int o = 32; o; o >>= 1) { bad_a += __shfl_down(bad_a, o); bad_b += __shfl_down(bad_b, o); bad_c += __shfl_down(bad_c, o); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out->bad_a, (unsigned long long)bad_a);
        atomicAdd(&out->bad_b, (unsigned long long)bad_b);
        atomicAdd(&out->bad_c, (unsigned long long)bad_c);
        atomicAdd(&out->pairs, 64ull * n);
    }
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 23;
    const int ez = argc > 2 ? atoi(argv[2]) : 0, ex = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t sz = argc > 4 && atoi(argv[4]) ? 0x80000000u : 0u, sx = argc > 5 && atoi(argv[5]) ? 0x80000000u : 0u;
    if (lg < 8 || lg > 23) { fprintf(stderr, "log2 of numerator mantissas must be 8..23\n"); return 2; }
    Counts* d;
    CHECK(hipMalloc(&d, sizeof(Counts)));
    CHECK(hipMemset(d, 0, sizeof(Counts)));
    const uint32_t total = 1u << lg, stride = (1u << 23) >> lg;   // lg < 23: an evenly spaced subset of the numerator mantissas, offset by 1 per slice so that odd mantissas appear
    const uint32_t per_launch = total < 4096 ? total : 4096;      // ~0.1 s per launch: progress lines, no watchdog risk
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (uint32_t done = 0; done < total; done += per_launch) {
        hipLaunchKernelGGL(walk, dim3((1u << 23) / 256), dim3(256), 0, 0, d, ez, ex, sz, sx, done * stride + (stride > 1 ? (done / per_launch) % stride : 0), per_launch, stride);
        if (((done / per_launch) & 255) == 255) {
            CHECK(hipDeviceSynchronize());
            fprintf(stderr, "  %u / %u numerator mantissas\n", done + per_launch, total);
        }
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    Counts h;
    CHECK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("{\"denominator_mantissas\": %u, \"numerator_mantissas\": %u, \"pairs\": %llu, \"ez\": %d, \"ex\": %d, \"neg_z\": %d, \"neg_x\": %d, "
           "\"mismatch_one_correction\": %llu, \"mismatch_two_corrections\": %llu, \"mismatch_no_newton\": %llu, \"seconds\": %.2f, \"pairs_per_s\": %.3g",
           1u << 23, total, h.pairs, ez, ex, sz != 0, sx != 0, h.bad_a, h.bad_b, h.bad_c, ms / 1e3, h.pairs / (ms / 1e3));
    printf(", \"first_mismatches_x_z_got_want\": [");
    for (uint32_t i = 0; i < (h.n_first < 8 ? h.n_first : 8); i++)
        printf("%s[\"%08x\", \"%08x\", \"%08x\", \"%08x\"]", i ? ", " : "", h.first[i][0], h.first[i][1], h.first[i][2], h.first[i][3]);
    printf("]}\n");
    return 0;
}
