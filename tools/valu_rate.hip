// tools/valu_rate.hip — VALU issue-rate microbenchmark used to calibrate the tap-loop cost model
// (DESIGN.md §4): how many cycles a SIMD spends per wave64 instruction for the instruction kinds the
// PatchMatch tap body is made of, with 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 2048
template <int KIND>
__global__ void kern(float* out, float seed, int reps) {
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + 1.0f, c = 0.25f, d = seed * 0.125f;
    v2f p = {a, b}, q = {c, d}, r = {b, a}, s2 = {d, c};
    int ia = (int)(a * 100.f), ib = threadIdx.x + 3;
    for (int it = 0; it < reps; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if (KIND == 0) { a = __builtin_fmaf(a, 0.999f, c); b = __builtin_fmaf(b, 1.001f, d); c = __builtin_fmaf(c, 0.999f, a); d = __builtin_fmaf(d, 1.001f, b); }
            if (KIND == 1) { p = __builtin_elementwise_fma(p, q, r); r = __builtin_elementwise_fma(r, q, s2); s2 = __builtin_elementwise_fma(s2, q, p); q = __builtin_elementwise_fma(q, (v2f){0.999f, 1.001f}, (v2f){1e-3f, 1e-3f}); }
            if (KIND == 2) { a = floorf(a * 1.5f) ; b = fminf(fmaxf(b, a), 7.f); c = floorf(c + b); d = fmaxf(d, c) ; }   // mul, floor, max, min, add, floor, max
            if (KIND == 3) { a = __builtin_amdgcn_rcpf(a + 1.f); b = __builtin_amdgcn_rcpf(b + 1.f); c = __builtin_amdgcn_rcpf(c + 1.f); d = __builtin_amdgcn_rcpf(d + 1.f); }
            if (KIND == 4) { ia = __mul24(ia, 3) + ib; ib = (ib << 2) + ia; ia = ia ^ (ib >> 3); ib = min(ib, ia) + 1; }
            if (KIND == 5) { a = (float)((unsigned)ia & 0xffu); b = (float)(((unsigned)ia >> 8) & 0xffu); ia = (int)(a + b) + ib; c = (float)(((unsigned)ib >> 16) & 0xffu); ib = ib + (int)c; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + p.x + p.y + q.x + r.y + s2.x + (float)ia + (float)ib;
}
template <int KIND>
void run(const char* name, int ops_per_unroll, float* d_out) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps, threads = 256;     // 256 CUs x wps blocks of 4 waves -> wps waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kern<KIND><<<blocks, threads>>>(d_out, 1.0f, 8);
        hipEventRecord(e0);
        kern<KIND><<<blocks, threads>>>(d_out, 1.0f, REP);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)wps * REP * 32 * ops_per_unroll;     // wave-instructions issued on one SIMD
        printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32 (4 indep chains)", 4, d_out);
    run<1>("v_pk_fma_f32 (4 chains)", 4, d_out);
    run<2>("mul/floor/max/min/add mix", 7, d_out);
    run<3>("v_rcp_f32 + add", 8, d_out);
    run<4>("int mul24/shift/xor/min", 8, d_out);
    run<5>("cvt ubyte / cvt int", 9, d_out);
    return 0;
}
