// tools/valu_rate_f64.hip — issue cost of the 64-bit min / max (the compare-exchange of a register-resident sorting network on
// (value, slot) keys packed into doubles: wmf_kernels.hip) against v_fma_f32.   hipcc --offload-arch=gfx950 -O3 -o valu_rate_f64 valu_rate_f64.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 1024
#define UNR 32
#define KERNEL64(NAME, ASM)                                                                            \
    __global__ void k_##NAME(double* out, double seed, int reps) {                                     \
        double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + 1.0, c = 0.25 + seed, d = seed * 0.125; \
        double x = seed * 3., y = 1.0 - seed;                                                          \
        for (int it = 0; it < reps; it++) {                                                            \
            _Pragma("unroll") for (int u = 0; u < UNR; u++) {                                          \
                asm volatile(ASM : "+v"(a) : "v"(x), "v"(y));                                          \
                asm volatile(ASM : "+v"(b) : "v"(x), "v"(y));                                          \
                asm volatile(ASM : "+v"(c) : "v"(x), "v"(y));                                          \
                asm volatile(ASM : "+v"(d) : "v"(x), "v"(y));                                          \
            }                                                                                          \
        }                                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                    \
    }
KERNEL64(min_f64, "v_min_f64 %0, %0, %1")
KERNEL64(max_f64, "v_max_f64 %0, %0, %1")
KERNEL64(fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(add_f64, "v_add_f64 %0, %0, %1")
__global__ void k_fma_f32(double* out, double seed, int reps) {
    float a = (float)seed + threadIdx.x * 1e-3f, b = 1.5f, c = 0.25f, d = 0.125f, x = 3.f, y = 0.5f;
    for (int it = 0; it < reps; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
typedef void (*kfn)(double*, double, int);
static void run(const char* name, kfn k, double* d_out, int wps) {
    const int blocks = 256 * wps, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0, REP);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s waves/SIMD %d: %.3f ns per wave64 instruction per SIMD\n", name, wps, ms * 1e6 / ((double)wps * REP * UNR * 4));
}
int main() {
    double* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(double));
    for (int wps : {1, 2, 4}) { run("fma_f32", k_fma_f32, d_out, wps); run("min_f64", k_min_f64, d_out, wps); run("max_f64", k_max_f64, d_out, wps); run("fma_f64", k_fma_f64, d_out, wps); run("add_f64", k_add_f64, d_out, wps); }
    return 0;
}
