// tools/valu_rate2.hip — per-instruction VALU issue cost on gfx950, relative to v_fma_f32, for exactly the
// instruction kinds the PatchMatch tap body and the Philox generator are made of.
// Each kernel issues 4 independent dependency chains of one instruction via inline asm, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip && ./valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 1024
#define UNR 32

#define KERNEL(NAME, ASM)                                                                              \
    __global__ void k_##NAME(float* out, float seed, int reps) {                                       \
        float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + 1.0f, c = 0.25f + seed, d = seed * 0.125f; \
        float x = seed * 3.f, y = 1.0f - seed;                                                         \
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

KERNEL(fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(mul, "v_mul_f32 %0, %0, %1")
KERNEL(add, "v_add_f32 %0, %0, %1")
KERNEL(sub, "v_sub_f32 %0, %0, %1")
KERNEL(max, "v_max_f32 %0, %0, %1")
KERNEL(med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(fract, "v_fract_f32 %0, %0")
KERNEL(floor, "v_floor_f32 %0, %0")
KERNEL(cvt_flr, "v_cvt_flr_i32_f32 %0, %0")
KERNEL(cvt_ub0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(cvt_ub1, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(and_b32, "v_and_b32 %0, %0, %1")
KERNEL(xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL(lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(mov, "v_mov_b32 %0, %1")
KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(rcp, "v_rcp_f32 %0, %0")
KERNEL(sqrt, "v_sqrt_f32 %0, %0")
KERNEL(exp, "v_exp_f32 %0, %0")
// round 3: could the four byte converts of a tap be folded into mixed-precision FMAs on a half-float texture?
KERNEL(fma_mix_lo, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]")                 // src1 read as f16 from the low half
KERNEL(fma_mix_hi, "v_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]")  // ... from the high half
KERNEL(fma_mix_2h, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]")  // two f16 operands
KERNEL(cvt_f32_f16, "v_cvt_f32_f16 %0, %0")
KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")

typedef void (*kfn)(float*, float, int);
static double base_ns = 0;
static void run(const char* name, kfn k, float* d_out) {
    const int wps = 4, blocks = 256 * wps, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f, REP);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)wps * REP * UNR * 4;
    const double ns = ms * 1e6 / instr_per_simd;
    if (base_ns == 0) base_ns = ns;
    printf("%-14s %.3f ns per wave64 instruction per SIMD = %.2f x v_fma_f32\n", name, ns, ns / base_ns);
}
#define RUN(NAME) run(#NAME, k_##NAME, d_out)
int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    RUN(fma); RUN(fma); RUN(mul); RUN(add); RUN(sub); RUN(max); RUN(med3); RUN(fract); RUN(floor); RUN(cvt_flr); RUN(cvt_ub0); RUN(cvt_ub1);
    RUN(cvt_f32_i32); RUN(cvt_i32_f32); RUN(mad_i24); RUN(mad_u24); RUN(lshl_add); RUN(add_u32); RUN(and_b32); RUN(xor_b32); RUN(lshr); RUN(bfe);
    RUN(mov); RUN(cndmask); RUN(mul_lo); RUN(mul_hi); RUN(rcp); RUN(sqrt); RUN(exp);
    RUN(fma_mix_lo); RUN(fma_mix_hi); RUN(fma_mix_2h); RUN(cvt_f32_f16); RUN(perm);
    return 0;
}
