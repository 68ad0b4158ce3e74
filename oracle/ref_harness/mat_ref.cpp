// mat_ref.cpp — C-ABI harness around the REFERENCE's own 3x3 matrix macros (config.h:60-240: a header of #defines over plain
// float arrays; nothing of CUDA is needed to expand them).  TEST INFRASTRUCTURE ONLY (oracle/), built by `make -C oracle ref` in the
// build container from the header where it lies.  ref_homography composes the macros in the order getHomography_cu does
// (gipuma.cu:207-224: outer product, matdivide by d, R minus it, times K1^-1, K2 times that) — with the array form
// `outer_product` where the reference uses `outer_product4` on a float4 (the same nine products; float4 is a CUDA type).
// What this pins: which products are summed in which order and that every element is DIVIDED by d (matdivide), i.e. the
// restatement's mat3mul / plane_homography.  What it cannot pin: where nvcc fuses multiply-adds.
#include "config.h"

extern "C" {

void ref_matmul(const float* m0, const float* m1, float* out) { matmul_cu(m0, m1, out); }
void ref_matvecmul(const float* m, const float* v, float* out) { matvecmul(m, v, out); }
void ref_homography(const float* K1_inv, const float* K2, const float* R, const float* t, const float* n, float d, float* H) {
    float tmp2[9], Rm[9];
    for (int i = 0; i < 9; i++) Rm[i] = R[i];
    outer_product(t, n, H);
    matdivide(H, d);
    matmatsub2(Rm, H);
    matmul_cu(H, K1_inv, tmp2);
    matmul_cu(K2, tmp2, H);
}

}  // extern "C"
