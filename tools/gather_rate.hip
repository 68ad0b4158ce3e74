// tools/gather_rate.hip — texture-addresser cost of the sweep's gather shapes on gfx950: 64 lanes, 4 rows x 16 lanes, lanes of a
// row `stride` bytes apart, each lane loading 4 / 8 / 16 bytes from an L2-resident image.  Prints ns per wave-instruction per CU.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 4096
// HOT: the origin moves only every 64 iterations, and then by a few texels: every gather hits L1 (the TA issue cost itself)
template <int BYTES, bool HOT>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ img, uint32_t* out, int pitch_dw, int stride_dw, int rows, uint32_t seed) {
    const int lane = threadIdx.x & 63, wv = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const int row = lane >> 4, col = lane & 15;
    uint32_t acc = 0;
    uint32_t pos = (uint32_t)wv * 2654435761u + seed;
    for (int it = 0; it < REP; it++) {
        if (!HOT || (it & 63) == 0) pos = pos * 1664525u + 1013904223u;   // wave-uniform pseudo-random origin
        const int y0 = (int)((pos >> 8) % (uint32_t)(rows - 8)), x0 = (int)((pos >> 20) % (uint32_t)(pitch_dw - 16 * stride_dw - 8));
        const uint32_t* p = img + (size_t)(y0 + row + (HOT ? (it & 3) : 0)) * pitch_dw + x0 + col * stride_dw + (HOT ? ((it >> 2) & 7) : 0);
        if (BYTES == 4) acc += p[0];
        else if (BYTES == 8) { const uint2 v = *(const uint2*)p; acc += v.x ^ v.y; }
        else { uint4 v; __builtin_memcpy(&v, p, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int BYTES, bool HOT>
static void run(const char* name, const uint32_t* img, uint32_t* out, int pitch_dw, int stride_dw, int rows) {
    const int blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BYTES, HOT>), dim3(blocks), dim3(256), 0, 0, img, out, pitch_dw, stride_dw, rows, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BYTES, HOT>), dim3(blocks), dim3(256), 0, 0, img, out, pitch_dw, stride_dw, rows, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)blocks * 4 * REP / 256.0;     // wave-instructions per CU
    printf("%-48s %6.2f ns per 64-lane gather per CU  (%.1f cycles at 2.24 GHz)\n", name, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.24);
}
int main() {
    const int pitch_dw = 6050, rows = 1024;                      // 24.8 MB: L2 + Infinity-Cache resident
    uint32_t *img, *out;
    hipMalloc(&img, (size_t)pitch_dw * rows * 4); hipMalloc(&out, 256 * 4 * 256 * 4);
    hipMemset(img, 1, (size_t)pitch_dw * rows * 4);
    run<4, false>("4 B per lane, lanes 8 B apart, random origin", img, out, pitch_dw, 2, rows);
    run<4, true>("4 B per lane, lanes 8 B apart, L1-hot", img, out, pitch_dw, 2, rows);
    run<4, false>("4 B per lane, lanes 4 B apart, random origin", img, out, pitch_dw, 1, rows);
    run<4, true>("4 B per lane, lanes 4 B apart, L1-hot", img, out, pitch_dw, 1, rows);
    run<8, false>("8 B per lane, lanes 8 B apart, random origin", img, out, pitch_dw, 2, rows);
    run<8, true>("8 B per lane, lanes 8 B apart, L1-hot", img, out, pitch_dw, 2, rows);
    run<16, false>("16 B per lane, lanes 8 B apart, random origin", img, out, pitch_dw, 2, rows);
    run<16, true>("16 B per lane, lanes 8 B apart, L1-hot", img, out, pitch_dw, 2, rows);
    run<16, false>("16 B per lane, lanes 16 B apart, random origin", img, out, pitch_dw, 4, rows);
    run<16, true>("16 B per lane, lanes 16 B apart, L1-hot", img, out, pitch_dw, 4, rows);
    run<8, false>("8 B per lane, lanes 16 B apart, random origin", img, out, pitch_dw, 4, rows);
    run<8, true>("8 B per lane, lanes 16 B apart, L1-hot", img, out, pitch_dw, 4, rows);
    return 0;
}
