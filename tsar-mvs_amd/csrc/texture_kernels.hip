// texture_kernels.hip — weak-texture region detection on the GPU: the CPU/OpenCV
// routine texture() of the reference (main.cpp:365-596, roberts :214-240, Connect :242-362) that produces
// the per-pixel region id (lines->canny) and the per-region flags (cannylines->text / size) the TSAR
// refinement kernels consume.
//
//   8-bit gray -> pyrDown x2 (5x5 [1 4 6 4 1]/16 per axis, REFLECT_101, (sum+128)>>8) -> Roberts cross
//   magnitude with the reference's uchar wrap -> threshold 4 -> border fix -> 4-connected components of
//   the flat pixels (union-find with atomicMin; root = first pixel in raster order) -> labels numbered in
//   raster order of the roots (device exclusive scan) -> per-label count / centroid / bounding box with
//   integer atomics (exact, order-independent) -> "true weak" classification -> labels at full resolution.
//
// Deviations: the HoughLinesP boundary closing (main.cpp:391-435) is OpenCV-internal and
// randomised and is not reproduced; components are the true 4-connected ones, whereas Connect()'s parent
// overwrite can lose a link in rare shapes.
#include <chrono>
#include <rocprim/device/device_scan.hpp>

#include "tsar_dev.h"

#define TX_BLOCK 256

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__global__ void tx_to_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int n) {
    const int p = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (p < n) out[p] = (uint8_t)img[p];
}

__global__ void tx_pyrdown_kernel(const uint8_t* __restrict__ src, int w, int h, uint8_t* __restrict__ dst) {
    const int dw = w / 2, dh = h / 2;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= dw || y >= dh) return;
    const int k[5] = {1, 4, 6, 4, 1};
    int sum = 0;
#pragma unroll
    for (int j = -2; j <= 2; j++) {
        const int sy = reflect101(2 * y + j, h);
        int row = 0;
#pragma unroll
        for (int i = -2; i <= 2; i++) row += k[i + 2] * src[(size_t)sy * w + reflect101(2 * x + i, w)];
        sum += k[j + 2] * row;
    }
    dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
}

__global__ void tx_roberts_kernel(const uint8_t* __restrict__ src, int w, int h, uint8_t* __restrict__ dst) {
    const int j = blockIdx.x * 32 + (threadIdx.x & 31), i = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (j >= w || i >= h) return;
    int t1, t2;
    if (i > 0 && i < h - 1 && j > 0 && j < w - 1) {
        const int a = (int)src[(size_t)i * w + j] - (int)src[(size_t)(i + 1) * w + j + 1];
        const int b = (int)src[(size_t)(i + 1) * w + j] - (int)src[(size_t)i * w + j + 1];
        t1 = a * a; t2 = b * b;
    } else {
        t1 = 100 * 50; t2 = t1;
    }
    const uint8_t mag = (uint8_t)(int)sqrt((double)(t1 + t2));   // (uchar)sqrt(..): wraps for magnitudes >= 256 (main.cpp:235)
    dst[(size_t)i * w + j] = mag > 4 ? 255 : 0;                  // cv::threshold(.., Robthr = 4, 255, THRESH_BINARY) :383
}

// main.cpp:441-446 then :447-452 (two launches keep the reference's order of the two loops)
__global__ void tx_border_rows_kernel(uint8_t* img, int w, int h) {
    const int y = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (y >= h) return;
    if (img[(size_t)y * w + 1] == 0) img[(size_t)y * w] = 0;
    if (img[(size_t)y * w + w - 2] == 0) img[(size_t)y * w + w - 1] = 0;
}
__global__ void tx_border_cols_kernel(uint8_t* img, int w, int h) {
    const int x = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (x >= w) return;
    if (img[(size_t)w + x] == 0) img[x] = 0;
    if (img[(size_t)(h - 2) * w + x] == 0) img[(size_t)(h - 1) * w + x] = 0;
}

// ---- connected components: lock-free union-find, parent links always point to a smaller index ---------
__device__ __forceinline__ int uf_find(const int* parent, int i) {
    int p = parent[i];
    while (p != i) { i = p; p = parent[i]; }
    return i;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    bool done = false;
    while (!done) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }     // a > b: hang a under b
        const int old = atomicMin(&parent[a], b);
        done = (old == a);
        a = old;                                           // somebody re-parented a meanwhile: merge that root with b
    }
}
__global__ void tx_ccl_init_kernel(const uint8_t* __restrict__ img, int* __restrict__ parent, int n) {
    const int p = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (p < n) parent[p] = p;
}
__global__ void tx_ccl_merge_kernel(const uint8_t* __restrict__ img, int* parent, int w, int h) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const int p = y * w + x;
    if (img[p] != 0) return;
    if (x > 0 && img[p - 1] == 0) uf_union(parent, p, p - 1);
    if (y > 0 && img[p - w] == 0) uf_union(parent, p, p - w);
}
__global__ void tx_ccl_flatten_kernel(const uint8_t* __restrict__ img, int* parent, int* __restrict__ is_root, int n) {
    const int p = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (p >= n) return;
    int r = -1;
    if (img[p] == 0) { r = uf_find(parent, p); }
    parent[p] = r;                                         // concurrent readers still reach the same root: r <= old parent chain
    is_root[p] = (r == p) ? 1 : 0;
}
// ---- boundary closing of large regions (main.cpp:385-435) ------------------------------------------------------
// The reference closes gaps in long straight region boundaries with cv::HoughLinesP + cv::line before the final
// labelling.  HoughLinesP is randomised and OpenCV-internal; this is a deterministic Hough transform with the same
// parameters (rho 1, theta 1 deg, threshold 110, minLineLength 160, maxLineGap 18; main.cpp:60-62,425), defined in
// oracle/tsar_oracle_texture.c orc_hough_close and reproduced here operation for operation.  Parity unpinned.
#define TX_HOUGH_THR 110
#define TX_HOUGH_MINLEN 160
#define TX_HOUGH_MAXGAP 18
#define TX_WEAK_COUNT 5000
#define TX_MAX_WEAK 1024

// counts[key] += 1 for every lane with valid set, one atomic per distinct key of the wave: neighbouring pixels mostly share their
// component, and a few large components would otherwise serialise ~10^6 atomics on a handful of addresses (milliseconds)
static __device__ __forceinline__ void wave_count_add(int* __restrict__ counts, int key, bool valid) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);                  // wave-uniform loop over a mask that shrinks every trip
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const int lead = __shfl(key, first);
        const unsigned long long same = __ballot(valid && key == lead);
        if (lane == first) atomicAdd(&counts[lead], __popcll(same));
        todo &= ~same;
    }
}
__global__ void tx_root_count_kernel(const int* __restrict__ root, int* __restrict__ cnt, int n) {
    const int p = blockIdx.x * TX_BLOCK + threadIdx.x;
    const int r = p < n ? root[p] : -1;
    wave_count_add(cnt, r, r >= 0);
}
__global__ void tx_weak_roots_kernel(const int* __restrict__ cnt, int n, int* __restrict__ list, int* __restrict__ nlist) {
    const int p = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (p < n && cnt[p] > TX_WEAK_COUNT) {
        const int k = atomicAdd(nlist, 1);
        if (k < TX_MAX_WEAK) list[k] = p;
    }
}
// boundary(L): pixels not in component L with a 4-neighbour in it (main.cpp:393-421); every boundary pixel votes
__global__ void tx_hough_vote_kernel(const int* __restrict__ root, int L, int w, int h, const float* __restrict__ cs, const float* __restrict__ sn,
                                     uint8_t* __restrict__ bmask, int* __restrict__ acc, int nrho, int rmax) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const int p = y * w + x;
    if (root[p] == L) return;
    if (!((x > 0 && root[p - 1] == L) || (x < w - 1 && root[p + 1] == L) || (y > 0 && root[p - w] == L) || (y < h - 1 && root[p + w] == L))) return;
    bmask[p] = 255;
    for (int t = 0; t < 180; t++) {
        const int r = (int)lrintf((float)x * cs[t] + (float)y * sn[t]);
        atomicAdd(&acc[(size_t)t * nrho + r + rmax], 1);
    }
}
__device__ void tx_draw_line8(uint8_t* img, int w, int h, int x0, int y0, int x1, int y1) {
    const int dx = abs(x1 - x0), sx = x0 < x1 ? 1 : -1;
    const int dy = -abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
    int err = dx + dy;
    for (;;) {
        if (x0 >= 0 && x0 < w && y0 >= 0 && y0 < h) img[(size_t)y0 * w + x0] = 255;
        if (x0 == x1 && y0 == y1) break;
        const int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x0 += sx; }
        if (e2 <= dx) { err += dx; y0 += sy; }
    }
}
// one thread per accumulator cell: local maxima above the threshold are walked across the image and their long
// runs of boundary pixels drawn into the edge image (idempotent writes of 255)
__global__ void tx_hough_segments_kernel(const int* __restrict__ acc, int nrho, int rmax, const float* __restrict__ cs, const float* __restrict__ sn,
                                         const uint8_t* __restrict__ bmask, uint8_t* edge, int w, int h) {
    const int r = blockIdx.x * TX_BLOCK + threadIdx.x, t = blockIdx.y;
    if (r >= nrho) return;
    const int v = acc[(size_t)t * nrho + r];
    if (v < TX_HOUGH_THR) return;
    const int left = r > 0 ? acc[(size_t)t * nrho + r - 1] : 0, right = r < nrho - 1 ? acc[(size_t)t * nrho + r + 1] : 0;
    const int up = t > 0 ? acc[(size_t)(t - 1) * nrho + r] : 0, down = t < 179 ? acc[(size_t)(t + 1) * nrho + r] : 0;
    if (!(v > left && v >= right && v > up && v >= down)) return;
    const float c = cs[t], s = sn[t], rho = (float)(r - rmax);
    const bool xmajor = fabsf(s) >= fabsf(c);
    const int n = xmajor ? w : h;
    int run = 0, sx = 0, sy = 0, lx = 0, ly = 0, gap = 0;
    for (int k = 0; k <= n; k++) {
        bool on = false;
        int x = 0, y = 0;
        if (k < n) {
            if (xmajor) { x = k; y = (int)lrintf((rho - (float)x * c) / s); }
            else        { y = k; x = (int)lrintf((rho - (float)y * s) / c); }
            on = x >= 0 && x < w && y >= 0 && y < h && bmask[(size_t)y * w + x] != 0;
        }
        if (on) {
            if (!run) { run = 1; sx = x; sy = y; }
            lx = x; ly = y; gap = 0;
        } else if (run && (++gap > TX_HOUGH_MAXGAP || k == n)) {
            if (abs(lx - sx) >= TX_HOUGH_MINLEN || abs(ly - sy) >= TX_HOUGH_MINLEN) tx_draw_line8(edge, w, h, sx, sy, lx, ly);
            run = 0; gap = 0;
        }
    }
}

// label = 1 + number of roots before this pixel's root in raster order; statistics with integer atomics
__global__ void tx_label_stats_kernel(const int* __restrict__ root, const int* __restrict__ root_rank, int32_t* __restrict__ lab, int w, int h,
                                      int* __restrict__ count, int* __restrict__ sumx, int* __restrict__ sumy, int* __restrict__ xmin,
                                      int* __restrict__ xmax, int* __restrict__ ymin, int* __restrict__ ymax) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    const bool valid = x < w && y < h;
    int l = 0;
    if (valid) {
        const int p = y * w + x;
        const int r = root[p];
        l = r < 0 ? 0 : root_rank[r] + 1;
        lab[p] = l;
    }
    // one set of atomics per distinct label of the wave (its 64 pixels are two 32-pixel rows: one or two labels, mostly),
    // the seven statistics reduced across the label's lanes first; integer sums and extrema: the result does not depend on order
    const int lane = threadIdx.x & 63;
    bool pending = valid;
    unsigned long long todo = __ballot(pending);
    while (todo) {                                              // wave-uniform
        const int first = __ffsll((long long)todo) - 1;
        const int lead = __shfl(l, first);
        const bool mine = pending && l == lead;
        int c = mine ? 1 : 0, sx = mine ? x : 0, sy = mine ? y : 0;
        int x0 = mine ? x : INT32_MAX, x1 = mine ? x : INT32_MIN, y0 = mine ? y : INT32_MAX, y1 = mine ? y : INT32_MIN;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_xor(c, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o);
            x0 = min(x0, __shfl_xor(x0, o)); x1 = max(x1, __shfl_xor(x1, o));
            y0 = min(y0, __shfl_xor(y0, o)); y1 = max(y1, __shfl_xor(y1, o));
        }
        if (lane == first) {
            atomicAdd(&count[lead], c);
            atomicAdd(&sumx[lead], sx);
            atomicAdd(&sumy[lead], sy);
            atomicMin(&xmin[lead], x0); atomicMax(&xmax[lead], x1);
            atomicMin(&ymin[lead], y0); atomicMax(&ymax[lead], y1);
        }
        pending = pending && !mine;
        todo = __ballot(pending);
    }
}
__global__ void tx_stats_init_kernel(int* xmin, int* xmax, int* ymin, int* ymax, int n, int w4, int h4) {
    const int i = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (i < n) { xmax[i] = 0; xmin[i] = w4 - 1; ymax[i] = 0; ymin[i] = h4 - 1; }
}
// main.cpp:513-532 + :575-593
__global__ void tx_classify_kernel(const int* __restrict__ count, const int* __restrict__ xmin, const int* __restrict__ xmax, const int* __restrict__ ymin,
                                   const int* __restrict__ ymax, float* __restrict__ text, float* __restrict__ size, int n) {
    const int t = blockIdx.x * TX_BLOCK + threadIdx.x;
    if (t >= n) return;
    float tx = 1.0f, sz = 0.0f;
    if (t > 0 && count[t] > 5000) {                                   // weaktextnum :62
        const int xs = xmax[t] - xmin[t], ys = ymax[t] - ymin[t];
        if (xs * ys < 2 * count[t] || count[t] > 100000) {            // `const int sizerat = 2.5` is 2 (:63, :526)
            tx = -1.0f;
            sz = (float)max(xs, ys);
        }
    }
    text[t] = tx;
    size[t] = sz;
}
// main.cpp:559-568
__global__ void tx_upsample_kernel(const int32_t* __restrict__ lab4, int w4, int h4, int w, int h, int32_t* __restrict__ out) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    int sx = x / 4, sy = y / 4;
    if (sx >= w4) sx--;
    if (sy >= h4) sy--;
    out[(size_t)y * w + x] = lab4[(size_t)sy * w4 + sx];
}

// Detects the weak-texture regions of the reference view and installs them as the context's regions
// (equivalent to tsar_set_regions).  labels_out [h][w] int32 (optional), n_regions_out, and host copies of
// text/size (optional, capacity `cap` entries).
extern "C" int tsar_detect_weak_texture(tsar_ctx* ctx, int32_t* labels_out, int mem, int* n_regions_out, float* text_out, float* size_out, int cap) {
    if (!ctx) return TSAR_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return TSAR_ERR_HIP; }
    if (!ctx->have_views) { ctx->err = "tsar_set_views has not been called"; return TSAR_ERR_STATE; }
    if (!ctx->hscene.use_quad) { ctx->err = "weak-texture detection needs 8-bit imagery"; return TSAR_ERR_INVALID; }
    const int w = ctx->w, h = ctx->h, w2 = w / 2, h2 = h / 2, w4 = w2 / 2, h4 = h2 / 2;
    if (w4 < 3 || h4 < 3) { ctx->err = "image too small for weak-texture detection"; return TSAR_ERR_INVALID; }
    const int n4 = w4 * h4;
    hipStream_t st = ctx->stream;
    const bool trace = ctx->trace_host;     // host-side steps on stderr (diagnostics)
    auto tr0 = std::chrono::steady_clock::now();
    auto TR = [&](const char* what) { if (trace) { hipStreamSynchronize(st); auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[weak_texture] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - tr0).count()); tr0 = n; } };
    ScratchScope scratch(ctx);           // temporaries come out of the context's arena (tsar_dev.h)
    auto dmalloc = [&](size_t bytes) -> void* { return scratch.alloc(bytes); };
    auto done = [&](int rc, const char* msg) { if (msg) ctx->err = msg; hipStreamSynchronize(st); scratch.release(); return rc; };
    uint8_t *g0 = (uint8_t*)dmalloc((size_t)w * h), *g2 = (uint8_t*)dmalloc((size_t)w2 * h2), *g4 = (uint8_t*)dmalloc(n4), *edge = (uint8_t*)dmalloc(n4);
    int *parent = (int*)dmalloc((size_t)n4 * 4), *is_root = (int*)dmalloc((size_t)n4 * 4), *rank = (int*)dmalloc((size_t)n4 * 4);
    int32_t* lab4 = (int32_t*)dmalloc((size_t)n4 * 4);
    if (!g0 || !g2 || !g4 || !edge || !parent || !is_root || !rank || !lab4) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
    const dim3 b(TX_BLOCK);
    auto grid2 = [](int ww, int hh) { return dim3((ww + 31) / 32, (hh + 7) / 8); };
    {
        ScopedKernelTimer tm(ctx, "weak_texture");
        hipLaunchKernelGGL(tx_to_u8_kernel, dim3((w * h + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, ctx->img[0], g0, w * h);
        hipLaunchKernelGGL(tx_pyrdown_kernel, grid2(w2, h2), b, 0, st, g0, w, h, g2);
        hipLaunchKernelGGL(tx_pyrdown_kernel, grid2(w4, h4), b, 0, st, g2, w2, h2, g4);
        hipLaunchKernelGGL(tx_roberts_kernel, grid2(w4, h4), b, 0, st, g4, w4, h4, edge);
    }
    TR("allocs + pyramid + edges");
    if (!(ctx->hscene.flags & TSAR_FLAG_NO_LINE_CLOSING)) {
        // first labelling (before the border fix) -> large components -> close gaps in their straight boundaries
        int* cnt0 = (int*)dmalloc((size_t)n4 * 4);
        int* wlist = (int*)dmalloc((size_t)TX_MAX_WEAK * 4 + 4);
        const int rmax = w4 + h4 + 2, nrho = 2 * rmax + 1;
        int* acc = (int*)dmalloc((size_t)180 * nrho * 4);
        uint8_t* bmask = (uint8_t*)dmalloc(n4);
        float* tabs = (float*)dmalloc(360 * 4);
        if (!cnt0 || !wlist || !acc || !bmask || !tabs) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
        float htab[360];
        for (int t = 0; t < 180; t++) {
            const double a = (double)t * 3.14159265358979323846 / 180.0;
            htab[t] = (float)cos(a);
            htab[180 + t] = (float)sin(a);
        }
        ScopedKernelTimer tm(ctx, "weak_texture_closing");
        hipMemcpyAsync(tabs, htab, sizeof htab, hipMemcpyHostToDevice, st);
        hipMemsetAsync(cnt0, 0, (size_t)n4 * 4, st);
        hipMemsetAsync(wlist, 0, (size_t)TX_MAX_WEAK * 4 + 4, st);
        hipLaunchKernelGGL(tx_ccl_init_kernel, dim3((n4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, edge, parent, n4);
        hipLaunchKernelGGL(tx_ccl_merge_kernel, grid2(w4, h4), b, 0, st, edge, parent, w4, h4);
        hipLaunchKernelGGL(tx_ccl_flatten_kernel, dim3((n4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, edge, parent, is_root, n4);
        hipLaunchKernelGGL(tx_root_count_kernel, dim3((n4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, parent, cnt0, n4);
        hipLaunchKernelGGL(tx_weak_roots_kernel, dim3((n4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, cnt0, n4, wlist + 1, wlist);
        std::vector<int> hl(TX_MAX_WEAK + 1);
        hipMemcpyAsync(hl.data(), wlist, (size_t)TX_MAX_WEAK * 4 + 4, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) return done(TSAR_ERR_HIP, "weak-texture kernels failed");
        TR("first labelling + large components D2H");
        const int nweak = hl[0] < TX_MAX_WEAK ? hl[0] : TX_MAX_WEAK;
        for (int k = 0; k < nweak; k++) {                     // independent of each other: masks come from the first labelling
            hipMemsetAsync(acc, 0, (size_t)180 * nrho * 4, st);
            hipMemsetAsync(bmask, 0, (size_t)n4, st);
            hipLaunchKernelGGL(tx_hough_vote_kernel, grid2(w4, h4), b, 0, st, parent, hl[1 + k], w4, h4, tabs, tabs + 180, bmask, acc, nrho, rmax);
            hipLaunchKernelGGL(tx_hough_segments_kernel, dim3((nrho + TX_BLOCK - 1) / TX_BLOCK, 180), b, 0, st, acc, nrho, rmax, tabs, tabs + 180, bmask, edge, w4, h4);
        }
    }
    TR("line closing");
    {
        ScopedKernelTimer tm(ctx, "weak_texture_label");
        hipLaunchKernelGGL(tx_border_rows_kernel, dim3((h4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, edge, w4, h4);
        hipLaunchKernelGGL(tx_border_cols_kernel, dim3((w4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, edge, w4, h4);
        hipLaunchKernelGGL(tx_ccl_init_kernel, dim3((n4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, edge, parent, n4);
        hipLaunchKernelGGL(tx_ccl_merge_kernel, grid2(w4, h4), b, 0, st, edge, parent, w4, h4);
        hipLaunchKernelGGL(tx_ccl_flatten_kernel, dim3((n4 + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, edge, parent, is_root, n4);
    }
    TR("second labelling");
    size_t tmp_bytes = 0;
    if (rocprim::exclusive_scan(nullptr, tmp_bytes, is_root, rank, 0, (size_t)n4, rocprim::plus<int>(), st) != hipSuccess) return done(TSAR_ERR_HIP, "scan sizing failed");
    void* tmp = dmalloc(tmp_bytes);
    if (!tmp) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
    if (rocprim::exclusive_scan(tmp, tmp_bytes, is_root, rank, 0, (size_t)n4, rocprim::plus<int>(), st) != hipSuccess) return done(TSAR_ERR_HIP, "scan failed");
    int last_rank = 0, last_flag = 0;
    hipMemcpyAsync(&last_rank, rank + (n4 - 1), 4, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(&last_flag, is_root + (n4 - 1), 4, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) return done(TSAR_ERR_HIP, "weak-texture kernels failed");
    TR("scan + count D2H");
    const int labelnum = last_rank + last_flag + 1;                          // + label 0 (edge pixels)
    int* stats = (int*)dmalloc((size_t)labelnum * 7 * 4);
    if (!stats) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
    int *count = stats, *sumx = stats + labelnum, *sumy = stats + 2 * labelnum, *xmin = stats + 3 * labelnum, *xmax = stats + 4 * labelnum,
        *ymin = stats + 5 * labelnum, *ymax = stats + 6 * labelnum;
    hipMemsetAsync(stats, 0, (size_t)labelnum * 3 * 4, st);
    hipLaunchKernelGGL(tx_stats_init_kernel, dim3((labelnum + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, xmin, xmax, ymin, ymax, labelnum, w4, h4);
    hipLaunchKernelGGL(tx_label_stats_kernel, grid2(w4, h4), b, 0, st, parent, rank, lab4, w4, h4, count, sumx, sumy, xmin, xmax, ymin, ymax);
    TR("label statistics");
    // install as the context's regions
    hipFree(ctx->region_text); hipFree(ctx->region_size); hipFree(ctx->region_n4);
    ctx->region_text = nullptr; ctx->region_size = nullptr; ctx->region_n4 = nullptr;
    if (hipMalloc((void**)&ctx->region_text, (size_t)labelnum * 4) != hipSuccess || hipMalloc((void**)&ctx->region_size, (size_t)labelnum * 4) != hipSuccess ||
        hipMalloc((void**)&ctx->region_n4, (size_t)labelnum * 16) != hipSuccess)
        return done(TSAR_ERR_NOMEM, "hipMalloc failed");
    hipMemsetAsync(ctx->region_n4, 0, (size_t)labelnum * 16, st);
    hipLaunchKernelGGL(tx_classify_kernel, dim3((labelnum + TX_BLOCK - 1) / TX_BLOCK), b, 0, st, count, xmin, xmax, ymin, ymax, ctx->region_text, ctx->region_size, labelnum);
    hipLaunchKernelGGL(tx_upsample_kernel, grid2(w, h), b, 0, st, lab4, w4, h4, w, h, ctx->canny);
    if (hipGetLastError() != hipSuccess) return done(TSAR_ERR_HIP, "weak-texture launch failed");
    ctx->n_regions = labelnum;
    if (labels_out) hipMemcpyAsync(labels_out, ctx->canny, (size_t)w * h * 4, mem == TSAR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st);
    if (text_out) hipMemcpyAsync(text_out, ctx->region_text, (size_t)(labelnum < cap ? labelnum : cap) * 4, hipMemcpyDeviceToHost, st);
    if (size_out) hipMemcpyAsync(size_out, ctx->region_size, (size_t)(labelnum < cap ? labelnum : cap) * 4, hipMemcpyDeviceToHost, st);
    if (n_regions_out) *n_regions_out = labelnum;
    TR("region tables + classify + upsample + outputs");
    return done(hipStreamSynchronize(st) == hipSuccess ? TSAR_OK : TSAR_ERR_HIP, nullptr);
}
