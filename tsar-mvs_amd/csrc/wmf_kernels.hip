// wmf_kernels.hip — multi-scale weighted median filter of the TSAR refinement:
// gipuma_WMF (reliability detection, reference gipuma.cu:1499-1698) and gipuma_WMF_Final
// (fill of unreliable pixels in textured regions, :1294-1497).
//
// The reference bubble-sorts four (value, weight) lists of up to 121 taps per pixel in 5 kB of
// per-thread local memory, O(n^2) data-moving swaps.  Here the same sorted order is obtained without
// moving data: each tap's rank is counted (stable: ties keep tap order) with the list held in VGPRs, and the
// lists are walked in rank order.  The reference's sort also touches slot `num` (a zero entry joins and the largest entry
// drops out); that is reproduced by ranking num+1 entries and walking the first num.
// Neighbours are read from launch-start copies of scale / depth / planes (the reference reads what
// other threads of the same launch are writing).
#include "tsar_device_math.h"

#define WMF_BLOCK 64
#define WMF_SLOTS 122   // 11 x 11 tap slots in enumeration order + the zero slot the reference's sort drags in (slot 121)
#define WMF_REGS 128    // the list being ranked, in registers: four 32-float vectors

// Tap slots are kept in ENUMERATION order (i outer, j inner: slot = 11 ii + jj), invalid ones flagged, instead of being
// compacted: the compacted index of the reference (`num++`) is monotone in the slot number, so stable order and ranks among
// the valid taps are the same — and a slot number is wave-uniform, which is what lets the list being ranked live in VGPRs.
// Only the weights persist per thread (scratch, 488 B: they are read by per-lane rank order in the walks).  The values being
// ranked are re-gathered from the launch-start planes list by list — reads that neighbouring pixels share through L1 / L2 —
// instead of being parked in scratch (five more arrays, ~2.4 KB per thread and ~580 MB of private memory in flight chip-wide,
// which is HBM traffic); the pixel a slot refers to is recomputed from the slot number.
struct WmfTaps {
    float w[WMF_SLOTS];
    uint64_t valid_lo, valid_hi;   // bit t of (hi:lo): slot t holds a tap (slot 121, the zero slot, always does)
    int num;                       // number of valid taps, excluding the zero slot
    int x, y, radius, gap;         // geometry of the tap grid
};
DEVFN bool slot_valid(const WmfTaps& t, int k) { return ((k < 64 ? t.valid_lo >> k : t.valid_hi >> (k - 64)) & 1u) != 0; }
// pixel index of tap slot k (k < 121): slot = 11 ii + jj, offsets (-radius + ii gap, -radius + jj gap)
DEVFN int slot_pixel(const WmfTaps& t, int k, int w) {
    const int ii = k / 11, jj = k - 11 * ii;
    return (t.y - t.radius + jj * t.gap) * w + (t.x - t.radius + ii * t.gap);
}
// the same for a slot number that comes out of the ranking (per-lane): never outside the image, whatever the list held
DEVFN int slot_pixel_safe(const WmfTaps& t, int k, int w, int h) {
    const int ii = k / 11, jj = k - 11 * ii;
    const int px = min(max(t.x - t.radius + ii * t.gap, 0), w - 1), py = min(max(t.y - t.radius + jj * t.gap, 0), h - 1);
    return py * w + px;
}
enum WmfList { WMF_DEPTH = 0, WMF_NX = 1, WMF_NY = 2, WMF_NZ = 3 };
template <int LIST>
DEVFN float slot_value(const float* __restrict__ depth_in, const float4* __restrict__ n_in, int q) {
    if (LIST == WMF_DEPTH) return depth_in[q];
    const float* nn = (const float*)(n_in + q);
    return nn[LIST - 1];
}

// Per-workgroup staging of the sorted order: pos[r] = slot with stable rank r (bytes, [rank][thread]).
struct WmfLds {
    unsigned char pos[WMF_SLOTS * WMF_BLOCK];
};

typedef float f32x32 __attribute__((ext_vector_type(32)));

// Counting without condition masks.  `r += (vj <= vk)` compiles to v_cmp (writes an SGPR pair) + v_addc (reads it): on gfx950
// a VALU-written SGPR needs two wait states before a VALU read (hipcc pads every pair with s_nop 1), and instructions that
// take a lane mask from SGPRs issue far slower than plain VGPR arithmetic (measured: 127 ms per launch with masks, even with
// the eight compares hoisted ahead of the eight adds).  The order of two floats is the sign of their difference — exact in
// IEEE arithmetic, zero only for equal operands; list entries are loaded as v + 0.0f so that -0 cannot appear, invalid slots
// are +inf — so each comparison is a v_sub_f32 whose sign bit is shifted into a 32-bit register (v_alignbit_b32), and one
// v_bcnt_u32_b32 per register and 32 comparisons adds the ones up: VGPR-only, full rate.
//   before-loop (entries that sort first on ties): vj <= vk  <=>  sign(vk - vj) == 0   -> counts zeros
//   after-loop:                                     vj <  vk  <=>  sign(vj - vk) == 1   -> counts ones
struct SignCount8 {
    uint32_t s[8];
    DEVFN void clear() {
#pragma unroll
        for (int c = 0; c < 8; c++) s[c] = 0u;
    }
    // push sign(a[c] - b) (FLIP = false) or sign(b - a[c]) (FLIP = true)
    template <bool FLIP>
    DEVFN void push(const float (&a)[8], float b) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float d = FLIP ? b - a[c] : a[c] - b;
            s[c] = __builtin_amdgcn_alignbit(s[c], __float_as_uint(d), 31);      // (s << 1) | sign(d)
        }
    }
    DEVFN void add_ones(int (&r)[8]) const {
#pragma unroll
        for (int c = 0; c < 8; c++) r[c] += __builtin_popcount(s[c]);
    }
    DEVFN void add_zeros(int (&r)[8], int pushed) const {
#pragma unroll
        for (int c = 0; c < 8; c++) r[c] += pushed - __builtin_popcount(s[c]);
    }
};

// pos[r] = slot of the entry with stable rank r among the valid slots (the zero slot included).
// The reference bubble-sorts each list in 5 kB of per-thread local memory; the previous version of this kernel counted ranks
// from an LDS copy of the list (39 KB per 64 threads -> one wave per SIMD, LDS-latency bound, 233 ms per launch at 24 Mpixel).
// Here the list sits in 128 VGPRs of its thread — the register file is the largest on-chip memory of a CU, 512 KB — and entry j,
// j wave-uniform, is read with a relative-index move (s_set_gpr_idx_on + v_mov): no LDS or scratch access in the O(n^2) part.
// Eight entries are ranked per pass: one indexed move feeds eight compare / add-carry pairs.  Invalid slots hold +inf, which
// is never "before" a valid entry.
template <int LIST>
DEVFN void rank_order(const float* __restrict__ depth_in, const float4* __restrict__ n_in, int w, const WmfTaps& t, WmfLds& l) {
    const int tid = threadIdx.x;
    f32x32 R0, R1, R2, R3;
    const float inf = __builtin_inff();
    auto load = [&](int k) -> float {                      // k is a compile-time constant after unrolling
        if (k == WMF_SLOTS - 1) return 0.0f;               // the zero slot
        if (k >= WMF_SLOTS) return inf;
        return slot_valid(t, k) ? slot_value<LIST>(depth_in, n_in, slot_pixel(t, k, w)) + 0.0f : inf;     // + 0.0f: -0 -> +0 (see SignCount8)
    };
#pragma unroll
    for (int k = 0; k < 32; k++) {
        R0[k] = load(k);
        R1[k] = load(32 + k);
        R2[k] = load(64 + k);
        R3[k] = load(96 + k);
    }
    auto entry = [&](int j) -> float {                    // j wave-uniform
        const int q = __builtin_amdgcn_readfirstlane(j);
        if (q < 32) return R0[q];
        if (q < 64) return R1[q - 32];
        if (q < 96) return R2[q - 64];
        return R3[q - 96];
    };
    for (int k0 = 0; k0 < WMF_SLOTS; k0 += 8) {
        float vk[8];
        int r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 8; c++) vk[c] = entry(min(k0 + c, WMF_SLOTS - 1));
        // entries before all eight: ties sort first.  One loop per 32-register vector, so that the choice of vector is not a
        // branch per entry (k0 is a multiple of 8: the ranges end on vector boundaries or inside one vector)
        // (at most 32 comparisons per sub-loop: one sign register per entry and sub-loop)
        SignCount8 sc;
#define WMF_BEFORE(VEC, LO, HI)                                                                  \
        if (k0 > (LO)) {                                                                         \
            const int hi_ = min(k0, (HI));                                                       \
            sc.clear();                                                                          \
            for (int j = (LO); j < hi_; j++) sc.push<false>(vk, VEC[__builtin_amdgcn_readfirstlane(j - (LO))]);   \
            sc.add_zeros(r, hi_ - (LO));                                                         \
        }
        WMF_BEFORE(R0, 0, 32)
        WMF_BEFORE(R1, 32, 64)
        WMF_BEFORE(R2, 64, 96)
        WMF_BEFORE(R3, 96, WMF_SLOTS)
#undef WMF_BEFORE
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {                             // the eight themselves
            if (k0 + jj >= WMF_SLOTS) break;
            const float vj = vk[jj];
#pragma unroll
            for (int c = 0; c < 8; c++) r[c] += (jj < c) ? (vj <= vk[c]) : ((jj > c) ? (vj < vk[c]) : 0);
        }
        // entries after all eight
#define WMF_AFTER(VEC, LO, HI)                                                                   \
        if (k0 + 8 < (HI)) {                                                                     \
            sc.clear();                                                                          \
            for (int j = max(k0 + 8, (LO)); j < (HI); j++) sc.push<true>(vk, VEC[__builtin_amdgcn_readfirstlane(j - (LO))]);   \
            sc.add_ones(r);                                                                      \
        }
        WMF_AFTER(R0, 0, 32)
        WMF_AFTER(R1, 32, 64)
        WMF_AFTER(R2, 64, 96)
        WMF_AFTER(R3, 96, WMF_SLOTS)
#undef WMF_AFTER
#pragma unroll
        for (int c = 0; c < 8; c++)
            if (k0 + c < WMF_SLOTS && slot_valid(t, k0 + c)) l.pos[r[c] * WMF_BLOCK + tid] = (unsigned char)(k0 + c);
    }
}
// clamped: with NaN values in a list the counted ranks are no permutation and a rank may stay unwritten (stale LDS byte)
DEVFN int pos_at(const WmfLds& l, int i) { return min((int)l.pos[i * WMF_BLOCK + threadIdx.x], WMF_SLOTS - 1); }
// Cumulative weight in rank order (gipuma.cu:1618-1650): acc += w[pos[i]] for i = 0 .. num-1, sequentially — the fp32 sums must
// be formed in exactly this order.  Each step is an LDS read (the slot) feeding a scratch read (its weight) at a per-lane
// address; taken one at a time that is ~1.5 us of latency per step (the weights of all resident waves do not fit L2).  So the
// walk goes in batches of 16: the 16 slots, then the 16 weights, are in flight together, and only the adds are sequential.
// Lanes past their own num add the zero slot's weight (acc + 0.0f == acc).  Returns the total; *kmed = the slot at which the
// sum first reaches `half`, or the last walked slot if it never does (FIND only).
#define WMF_WALK 16
template <bool FIND>
DEVFN float walk_ranked(const float* w, const WmfLds& l, int num, float half, int* kmed) {
    float acc = 0.f;
    bool found = false;
    int kfound = WMF_SLOTS - 1, klast = WMF_SLOTS - 1;
    for (int i0 = 0; __any(i0 < num); i0 += WMF_WALK) {
        int k[WMF_WALK];
        float wv[WMF_WALK];
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) k[u] = (i0 + u < num) ? pos_at(l, i0 + u) : WMF_SLOTS - 1;
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) wv[u] = w[k[u]];
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) {
            acc += wv[u];
            if (FIND) {
                const bool live = i0 + u < num;
                if (live) klast = k[u];
                if (live && !found && acc >= half) { found = true; kfound = k[u]; }
            }
        }
    }
    if (FIND) *kmed = found ? kfound : klast;
    return acc;
}
DEVFN int weighted_median_slot(const float* w, const WmfLds& l, int num, float half) {
    int k;
    walk_ranked<true>(w, l, num, half, &k);
    return k;
}
template <int LIST>
DEVFN float value_of_slot(const float* __restrict__ depth_in, const float4* __restrict__ n_in, int w, int h, const WmfTaps& t, int k) {
    return k == WMF_SLOTS - 1 ? 0.0f : slot_value<LIST>(depth_in, n_in, slot_pixel_safe(t, k, w, h));
}

DEVFN int collect_taps(const DevScene* __restrict__ sc, const float* __restrict__ scale_in, int x, int y, int radius, int gap, float sdiv, WmfTaps& t) {
    const float* __restrict__ img = sc->view[0].img;
    const int w = sc->w, h = sc->h;
    const float cen = img[(size_t)y * w + x];
    int num = 0, slot = 0;
    uint64_t lo = 0, hi = 0;
    for (int i = -radius; i <= radius; i += gap)
        for (int j = -radius; j <= radius; j += gap, slot++) {
            const int px = x + i, py = y + j;
            bool ok = px >= 0 && px < w && py >= 0 && py < h;
            const size_t q = ok ? (size_t)py * w + px : 0;
            ok = ok && scale_in[q] == 1.0f;
            float wt = 0.f;
            if (ok) {
                const float cd = fabsf(img[q] - cen);
                const float sd = sqrtf((float)(i * i + j * j)) / sdiv;
                wt = tsar_expf(-sd / 4.0f) * tsar_expf(-cd / 9.0f);   // sigma_spatial 2, sigma_color 3 (gipuma.cu:1537-1550)
                num++;
                if (slot < 64) lo |= 1ull << slot; else hi |= 1ull << (slot - 64);
            }
            t.w[slot] = wt;
        }
    // the zero slot the reference's sort drags in: value 0, weight 0, after every tap
    const int zs = WMF_SLOTS - 1;
    t.w[zs] = 0.f;
    hi |= 1ull << (zs - 64);
    t.valid_lo = lo; t.valid_hi = hi;
    t.num = num;
    t.x = x; t.y = y; t.radius = radius; t.gap = gap;
    return num;
}

// plane through the weighted-median-depth tap with the per-component weighted-median normal
DEVFN bool median_plane(const DevScene* __restrict__ sc, const float* __restrict__ depth_in, const float4* __restrict__ n_in, WmfTaps& t, WmfLds& l,
                        float4& out) {
    const DevRef& rf = sc->ref;
    const int num = t.num, w = sc->w;
    rank_order<WMF_DEPTH>(depth_in, n_in, w, t, l);
    const float wsum = walk_ranked<false>(t.w, l, num, 0.f, nullptr);
    const float half = wsum / 2.f;
    int weimid = -1;
    {
        // the depth walk breaks at the crossing; without one weimid stays unset (gipuma.cu:1641-1660)
        float acc = 0.f;
        bool found = false;
        int kf = 0;
        for (int i0 = 0; __any(i0 < num); i0 += WMF_WALK) {
            int k[WMF_WALK];
            float wv[WMF_WALK];
#pragma unroll
            for (int u = 0; u < WMF_WALK; u++) k[u] = (i0 + u < num) ? pos_at(l, i0 + u) : WMF_SLOTS - 1;
#pragma unroll
            for (int u = 0; u < WMF_WALK; u++) wv[u] = t.w[k[u]];
#pragma unroll
            for (int u = 0; u < WMF_WALK; u++) {
                acc += wv[u];
                if (i0 + u < num && !found && acc >= half) { found = true; kf = k[u]; }
            }
        }
        if (found) weimid = kf == WMF_SLOTS - 1 ? 0 : slot_pixel_safe(t, kf, w, sc->h);   // n[] of the zero slot is 0
    }
    float nm[3];
    rank_order<WMF_NX>(depth_in, n_in, w, t, l);
    nm[0] = value_of_slot<WMF_NX>(depth_in, n_in, w, sc->h, t, weighted_median_slot(t.w, l, num, half));
    rank_order<WMF_NY>(depth_in, n_in, w, t, l);
    nm[1] = value_of_slot<WMF_NY>(depth_in, n_in, w, sc->h, t, weighted_median_slot(t.w, l, num, half));
    rank_order<WMF_NZ>(depth_in, n_in, w, t, l);
    nm[2] = value_of_slot<WMF_NZ>(depth_in, n_in, w, sc->h, t, weighted_median_slot(t.w, l, num, half));
    if (weimid < 0) return false;
    const float depth_mid = rf.f * rf.baseline / depth_in[weimid];
    const double nrm = (double)sqrtf(dot3(nm, nm));   // `double xyzsqr = sqrtf(..)`, gipuma.cu:1663-1666
    nm[0] = (float)((double)nm[0] / nrm);
    nm[1] = (float)((double)nm[1] / nrm);
    nm[2] = (float)((double)nm[2] / nrm);
    out.x = nm[0]; out.y = nm[1]; out.z = nm[2];
    out.w = plane_offset(rf, nm, weimid % sc->w, weimid / sc->w, depth_mid);
    return true;
}

__global__ __launch_bounds__(WMF_BLOCK, 3) void wmf_detect_kernel(const DevScene* __restrict__ sc, const float* __restrict__ scale_in,
                                                               const float* __restrict__ depth, const float4* __restrict__ n4,
                                                               float* __restrict__ scale_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_BLOCK + threadIdx.x;
    if (p >= w * h) return;
    const int y = p / w, x = p - y * w;
    const int po = 1 << iter, repo = 1 << (3 - iter);
    const int radius = 80 / po, gap = 16 / po, ths = 24 / po;
    __shared__ WmfLds lds;
    WmfTaps t;
    float4 nm;
    float s = 0.0f;
    if (collect_taps(sc, scale_in, x, y, radius, gap, (float)repo, t) > 0 && median_plane(sc, depth, n4, t, lds, nm)) {
        const DevRef& rf = sc->ref;
        const float fb = rf.f * rf.baseline;
        const float disp_now = fb / plane_depth(rf, nm, x, y);
        const float disp_org = fb / plane_depth(rf, n4[p], x, y);
        s = fabsf(disp_now - disp_org) > (float)ths ? 0.0f : 1.0f;      // DEPTH_THS_MIN/MAX are 0 (gipuma.cu:38-39)
    }
    scale_out[p] = s;
}

__global__ __launch_bounds__(WMF_BLOCK, 3) void wmf_fill_kernel(const DevScene* __restrict__ sc, const int32_t* __restrict__ canny,
                                                             const float* __restrict__ region_text, const float* __restrict__ scale_in,
                                                             const float* __restrict__ depth_in, const float4* __restrict__ n_in,
                                                             float* __restrict__ scale_out, float* __restrict__ depth_out,
                                                             float4* __restrict__ n_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_BLOCK + threadIdx.x;
    if (p >= w * h) return;
    if (!(region_text[canny[p]] == 1.0f && scale_in[p] == 0.0f)) return;
    const int y = p / w, x = p - y * w;
    const int po = 1 << iter;
    const int radius = 5 * po, gap = po, ths = 32 / po;
    __shared__ WmfLds lds;
    WmfTaps t;
    float4 nm;
    const int num = collect_taps(sc, scale_in, x, y, radius, gap, (float)po, t);
    if (num < ths || num == 0) return;
    if (!median_plane(sc, depth_in, n_in, t, lds, nm)) return;
    const DevRef& rf = sc->ref;
    n_out[p] = nm;
    const float disp = rf.f * rf.baseline / plane_depth(rf, nm, x, y);
    if (disp <= sc->min_disp || disp >= sc->max_disp) { scale_out[p] = 0.0f; depth_out[p] = sc->min_disp; }
    else { scale_out[p] = 1.0f; depth_out[p] = disp; }
}

// iters launches of gipuma_WMF (final_pass = 0; the reference's loop runs 4, gipuma.cu:1809-1812) or of
// gipuma_WMF_Final (final_pass = 1; 6 in the reference, :1844-1847)
extern "C" int tsar_wmf(tsar_ctx* ctx, int iters, int final_pass) {
    if (!ctx) return TSAR_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return TSAR_ERR_HIP; }
    if (!ctx->have_state) { ctx->err = "no plane state"; return TSAR_ERR_STATE; }
    if (iters < 1 || iters > (final_pass ? 6 : 4)) { ctx->err = "tsar_wmf: iters must be 1..4 (detect) or 1..6 (final)"; return TSAR_ERR_INVALID; }
    if (final_pass && ctx->n_regions < 1) { ctx->err = "tsar_set_regions has not been called"; return TSAR_ERR_STATE; }
    const size_t np = (size_t)ctx->w * ctx->h;
    ScratchScope scratch(ctx);             // the launch-start snapshots come out of the context's scratch arena
    float* scale_snap = (float*)scratch.alloc(np * 4);
    float* depth_snap = final_pass ? (float*)scratch.alloc(np * 4) : nullptr;
    if (!scale_snap || (final_pass && !depth_snap)) {
        scratch.release();
        ctx->err = "device allocation failed";
        return TSAR_ERR_NOMEM;
    }
    const dim3 grid((unsigned)((np + WMF_BLOCK - 1) / WMF_BLOCK)), block(WMF_BLOCK);
    int rc = TSAR_OK;
    for (int it = 0; it < iters && rc == TSAR_OK; it++) {
        hipMemcpyAsync(scale_snap, ctx->scale, np * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (final_pass) {
            hipMemcpyAsync(depth_snap, ctx->depth, np * 4, hipMemcpyDeviceToDevice, ctx->stream);
            hipMemcpyAsync(ctx->buf[1].n4, ctx->buf[0].n4, np * 16, hipMemcpyDeviceToDevice, ctx->stream);
            ScopedKernelTimer tm(ctx, "wmf_fill");
            hipLaunchKernelGGL(wmf_fill_kernel, grid, block, 0, ctx->stream, ctx->dscene, ctx->canny, ctx->region_text, scale_snap, depth_snap,
                               ctx->buf[1].n4, ctx->scale, ctx->depth, ctx->buf[0].n4, it);
        } else {
            ScopedKernelTimer tm(ctx, "wmf_detect");
            hipLaunchKernelGGL(wmf_detect_kernel, grid, block, 0, ctx->stream, ctx->dscene, scale_snap, ctx->depth, ctx->buf[0].n4, ctx->scale, it);
        }
        if (hipGetLastError() != hipSuccess) { ctx->err = "wmf launch failed"; rc = TSAR_ERR_HIP; }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == TSAR_OK) { ctx->err = "wmf kernel failed"; rc = TSAR_ERR_HIP; }
    scratch.release();
    ctx->have_out = false;
    if (final_pass) ctx->cost_consistent = false;
    return rc;
}
