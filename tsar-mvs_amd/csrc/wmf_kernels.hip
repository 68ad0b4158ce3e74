// wmf_kernels.hip — multi-scale weighted median filter of the TSAR refinement:
// gipuma_WMF (reliability detection, reference gipuma.cu:1499-1698) and gipuma_WMF_Final
// (fill of unreliable pixels in textured regions, :1294-1497).
//
// The reference bubble-sorts four (value, weight) lists of up to 121 taps per pixel in 5 kB of
// per-thread local memory, O(n^2) data-moving swaps.  Here the same STABLE sorted order (ties keep tap order) comes out of a
// sorting network on 64-bit (value, slot) keys held in registers (sort_order below), and the lists are walked in that order.
// The reference's sort also touches slot `num` (a zero entry joins and the largest entry drops out); that is
// reproduced by sorting num + 1 entries and walking the first num.
// Neighbours are read from launch-start copies of scale / depth / planes (the reference reads what
// other threads of the same launch are writing).
#include "tsar_device_math.h"

#define WMF_BLOCK 64
#define WMF_SLOTS 122   // 11 x 11 tap slots in enumeration order + the zero slot the reference's sort drags in (slot 121)

// Tap slots are kept in ENUMERATION order (i outer, j inner: slot = 11 ii + jj), invalid ones flagged, instead of being
// compacted: the compacted index of the reference (`num++`) is monotone in the slot number, so the stable order among the valid taps
// is the same — and a slot number is a compile-time constant of the unrolled key construction.  The weights live in LDS
// ([slot][thread]: a walk's per-lane slot lands every lane on its own bank); the values being sorted are re-gathered from the
// launch-start planes list by list — reads that neighbouring pixels share through L1 / L2; the pixel a slot refers to is recomputed
// from the slot number.
struct WmfTaps {
    uint64_t valid_lo, valid_hi;   // bit t of (hi:lo): slot t holds a tap (slot 121, the zero slot, always does)
    int num;                       // number of valid taps, excluding the zero slot
    int x, y, radius, gap;         // geometry of the tap grid
};
// pixel of tap slot k (k < 121; slot = 11 ii + jj, offsets (-radius + ii gap, -radius + jj gap)), clamped into the image: a slot
// number that comes out of a walk is per-lane
DEVFN int slot_pixel_safe(const WmfTaps& t, int k, int w, int h) {
    const int ii = k / 11, jj = k - 11 * ii;
    const int px = min(max(t.x - t.radius + ii * t.gap, 0), w - 1), py = min(max(t.y - t.radius + jj * t.gap, 0), h - 1);
    return py * w + px;
}
// Per-workgroup staging of the sorted order: pos[r] = slot with stable rank r (bytes, [rank][thread]).
struct WmfLds {
    float w[WMF_SLOTS * WMF_BLOCK];             // bilateral weight of every tap slot, [slot][thread]
    unsigned char pos[WMF_SLOTS * WMF_BLOCK];   // the list's sorted order, [rank][thread]
    // The weight exp(-sd / 4) exp(-cd / 9) (gipuma.cu:1537-1550) factorises: sd depends on the tap slot alone (the same for every
    // pixel of a launch), and on 8-bit imagery cd = |I(tap) - I(centre)| is an integer 0..255.  Both factors are tabulated once per
    // workgroup by the same expressions — the same bits — and a tap costs a table look-up and one multiply instead of a square root,
    // a division and two exponentials (~50 of the ~55 instructions of a tap; round 5).  Float imagery keeps the direct form.
    float spatial[WMF_SLOTS - 1];               // exp(-sd / 4) per tap slot
    float colour[256];                          // exp(-cd / 9) for cd = 0..255
};

// ---- the stable order from a SORTING NETWORK on (value, slot) keys ---------------------------------------------------------------------
// Rounds 2-3 counted every tap's rank: O(n^2), 122^2 comparisons of 2 instructions per list with the list in 128 VGPRs (233 ms per
// launch in round 1 from an LDS copy, 104.6 with the registers and mask-free sign counting).  A sorting network is O(n log^2 n), but the order wanted
// is the STABLE one — ties keep tap order, and ties are the common case (planes spread by verbatim copies) — so the key has to carry
// the slot, and a two-register key would need its payload moved through lane masks (v_cndmask issues at 7.7x a v_fma here).  The way
// out: ONE 64-bit key whose ordering by v_min_f64 / v_max_f64 IS the lexicographic order of (value, slot).  The tap's fp32 value
// converts to fp64 exactly and leaves the low 29 mantissa bits zero; the slot number (7 bits) goes into the lowest bits — for a
// negative value 127 - slot, since a larger mantissa is then the smaller number.  Two different floats differ by at least 2^29
// double-ulps, so the slot bits only ever break ties, in tap order.  +0 with slot bits is a subnormal double (fp64 subnormals are not
// flushed in this mode): zeros order by slot and stay between the negatives and the positives.  Invalid slots get huge keys
// (2^1023 + slot) and sort last; a NaN value becomes 2^1022 + slot (v_min_f64 returns the other operand) and an infinity +-2^1000 with
// its slot bits (classified BEFORE the slot bits go in: they would turn it into a signalling NaN), so the keys are always a
// permutation.  A compare-exchange is two instructions, the network (wmf_sort_network.h: Batcher's odd-even merge sort, 1401
// comparators) 2 802 per list against ~30 000 for the counting; the 122 keys live in 244 VGPRs (the rest spills to AGPRs), so the
// kernel runs ONE wave per SIMD — with a ninth of the instructions to issue, and with everything a lone wave would wait for kept
// short: the values of a list are loaded in batches of 32 before the first is converted, weights and sorted slots live in LDS.
// Measured at 24 MP: 100 -> 65 ms per detection launch with the weights still in scratch, see profiles/r04.
DEVFN double wmf_key(float v, int k, uint32_t valid_mask) {       // valid_mask: all ones / zero
    const double d = (double)(v + 0.0f);                          // + 0.0f: -0 -> +0
    uint32_t lo = (uint32_t)__double_as_longlong(d), hi = (uint32_t)((unsigned long long)__double_as_longlong(d) >> 32);
    const uint32_t m = (uint32_t)((int32_t)hi >> 31);             // all ones for a negative value
    // +-inf: slot bits in an infinity's mantissa would make a signalling NaN, which v_min_f64 quiets instead of replacing (IEEE mode)
    // and the network would then lose a slot.  An infinity becomes +-2^1000 instead — beyond every finite float, below the NaN and
    // invalid-slot keys — so that -inf sorts first and +inf last among the values, ties in tap order, like the reference's `>` sort.
    hi -= ((hi & 0x7FFFFFFFu) == 0x7FF00000u && lo == 0u) ? 0x01800000u : 0u;      // 0x7FF.. -> 0x7E7..
    lo |= (uint32_t)k ^ (m & 127u);
    double key = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    const double nan_key = __longlong_as_double((long long)((0x7FD00000ull << 32) | (unsigned)k));
    asm("v_min_f64 %0, %1, %2" : "=v"(key) : "v"(key), "v"(nan_key));                    // NaN -> its own large key
    lo = (uint32_t)__double_as_longlong(key); hi = (uint32_t)((unsigned long long)__double_as_longlong(key) >> 32);
    hi = (hi & valid_mask) | (0x7FE00000u & ~valid_mask);         // invalid slot: 2^1023 + k
    lo = (lo & valid_mask) | ((uint32_t)k & ~valid_mask);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
DEVFN int wmf_key_slot(double key) {
    const uint32_t lo = (uint32_t)__double_as_longlong(key), hi = (uint32_t)((unsigned long long)__double_as_longlong(key) >> 32);
    const uint32_t m = (uint32_t)((int32_t)hi >> 31);
    return (int)((lo ^ (m & 127u)) & 127u);
}
// pos[r] = slot with stable rank r among the valid slots (the zero slot included), for list `list` (0 depth, 1..3 normal components):
// wave-uniform runtime argument, so that the network exists once in the binary
__device__ __noinline__ void sort_order(int list, const float* __restrict__ depth_in, const float4* __restrict__ n_in, int w, int h, const WmfTaps& t, WmfLds& l) {
    // (global address space spelled out: a generic pointer out of a select compiles to flat loads)
    typedef const float __attribute__((address_space(1)))* gptr;
    const gptr base = list == 0 ? (gptr)depth_in : (gptr)((const float*)n_in + (list - 1));
    const int stride = list == 0 ? 1 : 4;
    const uint32_t vw[4] = {(uint32_t)t.valid_lo, (uint32_t)(t.valid_lo >> 32), (uint32_t)t.valid_hi, (uint32_t)(t.valid_hi >> 32)};
    double key[WMF_SLOTS];
    // values in batches of 32, software-pipelined: the loads of batch b + 1 are issued before batch b is converted (one wave per SIMD
    // has nobody else to hide a load's latency behind, and while the keys are still being made there are registers to spare)
    auto load_batch = [&](int k0, float (&v)[32]) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            const int k = k0 + u;
            if (k >= WMF_SLOTS - 1) break;
            const int ii = k / 11, jj = k - 11 * ii;
            // (clamped like slot_pixel_safe: a slot outside the image is invalid, its value is loaded from the border and discarded)
            const int px = min(max(t.x - t.radius + ii * t.gap, 0), w - 1), py = min(max(t.y - t.radius + jj * t.gap, 0), h - 1);
            v[u] = base[(py * w + px) * stride];
        }
    };
    auto make_batch = [&](int k0, const float (&v)[32]) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            const int k = k0 + u;
            if (k >= WMF_SLOTS - 1) break;
            const uint32_t mask = (uint32_t)__builtin_amdgcn_sbfe((int)vw[k >> 5], k & 31, 1);      // v_bfe_i32 of one bit: 0 or all ones
            key[k] = wmf_key(v[u], k, mask);
        }
    };
    {
        float va[32], vb[32];
        load_batch(0, va);
        load_batch(32, vb);
        __builtin_amdgcn_sched_barrier(0);
        make_batch(0, va);
        load_batch(64, va);
        __builtin_amdgcn_sched_barrier(0);
        make_batch(32, vb);
        load_batch(96, vb);
        __builtin_amdgcn_sched_barrier(0);
        make_batch(64, va);
        make_batch(96, vb);
    }
    key[WMF_SLOTS - 1] = __longlong_as_double((long long)(WMF_SLOTS - 1));       // the zero slot: value +0, slot 121, always valid
#define CE(I, J) { double lo_, hi_; asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo_), "=&v"(hi_) : "v"(key[I]), "v"(key[J])); key[I] = lo_; key[J] = hi_; }
#include "wmf_sort_network.h"
#undef CE
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < WMF_SLOTS; r++) l.pos[r * WMF_BLOCK + tid] = (unsigned char)wmf_key_slot(key[r]);
}

// (clamped: a slot number is at most 121 whatever the byte holds)
DEVFN int pos_at(const WmfLds& l, int i) { return min((int)l.pos[i * WMF_BLOCK + threadIdx.x], WMF_SLOTS - 1); }
// Cumulative weight in rank order (gipuma.cu:1618-1650): acc += w[pos[i]] for i = 0 .. num-1, sequentially — the fp32 sums must
// be formed in exactly this order.  Each step is an LDS read (the slot) feeding a second LDS read (its weight) at a per-lane
// address; the walk goes in batches of 16: the 16 slots, then the 16 weights, are in flight together, and only the adds are sequential.
// Lanes past their own num add the zero slot's weight (acc + 0.0f == acc).  Returns the total; *kmed = the slot at which the
// sum first reaches `half`, or the last walked slot if it never does (FIND only).
#define WMF_WALK 16
template <bool FIND>
DEVFN float walk_ranked(const WmfLds& l, int num, float half, int* kmed) {
    float acc = 0.f;
    bool found = false;
    int kfound = WMF_SLOTS - 1, klast = WMF_SLOTS - 1;
    for (int i0 = 0; __any(i0 < num); i0 += WMF_WALK) {
        int k[WMF_WALK];
        float wv[WMF_WALK];
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) k[u] = (i0 + u < num) ? pos_at(l, i0 + u) : WMF_SLOTS - 1;
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) wv[u] = l.w[k[u] * WMF_BLOCK + threadIdx.x];
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
DEVFN int weighted_median_slot(const WmfLds& l, int num, float half) {
    int k;
    walk_ranked<true>(l, num, half, &k);
    return k;
}
// the two weight tables of WmfLds, by ALL 64 lanes of the workgroup (= one wave): called before any lane leaves the kernel
DEVFN void init_weight_tables(const DevScene* __restrict__ sc, int radius, int gap, float sdiv, WmfLds& l) {
    const int tid = threadIdx.x;
    for (int k = tid; k < WMF_SLOTS - 1; k += WMF_BLOCK) {
        const int ii = k / 11, jj = k - 11 * ii, i = -radius + ii * gap, j = -radius + jj * gap;
        const float sd = sqrtf((float)(i * i + j * j)) / sdiv;
        l.spatial[k] = tsar_expf(-sd / 4.0f);
    }
    if (sc->use_quad)
        for (int e = tid; e < 256; e += WMF_BLOCK) l.colour[e] = tsar_expf(-(float)e / 9.0f);
    __syncthreads();                            // one wave per workgroup: an LDS fence (s_waitcnt lgkmcnt(0) + s_barrier)
}

DEVFN int collect_taps(const DevScene* __restrict__ sc, const float* __restrict__ scale_in, int x, int y, int radius, int gap, float sdiv, WmfTaps& t, WmfLds& l) {
    const float* __restrict__ img = sc->view[0].img;
    const int w = sc->w, h = sc->h;
    const float cen = img[(size_t)y * w + x];
    const bool tables = sc->use_quad != 0;      // wave-uniform: 8-bit imagery (init_weight_tables ran before any lane left)
    int num = 0;
    uint64_t lo = 0, hi = 0;
    // The grid is 11 x 11 in every launch (radius = 5 gap).  One column of 11 taps at a time: the 22 loads (reliability flag and
    // image value, from clamped positions) are issued together and the taps computed afterwards — the kernel runs one wave per SIMD,
    // so a load that waits for the one before it is time nobody else fills.
    typedef const float __attribute__((address_space(1)))* gptr;
    const gptr gscale = (gptr)scale_in, gimg = (gptr)img;
#pragma unroll 1
    for (int ii = 0; ii < 11; ii++) {
        const int i = -radius + ii * gap, px = x + i;
        const bool okx = px >= 0 && px < w;
        const int pxc = min(max(px, 0), w - 1);
        float sv[11], iv[11];
#pragma unroll
        for (int jj = 0; jj < 11; jj++) {
            const int pyc = min(max(y - radius + jj * gap, 0), h - 1);
            sv[jj] = gscale[pyc * w + pxc];
            iv[jj] = gimg[pyc * w + pxc];
        }
#pragma unroll
        for (int jj = 0; jj < 11; jj++) {
            const int j = -radius + jj * gap, py = y + j, slot = 11 * ii + jj;
            const bool ok = okx && py >= 0 && py < h && sv[jj] == 1.0f;
            // sigma_spatial 2, sigma_color 3 (gipuma.cu:1537-1550): wt = exp(-sd / 4) * exp(-cd / 9), from the tables on 8-bit imagery
            // (computed for every lane, kept where the tap is valid: no divergent block per tap)
            const float cd = fabsf(iv[jj] - cen);
            const float ws = l.spatial[slot];
            const float wc = tables ? l.colour[min((int)cd, 255)] : tsar_expf(-cd / 9.0f);
            const float wt = ok ? ws * wc : 0.f;
            num += ok ? 1 : 0;
            if (slot < 64) lo |= (uint64_t)ok << slot; else hi |= (uint64_t)ok << (slot - 64);
            l.w[slot * WMF_BLOCK + threadIdx.x] = wt;
        }
    }
    // the zero slot the reference's sort drags in: value 0, weight 0, after every tap
    const int zs = WMF_SLOTS - 1;
    l.w[zs * WMF_BLOCK + threadIdx.x] = 0.f;
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
    sort_order(0, depth_in, n_in, w, sc->h, t, l);
    const float wsum = walk_ranked<false>(l, num, 0.f, nullptr);
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
            for (int u = 0; u < WMF_WALK; u++) wv[u] = l.w[k[u] * WMF_BLOCK + threadIdx.x];
#pragma unroll
            for (int u = 0; u < WMF_WALK; u++) {
                acc += wv[u];
                if (i0 + u < num && !found && acc >= half) { found = true; kf = k[u]; }
            }
        }
        if (found) weimid = kf == WMF_SLOTS - 1 ? 0 : slot_pixel_safe(t, kf, w, sc->h);   // n[] of the zero slot is 0
    }
    float nm[3];
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        sort_order(1 + c, depth_in, n_in, w, sc->h, t, l);
        const int k = weighted_median_slot(l, num, half);
        nm[c] = k == WMF_SLOTS - 1 ? 0.0f : ((const float*)(n_in + slot_pixel_safe(t, k, w, sc->h)))[c];     // n[] of the zero slot is 0
    }
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

__global__ __launch_bounds__(WMF_BLOCK, 1) void wmf_detect_kernel(const DevScene* __restrict__ sc, const float* __restrict__ scale_in,
                                                               const float* __restrict__ depth, const float4* __restrict__ n4,
                                                               float* __restrict__ scale_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_BLOCK + threadIdx.x;
    const int po = 1 << iter, repo = 1 << (3 - iter);
    const int radius = 80 / po, gap = 16 / po, ths = 24 / po;
    __shared__ WmfLds lds;
    init_weight_tables(sc, radius, gap, (float)repo, lds);
    if (p >= w * h) return;
    const int y = p / w, x = p - y * w;
    WmfTaps t;
    float4 nm;
    float s = 0.0f;
    if (collect_taps(sc, scale_in, x, y, radius, gap, (float)repo, t, lds) > 0 && median_plane(sc, depth, n4, t, lds, nm)) {
        const DevRef& rf = sc->ref;
        const float fb = rf.f * rf.baseline;
        const float disp_now = fb / plane_depth(rf, nm, x, y);
        const float disp_org = fb / plane_depth(rf, n4[p], x, y);
        s = fabsf(disp_now - disp_org) > (float)ths ? 0.0f : 1.0f;      // DEPTH_THS_MIN/MAX are 0 (gipuma.cu:38-39)
    }
    scale_out[p] = s;
}

__global__ __launch_bounds__(WMF_BLOCK, 1) void wmf_fill_kernel(const DevScene* __restrict__ sc, const int32_t* __restrict__ canny,
                                                             const float* __restrict__ region_text, const float* __restrict__ scale_in,
                                                             const float* __restrict__ depth_in, const float4* __restrict__ n_in,
                                                             float* __restrict__ scale_out, float* __restrict__ depth_out,
                                                             float4* __restrict__ n_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_BLOCK + threadIdx.x;
    const bool active = p < w * h && region_text[canny[p]] == 1.0f && scale_in[p] == 0.0f;
    if (!__any(active)) return;                 // (most workgroups: nothing unreliable in a textured region)
    const int po = 1 << iter;
    const int radius = 5 * po, gap = po, ths = 32 / po;
    __shared__ WmfLds lds;
    init_weight_tables(sc, radius, gap, (float)po, lds);
    if (!active) return;
    const int y = p / w, x = p - y * w;
    WmfTaps t;
    float4 nm;
    const int num = collect_taps(sc, scale_in, x, y, radius, gap, (float)po, t, lds);
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
