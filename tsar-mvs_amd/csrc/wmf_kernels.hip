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
//
// Round 5: TWO LANES PER PIXEL.  Round 4 kept a pixel's 122 keys in one lane — 244 VGPRs, the rest in AGPRs, ONE wave per SIMD, and
// the counters (profiles/r05/README.md section 5) showed that lone wave issuing a VALU instruction every ~4 cycles for 53 % of its
// time and waiting for the rest.  Now a pixel is a PAIR of neighbouring lanes (lane = 2 pixel + half): each half collects, weighs
// and sorts 64 of the pixel's 128 entries (121 taps, the zero slot, 6 pads) in 128 VGPRs, the second half on NEGATED keys so that
// its run comes out descending; one cross stage through a quad-perm DPP move — key = min(key, -partner's key), the same
// instruction for both halves — is the first stage of a bitonic merge, the other six stages are in-lane again.  Same keys, same
// total order, hence the same bits as before; two waves per SIMD by registers and LDS (21.5 KB per 32 pixels).
// The sequential parts (the rank-order walks, the plane through the medians) are run by both halves of a pair redundantly.
#include "tsar_device_math.h"

#define WMF_BLOCK 64     // threads per workgroup = one wave = 32 pixels x 2 halves
#define WMF_PIX 32
#define WMF_HALF 64      // entries per lane
#define WMF_ELEMS 128    // entries per pixel: tap slots 0..120 in enumeration order (i outer, j inner: slot = 11 ii + jj), slot 121 = the
#define WMF_TAPS 121     //   zero slot the reference's sort drags in, 122..127 = pads (never valid)
#define WMF_ZERO 121

// Tap slots are kept in ENUMERATION order, invalid ones flagged, instead of being compacted: the compacted index of the reference
// (`num++`) is monotone in the slot number, so the stable order among the valid taps is the same.  The weights live in LDS
// ([entry][pixel]); the values being sorted are re-gathered from the launch-start planes list by list — reads that neighbouring
// pixels share through L1 / L2; the pixel a slot refers to is recomputed from the slot number.
struct WmfTaps {
    uint64_t valid;                // bit r: this lane's entry r (= entry 64 half + r of the pixel) holds a tap (the zero slot always does)
    int num;                       // number of valid taps of the PIXEL (both halves), excluding the zero slot
    int x, y, radius, gap;         // geometry of the tap grid
};
// pixel of tap slot k (k < 121; offsets (-radius + ii gap, -radius + jj gap)), clamped into the image
DEVFN int slot_pixel_safe(const WmfTaps& t, int k, int w, int h) {
    const int ii = k / 11, jj = k - 11 * ii;
    const int px = min(max(t.x - t.radius + ii * t.gap, 0), w - 1), py = min(max(t.y - t.radius + jj * t.gap, 0), h - 1);
    return py * w + px;
}
struct WmfLds {
    float w[WMF_ELEMS * WMF_PIX];               // bilateral weight of every entry, [entry][pixel] (zero slot and pads: 0)
    unsigned char pos[WMF_ELEMS * WMF_PIX];     // the list's sorted order, [rank][pixel]
    // The weight exp(-sd / 4) exp(-cd / 9) (gipuma.cu:1537-1550) factorises: sd depends on the tap slot alone (the same for every
    // pixel of a launch), and on 8-bit imagery cd = |I(tap) - I(centre)| is an integer 0..255.  Both factors are tabulated once per
    // workgroup by the same expressions — the same bits — and a tap costs a table look-up and one multiply instead of a square root,
    // a division and two exponentials.  Float imagery keeps the direct form of the colour factor.
    float spatial[WMF_TAPS];                    // exp(-sd / 4) per tap slot
    float colour[256];                          // exp(-cd / 9) for cd = 0..255
};

// ---- the stable order from a SORTING NETWORK on (value, slot) keys ---------------------------------------------------------------------
// ONE 64-bit key whose ordering by v_min_f64 / v_max_f64 IS the lexicographic order of (value, slot).  The tap's fp32 value converts
// to fp64 exactly and leaves the low 29 mantissa bits zero; the slot number (7 bits) goes into the lowest bits — for a negative
// value 127 - slot, since a larger mantissa is then the smaller number.  Two different floats differ by at least 2^29 double-ulps,
// so the slot bits only ever break ties, in tap order.  +0 with slot bits is a subnormal double (fp64 subnormals are not flushed in
// this mode): zeros order by slot and stay between the negatives and the positives.  Invalid slots get huge keys (2^1023 + slot) and
// sort last; a NaN value becomes 2^1022 + slot (v_min_f64 returns the other operand) and an infinity +-2^1000 with its slot bits
// (classified BEFORE the slot bits go in: they would turn it into a signalling NaN), so the keys are always a permutation.  A
// compare-exchange is two instructions and moves no payload.
// Built in three steps, so that the non-finite cases cost nothing where there are none: (1) convert and put the slot bits in — pure bit
// arithmetic, an infinity or NaN just carries them along; (2) ONLY IF some lane of the wave loaded a non-finite value in this batch
// (a running maximum of the value bits tells): +-inf -> +-2^1000 (slot bits in an infinity's mantissa would make a signalling NaN,
// which v_min_f64 quiets instead of ordering, and the network would lose a slot; so -inf sorts first and +inf last among the values,
// ties in tap order, like the reference's `>` sort), NaN -> 2^1022 + slot; (3) the validity mask: an invalid slot is 2^1023 + slot.
DEVFN double wmf_key_bits(float v, int k) {
    const double d = (double)(v + 0.0f);                          // + 0.0f: -0 -> +0
    uint32_t lo = (uint32_t)__double_as_longlong(d);
    const uint32_t hi = (uint32_t)((unsigned long long)__double_as_longlong(d) >> 32);
    const uint32_t m = (uint32_t)((int32_t)hi >> 31);             // all ones for a negative value
    lo |= (uint32_t)k ^ (m & 127u);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
DEVFN double wmf_key_fix_nonfinite(double key, int k) {
    uint32_t lo = (uint32_t)__double_as_longlong(key), hi = (uint32_t)((unsigned long long)__double_as_longlong(key) >> 32);
    if ((hi & 0x7FF00000u) == 0x7FF00000u) {
        if ((hi & 0x000FFFFFu) == 0u && (lo >> 29) == 0u) hi -= 0x01800000u;      // +-inf (mantissa zero above the slot bits) -> +-2^1000
        else { hi = 0x7FD00000u; lo = (uint32_t)k; }                             // NaN -> 2^1022 + k
    }
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
DEVFN double wmf_key_mask(double key, int k, uint32_t valid_mask) {
    uint32_t lo = (uint32_t)__double_as_longlong(key), hi = (uint32_t)((unsigned long long)__double_as_longlong(key) >> 32);
    hi = (hi & valid_mask) | (0x7FE00000u & ~valid_mask);         // invalid slot: 2^1023 + k
    lo = (lo & valid_mask) | ((uint32_t)k & ~valid_mask);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
DEVFN int wmf_key_slot(double key) {
    const uint32_t lo = (uint32_t)__double_as_longlong(key), hi = (uint32_t)((unsigned long long)__double_as_longlong(key) >> 32);
    const uint32_t m = (uint32_t)((int32_t)hi >> 31);
    return (int)((lo ^ (m & 127u)) & 127u);
}
DEVFN double flip_sign(double key, uint32_t sign) {               // sign: 0 or 0x80000000
    const unsigned long long b = (unsigned long long)__double_as_longlong(key) ^ ((unsigned long long)sign << 32);
    return __longlong_as_double((long long)b);
}
// the partner lane's register (lane ^ 1) through a DPP move, quad_perm [1, 0, 3, 2]
DEVFN double partner_of(double key) {
    const uint32_t lo = (uint32_t)__double_as_longlong(key), hi = (uint32_t)((unsigned long long)__double_as_longlong(key) >> 32);
    const uint32_t plo = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0xB1, 0xf, 0xf, true);
    const uint32_t phi = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0xB1, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)phi << 32) | plo));
}
// The tap grid of a pixel whose whole window is inside the image: byte offset of entry (64 half + r) from the plane's base =
// origin + ii * col + jj * row with ii = entry / 11, jj = entry % 11 (px = x - radius + ii gap, py = y - radius + jj gap).
struct TapWindow {
    bool interior;                 // wave-uniform: every active lane's window is inside the image (and byte offsets fit 32 bits)
    unsigned origin;               // per pixel: byte offset of the window's first tap
    unsigned col, row;             // scalars: byte step of one tap column / tap row
};
DEVFN TapWindow tap_window(const WmfTaps& t, int w, int h, unsigned elem_bytes) {
    TapWindow tw;
    const bool in = t.x - t.radius >= 0 && t.x + t.radius < w && t.y - t.radius >= 0 && t.y + t.radius < h;
    tw.interior = __all(in) && ((unsigned long long)w * (unsigned)h * 16ull < (1ull << 32));
    const unsigned eb = (unsigned)__builtin_amdgcn_readfirstlane((int)elem_bytes), gap = (unsigned)__builtin_amdgcn_readfirstlane(t.gap),
                   ws = (unsigned)__builtin_amdgcn_readfirstlane(w);
    tw.col = gap * eb;
    tw.row = gap * ws * eb;
    tw.origin = (unsigned)((t.y - t.radius) * w + (t.x - t.radius)) * eb;
    return tw;
}
// a raw buffer descriptor over a whole plane (base made scalar): loads take a 32-bit vector offset AND a scalar offset, so the
// tap's own offset rides in an SGPR and the lane pays ONE vector instruction per tap (origin + half * (offset_b - offset_a))
DEVFN __amdgpu_buffer_rsrc_t plane_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    const unsigned long long bs = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc((void*)bs, 0, 0xffffffffu, 0x00020000);
}
DEVFN float tap_load(__amdgpu_buffer_rsrc_t rs, const TapWindow& tw, int r, int half) {     // r compile-time; entries >= 121 read the last tap's address
    const int ea = r, eb = r + WMF_HALF < WMF_TAPS ? r + WMF_HALF : WMF_TAPS - 1;
    const unsigned offa = (unsigned)(ea / 11) * tw.col + (unsigned)(ea % 11) * tw.row;      // scalar arithmetic
    const unsigned offb = (unsigned)(eb / 11) * tw.col + (unsigned)(eb % 11) * tw.row;
    // The hardware adds vector and scalar offset WITHOUT wrapping at 2^32 (a sum beyond it is out of range and reads 0), so the
    // vector part must never go below the origin: the scalar part is the SMALLER of the two offsets, the difference is added in the
    // half that has the larger one.  row = w col and |ii_b - ii_a| <= 10 < w: which is smaller is known at compile time.
    const bool b_larger = (eb % 11 != ea % 11) ? (eb % 11 > ea % 11) : (eb / 11 >= ea / 11);
    const unsigned hmask = 0u - (unsigned)half;                                              // all ones in the second half
    const unsigned voff = b_larger ? tw.origin + ((offb - offa) & hmask) : tw.origin + ((offa - offb) & ~hmask);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)(b_larger ? offa : offb), 0));
}
// pos[r] = entry with stable rank r among the pixel's 128 entries, for list `list` (0 depth, 1..3 normal components): wave-uniform
// runtime argument, so that the network exists once in the binary
__device__ __noinline__ void sort_order(int list, const float* __restrict__ depth_in, const float4* __restrict__ n_in, int w, int h, const WmfTaps& t, WmfLds& l) {
    // (global address space spelled out: a generic pointer out of a select compiles to flat loads)
    typedef const float __attribute__((address_space(1)))* gptr;
    const gptr base = list == 0 ? (gptr)depth_in : (gptr)((const float*)n_in + (list - 1));
    const int stride = list == 0 ? 1 : 4;
    const int half = threadIdx.x & 1, pix = threadIdx.x >> 1, e0 = half * WMF_HALF;
    const uint32_t sign = (uint32_t)half << 31;                   // the second half sorts NEGATED keys
    const uint32_t vw[2] = {(uint32_t)t.valid, (uint32_t)(t.valid >> 32)};
    double key[WMF_HALF];
    // Where every window of the wave lies inside the image (all but the border waves) no tap needs a clamp and a tap's address is
    // the window origin's plus an offset that depends on the entry alone: the two candidate offsets of register r (entry r of the
    // first half, 64 + r of the second) are SCALARS, and a lane's load costs two vector instructions (pick, add) instead of the
    // ~13 of the division by 11, four clamps and the index arithmetic.
    const TapWindow tw = tap_window(t, w, h, list == 0 ? 4u : 16u);
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc((const void*)base);
    // values in batches of 16, software-pipelined: the loads of batch b + 1 are issued before batch b is converted
    auto load_batch = [&](int r0, float (&v)[16]) {
        if (tw.interior) {
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = tap_load(rs, tw, r0 + u, half);
            return;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = min(e0 + r0 + u, WMF_TAPS - 1);          // (zero slot and pads: a tap's value is loaded and discarded)
            const int ii = e / 11, jj = e - 11 * ii;
            const int px = min(max(t.x - t.radius + ii * t.gap, 0), w - 1), py = min(max(t.y - t.radius + jj * t.gap, 0), h - 1);
            v[u] = base[(py * w + px) * stride];
        }
    };
    auto make_batch = [&](int r0, const float (&v)[16]) {
        uint32_t nonfinite = 0;                  // running maximum of (value bits << 1): >= 0xFF000000 iff an exponent was all ones
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int r = r0 + u, e = e0 + r;
            const float val = e >= WMF_TAPS ? 0.0f : v[u];                                         // the zero slot: value +0 (pads: invalid)
            key[r] = wmf_key_bits(val, e);
            nonfinite = max(nonfinite, __float_as_uint(val) << 1);
        }
        if (__any(nonfinite >= 0xFF000000u)) {   // rare, wave-uniform
#pragma unroll
            for (int u = 0; u < 16; u++) key[r0 + u] = wmf_key_fix_nonfinite(key[r0 + u], e0 + r0 + u);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int r = r0 + u;
            const uint32_t mask = (uint32_t)__builtin_amdgcn_sbfe((int)vw[r >> 5], r & 31, 1);      // v_bfe_i32 of one bit: 0 or all ones
            key[r] = flip_sign(wmf_key_mask(key[r], e0 + r, mask), sign);
        }
    };
    {
        float va[16], vb[16];
        load_batch(0, va);
        load_batch(16, vb);
        __builtin_amdgcn_sched_barrier(0);
        make_batch(0, va);
        load_batch(32, va);
        __builtin_amdgcn_sched_barrier(0);
        make_batch(16, vb);
        load_batch(48, vb);
        __builtin_amdgcn_sched_barrier(0);
        make_batch(32, va);
        make_batch(48, vb);
    }
#define CE(I, J) { double lo_, hi_; asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo_), "=&v"(hi_) : "v"(key[I]), "v"(key[J])); key[I] = lo_; key[J] = hi_; }
#define WMF_NET_SORT64
#include "wmf_sort_network.h"
#undef WMF_NET_SORT64
    // the cross stage: the first half keeps min(a_r, b_r) of its ascending run a and the partner's descending run b; the second half,
    // in its negated domain, min(-b_r, -a_r) = -max(a_r, b_r): one instruction sequence for both
#pragma unroll
    for (int r = 0; r < WMF_HALF; r++) {
        const double p = partner_of(key[r]);
        asm("v_min_f64 %0, %1, -%2" : "=v"(key[r]) : "v"(key[r]), "v"(p));     // (the negation is the instruction's source modifier)
    }
#define WMF_NET_MERGE64
#include "wmf_sort_network.h"
#undef WMF_NET_MERGE64
#undef CE
    // rank of register r: r in the first half; 127 - r in the second (its negated keys ascend = the keys descend)
    // (the slot bits of a NEGATED key are those of the key under the other sign; bit 7 of a key's low word is zero in every kind of
    // key — the conversion leaves 29 zero bits, the replacement keys hold the slot alone — so the low byte IS the slot: no mask)
    const uint32_t sign_fix = half ? ~0u : 0u;
#pragma unroll
    for (int r = 0; r < WMF_HALF; r++) {
        const int rank = half ? WMF_ELEMS - 1 - r : r;
        const uint32_t lo = (uint32_t)__double_as_longlong(key[r]), hi = (uint32_t)((unsigned long long)__double_as_longlong(key[r]) >> 32);
        const uint32_t m = (uint32_t)((int32_t)hi >> 31) ^ sign_fix;          // sign mask of the key as built
        l.pos[rank * WMF_PIX + pix] = (unsigned char)(lo ^ (m & 127u));
    }
    // Rank num — the largest valid entry, the one the reference's sort drops — is never looked up (walks cover ranks
    // < num): it is replaced by the zero slot, so that EVERY rank >= num now carries weight 0 and the walks need no "still inside my
    // own list" test per step (their batches of 16 may run past num; num <= 121 keeps them inside the 128 ranks).
    l.pos[t.num * WMF_PIX + pix] = (unsigned char)WMF_ZERO;
}

// (every rank's byte is written by this wave's sort before any walk reads it, and wmf_key_slot masks with 127: no clamp needed)
DEVFN int pos_at(const WmfLds& l, int i, int pix) { return (int)l.pos[i * WMF_PIX + pix]; }
// Cumulative weight in rank order (gipuma.cu:1618-1650): acc += w[pos[i]] for i = 0 .. num-1, sequentially — the fp32 sums must
// be formed in exactly this order.  Each step is an LDS read (the entry) feeding a second LDS read (its weight); the walk goes in
// batches of 16: the 16 entries, then the 16 weights, are in flight together, and only the adds are sequential.  Both halves of a
// pixel run the same walk (the same addresses: LDS broadcasts).  Steps past a pixel's own num add a weight of 0 (every rank >= num
// holds the zero slot or an invalid entry: acc + 0.0f == acc).  The rank at which the sum first reaches `half` is COUNTED, not searched: the weights are >= 0, so the
// partial sums never decrease and the first i with acc_i >= half is the number of steps with acc_i < half (three instructions per
// step: add, compare, add-with-carry; the steps past num repeat the last partial sum and are cut off by min(., num)).
// Returns the total (TOTAL walks); *below = that count, num if the sum never reaches `half` — the counting walks stop as soon as
// every pixel of the wave has crossed (the count cannot change after that), about half way on average.
#define WMF_WALK 16
template <bool TOTAL>     // TOTAL: walk every rank and return the sum; otherwise stop once every pixel of the wave has crossed `half`
DEVFN float walk_ranked(const WmfLds& l, int pix, int num, float half, int* below_out) {
    float acc = 0.f;
    int below = 0;
    for (int i0 = 0; TOTAL ? __any(i0 < num) : __any(i0 < num && acc < half); i0 += WMF_WALK) {
        int k[WMF_WALK];
        float wv[WMF_WALK];
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) k[u] = pos_at(l, i0 + u, pix);          // (ranks >= num: weight 0, see sort_order; i0 + u <= 127)
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) wv[u] = l.w[k[u] * WMF_PIX + pix];
#pragma unroll
        for (int u = 0; u < WMF_WALK; u++) {
            acc += wv[u];
            below += acc < half ? 1 : 0;
        }
    }
    if (below_out) *below_out = min(below, num);
    return acc;
}
// the entry at which the running sum first reaches `half`, or the last walked entry if it never does (gipuma.cu:1618-1650)
DEVFN int weighted_median_slot(const WmfLds& l, int pix, int num, float half) {
    int below;
    walk_ranked<false>(l, pix, num, half, &below);
    return pos_at(l, min(below, num - 1), pix);
}
// the two weight tables of WmfLds, by ALL 64 lanes of the workgroup (= one wave): called before any lane leaves the kernel
DEVFN void init_weight_tables(const DevScene* __restrict__ sc, int radius, int gap, float sdiv, WmfLds& l) {
    const int tid = threadIdx.x;
    for (int k = tid; k < WMF_TAPS; k += WMF_BLOCK) {
        const int ii = k / 11, jj = k - 11 * ii, i = -radius + ii * gap, j = -radius + jj * gap;
        const float sd = sqrtf((float)(i * i + j * j)) / sdiv;
        l.spatial[k] = tsar_expf(-sd / 4.0f);
    }
    if (sc->use_quad)
        for (int e = tid; e < 256; e += WMF_BLOCK) l.colour[e] = tsar_expf(-(float)e / 9.0f);
    __syncthreads();                            // one wave per workgroup: an LDS fence (s_waitcnt lgkmcnt(0) + s_barrier)
}

// this lane's 64 entries of the pixel: reliability flag and image value of each tap (16 taps = 32 loads in flight), weight to LDS
DEVFN int collect_taps(const DevScene* __restrict__ sc, const float* __restrict__ scale_in, int x, int y, int radius, int gap, WmfTaps& t, WmfLds& l) {
    const float* __restrict__ img = sc->view[0].img;
    const int w = sc->w, h = sc->h;
    const int half = threadIdx.x & 1, pix = threadIdx.x >> 1, e0 = half * WMF_HALF;
    const float cen = img[(size_t)y * w + x];
    const bool tables = sc->use_quad != 0;      // wave-uniform: 8-bit imagery (init_weight_tables ran before any lane left)
    int num = 0;
    uint32_t valid_lo = 0, valid_hi = 0;
    typedef const float __attribute__((address_space(1)))* gptr;
    const gptr gscale = (gptr)scale_in, gimg = (gptr)img;
    t.x = x; t.y = y; t.radius = radius; t.gap = gap;
    const TapWindow tw = tap_window(t, w, h, 4u);
    const __amdgpu_buffer_rsrc_t rs_scale = plane_rsrc(scale_in), rs_img = plane_rsrc(img);
#pragma unroll
    for (int r0 = 0; r0 < WMF_HALF; r0 += 16) {
        float sv[16], iv[16];
        bool inside[16];
        if (tw.interior) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                inside[u] = true;
                sv[u] = tap_load(rs_scale, tw, r0 + u, half);
                iv[u] = tap_load(rs_img, tw, r0 + u, half);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = min(e0 + r0 + u, WMF_TAPS - 1);
                const int ii = e / 11, jj = e - 11 * ii;
                const int px = x - radius + ii * gap, py = y - radius + jj * gap;
                inside[u] = px >= 0 && px < w && py >= 0 && py < h;
                const int q = min(max(py, 0), h - 1) * w + min(max(px, 0), w - 1);
                sv[u] = gscale[q];
                iv[u] = gimg[q];
            }
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int r = r0 + u, e = e0 + r;
            const bool tap = e < WMF_TAPS;
            const bool ok = tap && inside[u] && sv[u] == 1.0f;
            // sigma_spatial 2, sigma_color 3 (gipuma.cu:1537-1550): wt = exp(-sd / 4) * exp(-cd / 9), from the tables on 8-bit imagery
            const float cd = fabsf(iv[u] - cen);
            const float ws = l.spatial[min(e, WMF_TAPS - 1)];
            const float wc = tables ? l.colour[(int)cd & 255] : tsar_expf(-cd / 9.0f);       // (8-bit imagery: cd is an integer 0..255)
            l.w[e * WMF_PIX + pix] = ok ? ws * wc : 0.f;          // (zero slot and pads: weight 0)
            num += ok ? 1 : 0;
            const uint32_t bit = (uint32_t)(ok || e == WMF_ZERO) << (r & 31);     // the zero slot the reference's sort drags in: always valid
            if (r < 32) valid_lo |= bit; else valid_hi |= bit;
        }
    }
    num += __builtin_amdgcn_mov_dpp(num, 0xB1, 0xf, 0xf, true);   // + the partner half's taps
    t.valid = ((uint64_t)valid_hi << 32) | valid_lo;
    t.num = num;
    return num;
}

// plane through the weighted-median-depth tap with the per-component weighted-median normal
DEVFN bool median_plane(const DevScene* __restrict__ sc, const float* __restrict__ depth_in, const float4* __restrict__ n_in, WmfTaps& t, WmfLds& l,
                        float4& out) {
    const DevRef& rf = sc->ref;
    const int num = t.num, w = sc->w, pix = threadIdx.x >> 1;
    sort_order(0, depth_in, n_in, w, sc->h, t, l);
    const float wsum = walk_ranked<true>(l, pix, num, 0.f, nullptr);
    const float half = wsum / 2.f;
    int weimid = -1;
    {
        // the depth walk breaks at the crossing; without one weimid stays unset (gipuma.cu:1641-1660)
        int below;
        walk_ranked<false>(l, pix, num, half, &below);
        if (below < num) {
            const int kf = pos_at(l, below, pix);
            weimid = kf >= WMF_TAPS ? 0 : slot_pixel_safe(t, kf, w, sc->h);   // n[] of the zero slot is 0
        }
    }
    float nm[3];
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        sort_order(1 + c, depth_in, n_in, w, sc->h, t, l);
        const int k = weighted_median_slot(l, pix, num, half);
        nm[c] = k >= WMF_TAPS ? 0.0f : ((const float*)(n_in + slot_pixel_safe(t, k, w, sc->h)))[c];     // n[] of the zero slot is 0
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

__global__ __launch_bounds__(WMF_BLOCK, 2) void wmf_detect_kernel(const DevScene* __restrict__ sc, const float* __restrict__ scale_in,
                                                               const float* __restrict__ depth, const float4* __restrict__ n4,
                                                               float* __restrict__ scale_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_PIX + (threadIdx.x >> 1);
    const int po = 1 << iter, repo = 1 << (3 - iter);
    const int radius = 80 / po, gap = 16 / po, ths = 24 / po;
    __shared__ WmfLds lds;
    init_weight_tables(sc, radius, gap, (float)repo, lds);
    if (p >= w * h) return;                     // (both halves of a pixel together)
    const int y = p / w, x = p - y * w;
    WmfTaps t;
    float4 nm;
    float s = 0.0f;
    if (collect_taps(sc, scale_in, x, y, radius, gap, t, lds) > 0 && median_plane(sc, depth, n4, t, lds, nm)) {
        const DevRef& rf = sc->ref;
        const float fb = rf.f * rf.baseline;
        const float disp_now = fb / plane_depth(rf, nm, x, y);
        const float disp_org = fb / plane_depth(rf, n4[p], x, y);
        s = fabsf(disp_now - disp_org) > (float)ths ? 0.0f : 1.0f;      // DEPTH_THS_MIN/MAX are 0 (gipuma.cu:38-39)
    }
    if ((threadIdx.x & 1) == 0) scale_out[p] = s;
}

__global__ __launch_bounds__(WMF_BLOCK, 2) void wmf_fill_kernel(const DevScene* __restrict__ sc, const int32_t* __restrict__ canny,
                                                             const float* __restrict__ region_text, const float* __restrict__ scale_in,
                                                             const float* __restrict__ depth_in, const float4* __restrict__ n_in,
                                                             float* __restrict__ scale_out, float* __restrict__ depth_out,
                                                             float4* __restrict__ n_out, int iter) {
    const int w = sc->w, h = sc->h;
    const int p = blockIdx.x * WMF_PIX + (threadIdx.x >> 1);
    const bool active = p < w * h && region_text[canny[p]] == 1.0f && scale_in[p] == 0.0f;
    if (!__any(active)) return;                 // (most workgroups: nothing unreliable in a textured region)
    const int po = 1 << iter;
    const int radius = 5 * po, gap = po, ths = 32 / po;
    __shared__ WmfLds lds;
    init_weight_tables(sc, radius, gap, (float)po, lds);
    if (!active) return;                        // (both halves of a pixel together)
    const int y = p / w, x = p - y * w;
    WmfTaps t;
    float4 nm;
    const int num = collect_taps(sc, scale_in, x, y, radius, gap, t, lds);
    if (num < ths || num == 0) return;
    if (!median_plane(sc, depth_in, n_in, t, lds, nm)) return;
    if (threadIdx.x & 1) return;
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
    const dim3 grid((unsigned)((np + WMF_PIX - 1) / WMF_PIX)), block(WMF_BLOCK);
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
