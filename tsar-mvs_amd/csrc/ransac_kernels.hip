// ransac_kernels.hip — per-region plane fit of the TSAR "correlative refinement"
// (reference main.cpp:1520-1730: single-threaded CPU RANSAC, ~7e8 point-plane residuals per region,
// where the live pipeline's wall time goes; calcLinePara main.cpp:147-164).
//
// MI355X shape: (1) the reliable pixels of all textureless regions are compacted in raster order by a
// device select + a stable radix sort by region (rocPRIM; plain library primitives), (2) one kernel
// back-projects them to 3-D points, (3) the fit: the 10 000 three-point hypotheses are counted on
// 125 workgroups per region and phase, the 4 000 sequential perturbation steps on one workgroup per
// region with a lookahead that shares each pass over the points between the candidate planes of
// several steps (see "the fit, spread over the chip" below).  Arithmetic is double precision in the
// reference's operation order, so the result is reproducible bit for bit on the CPU oracle.
//
// Deterministic choices (the reference uses rand() and a time-seeded shuffle): raster-order point
// lists, even subsampling to 49999 points above 50000, Philox draws keyed by (draw, stage, region).
#include <chrono>
#include <mutex>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "tsar_device_math.h"

#define TSAR_FLAG_FIX_PLANE_FIT (1u << 3)
#define RS_BLOCK 1024
#define RS_MAXPTS 49999

// flag[p] = 1 where pixel p is reliable (scale == 1) and lies in a textureless region (main.cpp:1527-1536); a plain streaming
// kernel, so that the compaction itself is rocPRIM's flagged select over bytes (the predicate form, with its two dependent
// gathers per element inside the select kernel, took 6 ms over 24 M pixels)
__global__ void ransac_flag_kernel(const int32_t* __restrict__ canny, const float* __restrict__ scale, const int32_t* __restrict__ slot_of_region,
                                   size_t np, uint8_t* __restrict__ flag) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (size_t)gridDim.x * blockDim.x)
        flag[p] = (scale[p] == 1.0f && slot_of_region[canny[p]] >= 0) ? 1 : 0;
}

__global__ void ransac_keys_kernel(const uint32_t* __restrict__ pix, int n, const int32_t* __restrict__ canny,
                                   const int32_t* __restrict__ slot_of_region, uint32_t* __restrict__ keys, int* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = (uint32_t)slot_of_region[canny[pix[i]]];
    keys[i] = s;
    // one atomic per distinct slot of the wave: the list is in raster order, so a wave's 64 pixels lie in one or two regions,
    // and a few large regions would otherwise serialise millions of atomics on a handful of addresses (13 ms at 24 MP).
    // (The loop is wave-uniform over a mask that shrinks every trip; a `while (pending)` loop around readfirstlane(s) is folded
    // by the compiler into a single trip — the read is loop-invariant to it, and a trip that does not exit would never end.)
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(1);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t lead = (uint32_t)__shfl((int)s, first);
        const unsigned long long same = __ballot(s == lead);
        if (lane == first) atomicAdd(&counts[lead], __popcll(same));
        todo &= ~same;
    }
}

// main.cpp:1571-1594: depth from lines->depth (= f*b/depth), back-projection through M^-1
__global__ void ransac_points_kernel(const DevScene* __restrict__ sc, const uint32_t* __restrict__ pix_sorted, const uint32_t* __restrict__ keys_sorted,
                                     int n, const float* __restrict__ depth, const int* __restrict__ slot_start, const int* __restrict__ slot_total,
                                     const int* __restrict__ pts_start, float* __restrict__ pts) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int s = (int)keys_sorted[q];
    const int64_t i = q - slot_start[s], total = slot_total[s];
    const int64_t keep = total > 50000 ? RS_MAXPTS : total;
    const int64_t hi = ((i + 1) * keep) / total, lo = (i * keep) / total;
    if (hi <= lo) return;
    const DevRef& rf = sc->ref;
    const uint32_t p = pix_sorted[q];
    const int x = (int)(p % (uint32_t)sc->w), y = (int)(p / (uint32_t)sc->w);
    const float dd = rf.f * rf.baseline / depth[p];
    const float ptx = dd * (float)x - rf.P34[0], pty = dd * (float)y - rf.P34[1], ptz = dd - rf.P34[2];
    float* o = pts + 3 * ((size_t)pts_start[s] + (size_t)(hi - 1));
    o[0] = rf.Minv[0] * ptx + rf.Minv[1] * pty + rf.Minv[2] * ptz;
    o[1] = rf.Minv[3] * ptx + rf.Minv[4] * pty + rf.Minv[5] * ptz;
    o[2] = rf.Minv[6] * ptx + rf.Minv[7] * pty + rf.Minv[8] * ptz;
}

DEVFN void philox_raw(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1, uint32_t r[4]) {
    uint32_t c3 = 0u;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
DEVFN uint32_t rnd_index(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

// ---- the fit, spread over the chip ---------------------------------------------------------------------
// The reference's loop (main.cpp:1603-1711) is sequential only in how it COMPARES inlier counts:
//   stage 1: 10 000 three-point hypotheses; between two threshold adaptations (every 1 000) the inlier threshold is
//            fixed, so the 999 counts of a phase are independent of each other.  ransac_count_kernel evaluates them on
//            125 workgroups per region (8 hypotheses share each pass over the points), ransac_adapt_kernel then replays the
//            sequential accept rule on the stored counts — `cnt >= maximum` in index order picks the LAST hypothesis that
//            attains the phase's maximum count — and performs the adaptation step (hypothesis 1000 j, the `max2` re-count).
//   stage 2: 1 000 rounds x 4 perturbation scales, each perturbing the current best plane: truly sequential, one
//            workgroup per region.  ransac_refine_kernel can look G steps ahead: the 2^G - 1 candidate planes of every
//            accept / reject history of the next G steps are counted in ONE pass over the points, then the G decisions are
//            replayed in order on the counts.  Same planes, same counts, same decisions as the sequential loop (G = 1).
// State between kernels lives in RansacState (one per region slot); counts in cnt[slot][1000].
struct RansacState {
    double pl[4];          // best plane so far (a, b, c, d)
    int maximum;           // its inlier count
    float depth_abs_f;     // inlier threshold, kept as the float the reference keeps (main.cpp:1551-1552)
};

template <int BS>
DEVFN int block_count(const float* __restrict__ pts, int n, double a, double b, double c, double d, double thr, int* sh) {
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += BS) {
        const double resid = fabs((double)pts[3 * i] * a + (double)pts[3 * i + 1] * b + (double)pts[3 * i + 2] * c + d);
        cnt += resid < thr;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_down(cnt, o);
    __syncthreads();                       // previous readers of sh are done
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int wv = 0; wv < BS / 64; wv++) tot += sh[wv];
    return tot;
}

// The same count for NB planes in one pass over the points; pl in LDS (every thread reads all of them).
// (Deciding the comparison in single precision with an error band and falling back to doubles inside it was built and measured:
// bit-identical counts, no gain — a non-packed v_fma_f32 issues at the rate of v_fma_f64 on this chip, and the compares and the
// count do not pack.)
template <int BS, int NB>
DEVFN void block_count_batch(const float* __restrict__ pts, int n, const double (*pl)[4], double thr, int (*shb)[NB], int* out) {
    double P[NB][4];
#pragma unroll
    for (int q = 0; q < NB; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) P[q][e] = pl[q][e];
    int cnt[NB];
#pragma unroll
    for (int q = 0; q < NB; q++) cnt[q] = 0;
    for (int i = threadIdx.x; i < n; i += BS) {
        const double px = (double)pts[3 * i], py = (double)pts[3 * i + 1], pz = (double)pts[3 * i + 2];
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const double resid = fabs(px * P[q][0] + py * P[q][1] + pz * P[q][2] + P[q][3]);
            cnt[q] += resid < thr;
        }
    }
#pragma unroll
    for (int q = 0; q < NB; q++)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cnt[q] += __shfl_down(cnt[q], o);
    __syncthreads();                       // previous readers of shb are done
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < NB; q++) shb[threadIdx.x >> 6][q] = cnt[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB; q++) {
        int tot = 0;
#pragma unroll
        for (int wv = 0; wv < BS / 64; wv++) tot += shb[wv][q];
        out[q] = tot;
    }
}

// hypothesis k of main.cpp:1603-1640: three Philox-drawn points -> unit plane (calcLinePara main.cpp:147-164)
DEVFN void ransac_hypothesis(const float* __restrict__ pts, int n, int rg, int k, uint32_t k0, uint32_t k1, uint32_t flags, double* pl) {
    uint32_t r[4];
    philox_raw((uint32_t)k, 0x52414E53u, (uint32_t)rg, k0, k1, r);
    const float* p1 = pts + 3 * rnd_index(r[0], (uint32_t)n);
    const float* p2 = pts + 3 * rnd_index(r[1], (uint32_t)n);
    const float* p3 = pts + 3 * rnd_index(r[2], (uint32_t)n);
    const double x1 = p1[0], y1 = p1[1], z1 = p1[2], x2 = p2[0], y2 = p2[1], z2 = p2[2], x3 = p3[0], y3 = p3[1], z3 = p3[2];
    // calcLinePara main.cpp:159: the first component is written with (y3 - y1) twice
    double ta = (flags & TSAR_FLAG_FIX_PLANE_FIT) ? (y2 - y1) * (z3 - z1) - (z2 - z1) * (y3 - y1) : (y3 - y1) * (z3 - z1) - (z2 - z1) * (y3 - y1);
    double tb = (x3 - x1) * (z2 - z1) - (x2 - x1) * (z3 - z1);
    double tc = (x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1);
    double td = -(ta * x1 + tb * y1 + tc * z1);
    const double sq = sqrt(ta * ta + tb * tb + tc * tc);
    pl[0] = ta / sq; pl[1] = tb / sq; pl[2] = tc / sq; pl[3] = td / sq;
}

#define RS_PHASE 1000            // hypotheses between two threshold adaptations (main.cpp:1642)
#define RS_BATCH 8               // hypotheses sharing one pass over the points in ransac_count_kernel
#define RS_COUNT_BLOCK 256
#define RS_BATCHES ((RS_PHASE - 1 + RS_BATCH - 1) / RS_BATCH)     // 125 workgroups per region and phase

// Sequential accept rule over the stored counts of hypotheses base+1 .. base+999: `if (cnt >= maximum)` in index order ends
// on the last index attaining max(maximum, max cnt).  Returns that index (or -1) to every thread; *best_cnt its count.
DEVFN int replay_phase(const int* __restrict__ cnt, int maximum, int* sh_val, int* sh_idx, int* best_cnt) {
    int v = -1, ix = -1;
    for (int i = 1 + (int)threadIdx.x; i < RS_PHASE; i += RS_BLOCK) {
        const int c = cnt[i];
        if (c > v || (c == v && i > ix)) { v = c; ix = i; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int v2 = __shfl_down(v, o), i2 = __shfl_down(ix, o);
        if (v2 > v || (v2 == v && i2 > ix)) { v = v2; ix = i2; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh_val[threadIdx.x >> 6] = v; sh_idx[threadIdx.x >> 6] = ix; }
    __syncthreads();
    v = -1; ix = -1;
#pragma unroll
    for (int wv = 0; wv < RS_BLOCK / 64; wv++) {
        const int v2 = sh_val[wv], i2 = sh_idx[wv];
        if (v2 > v || (v2 == v && i2 > ix)) { v = v2; ix = i2; }
    }
    *best_cnt = v;
    return (ix >= 0 && v >= maximum) ? ix : -1;
}

// phase j = 0..9: (j > 0) replay the counts of hypotheses 1000(j-1)+1 .. 1000 j - 1, then the adaptation step k = 1000 j
// (main.cpp:1603-1662).  One 1024-thread workgroup per region slot.
__global__ __launch_bounds__(RS_BLOCK) void ransac_adapt_kernel(const float* __restrict__ pts_all, const int* __restrict__ pts_start,
                                                                const int* __restrict__ pts_count, const int* __restrict__ region_of_slot,
                                                                const float* __restrict__ region_size, uint32_t k0, uint32_t k1, uint32_t flags,
                                                                RansacState* __restrict__ state, const int* __restrict__ cnt_all, int phase) {
    __shared__ int sh[RS_BLOCK / 64], sh2[RS_BLOCK / 64];
    const int slot = blockIdx.x;
    const int rg = region_of_slot[slot];
    const int n = pts_count[slot];
    const float* __restrict__ pts = pts_all + 3 * (size_t)pts_start[slot];
    RansacState st;
    if (phase == 0) {
        st.pl[0] = 0; st.pl[1] = 0; st.pl[2] = 1; st.pl[3] = -1;
        st.maximum = 0;
        st.depth_abs_f = (float)(0.0003 * (double)sqrtf(region_size[rg] / 20));    // `float depth_abs = 0.0003 * sqrtf(size / 20)` main.cpp:1551-1552
    } else {
        st = state[slot];
    }
    if (n > 0) {
        if (phase > 0) {
            int best_cnt;
            const int ix = replay_phase(cnt_all + (size_t)slot * RS_PHASE, st.maximum, sh, sh2, &best_cnt);
            if (ix >= 0) {
                ransac_hypothesis(pts, n, rg, (phase - 1) * RS_PHASE + ix, k0, k1, flags, st.pl);
                st.maximum = best_cnt;
            }
        }
        if (phase < 10) {
            double pl[4];
            ransac_hypothesis(pts, n, rg, phase * RS_PHASE, k0, k1, flags, pl);
            double depth_abs = st.depth_abs_f;
            const int cnt = block_count<RS_BLOCK>(pts, n, pl[0], pl[1], pl[2], pl[3], depth_abs, sh);
            if (cnt >= st.maximum) { st.pl[0] = pl[0]; st.pl[1] = pl[1]; st.pl[2] = pl[2]; st.pl[3] = pl[3]; st.maximum = cnt; }
            // adaptive inlier threshold, main.cpp:1642-1661
            const double rat = (double)st.maximum / (double)n;
            if (rat < 0.3 && depth_abs < 0.003) {
                st.depth_abs_f = (float)((double)st.depth_abs_f + 0.0001);
            } else {
                const int max2 = block_count<RS_BLOCK>(pts, n, st.pl[0], st.pl[1], st.pl[2], st.pl[3], (double)st.depth_abs_f + 0.0001, sh);
                if ((double)max2 > (double)st.maximum + (double)n * 0.02) {
                    st.depth_abs_f = (float)((double)st.depth_abs_f + 0.0001);
                    st.maximum = max2;
                }
            }
        }
    }
    if (threadIdx.x == 0) state[slot] = st;
}

// inlier counts of hypotheses 1000 j + 1 + 8 b .. + 8 at the phase's threshold: grid (125, slots), 256 threads
__global__ __launch_bounds__(RS_COUNT_BLOCK) void ransac_count_kernel(const float* __restrict__ pts_all, const int* __restrict__ pts_start,
                                                                      const int* __restrict__ pts_count, const int* __restrict__ region_of_slot,
                                                                      uint32_t k0, uint32_t k1, uint32_t flags, const RansacState* __restrict__ state,
                                                                      int* __restrict__ cnt_all, int phase) {
    __shared__ int shb[RS_COUNT_BLOCK / 64][RS_BATCH];
    __shared__ double shpl[RS_BATCH][4];
    const int slot = blockIdx.y;
    const int n = pts_count[slot];
    if (n <= 0) return;
    const int rg = region_of_slot[slot];
    const float* __restrict__ pts = pts_all + 3 * (size_t)pts_start[slot];
    const int i0 = 1 + RS_BATCH * (int)blockIdx.x;                     // index within the phase, 1..999
    const int nb = min(RS_BATCH, RS_PHASE - i0);
    // one thread per hypothesis builds its plane (three dependent point loads, a double sqrt and four divisions)
    if (threadIdx.x < RS_BATCH) {
        double mine[4];
        ransac_hypothesis(pts, n, rg, phase * RS_PHASE + i0 + ((int)threadIdx.x < nb ? (int)threadIdx.x : 0), k0, k1, flags, mine);
#pragma unroll
        for (int e = 0; e < 4; e++) shpl[threadIdx.x][e] = mine[e];
    }
    __syncthreads();
    int cnt[RS_BATCH];
    block_count_batch<RS_COUNT_BLOCK, RS_BATCH>(pts, n, shpl, (double)state[slot].depth_abs_f, shb, cnt);
    if (threadIdx.x < nb) {
        int mine = 0;
#pragma unroll
        for (int q = 0; q < RS_BATCH; q++) mine = ((int)threadIdx.x == q) ? cnt[q] : mine;
        cnt_all[(size_t)slot * RS_PHASE + i0 + threadIdx.x] = mine;
    }
}

// perturbed plane of step t (round t / 4, scale t % 4) around `base`, main.cpp:1667-1700
DEVFN void ransac_perturb(const double* base, int t, int rg, uint32_t k0, uint32_t k1, double* out) {
    uint32_t r[4];
    philox_raw((uint32_t)t, 0x52414E54u, (uint32_t)rg, k0, k1, r);
    const int scn = t & 3;
    const int j = scn == 0 ? 2000 : (scn == 1 ? 200 : (scn == 2 ? 20 : 2));
    const int med = j / 2;
    const double da = (double)((int)rnd_index(r[0], (uint32_t)j) - med) / 10000;
    const double db = (double)((int)rnd_index(r[1], (uint32_t)j) - med) / 10000;
    const double dc = (double)((int)rnd_index(r[2], (uint32_t)j) - med) / 10000;
    const double dd = (double)((int)rnd_index(r[3], (uint32_t)j) - med) / 1000;
    double ra = base[0] + da, rb = base[1] + db, rc = base[2] + dc, rd = base[3] + dd;
    const double sq = sqrt(ra * ra + rb * rb + rc * rc);
    out[0] = ra / sq; out[1] = rb / sq; out[2] = rc / sq; out[3] = rd / sq;
}

// Stage 2 with a lookahead of G steps.  Candidate node q (1-based heap index, level g = floor(log2 q)) is the plane tried at
// step t0 + g if the decisions of steps t0 .. t0 + g - 1 were the bits of q below its leading one (1 = accepted), read from
// the most significant down.  Node q perturbs the base of its history: the candidate of its nearest ancestor-by-accept, or the
// current best plane if the history holds no accept.
template <int G>
__global__ __launch_bounds__(RS_BLOCK) void ransac_refine_kernel(const float* __restrict__ pts_all, const int* __restrict__ pts_start,
                                                                 const int* __restrict__ pts_count, const int* __restrict__ region_of_slot,
                                                                 uint32_t k0, uint32_t k1, uint32_t flags, const RansacState* __restrict__ state,
                                                                 const int* __restrict__ cnt_all, float4* __restrict__ region_n4,
                                                                 float* __restrict__ inlier_ratio) {
    constexpr int NODES = (1 << G) - 1;
    __shared__ int sh[RS_BLOCK / 64], sh2[RS_BLOCK / 64];
    __shared__ int shb[RS_BLOCK / 64][NODES];
    __shared__ double shpl[NODES][4];
    const int slot = blockIdx.x;
    const int rg = region_of_slot[slot];
    const int n = pts_count[slot];
    const float* __restrict__ pts = pts_all + 3 * (size_t)pts_start[slot];
    RansacState st = state[slot];
    if (n > 0) {
        {   // the last phase of stage 1 still has to be replayed
            int best_cnt;
            const int ix = replay_phase(cnt_all + (size_t)slot * RS_PHASE, st.maximum, sh, sh2, &best_cnt);
            if (ix >= 0) {
                ransac_hypothesis(pts, n, rg, 9 * RS_PHASE + ix, k0, k1, flags, st.pl);
                st.maximum = best_cnt;
            }
        }
        const double depth_abs = st.depth_abs_f;
        for (int t0 = 0; t0 < 4000; t0 += G) {
            const int g_here = min(G, 4000 - t0);
            // thread q-1 builds node q by walking its history from the root: at most G perturbations in a row
            if (threadIdx.x < NODES) {
                const int q = (int)threadIdx.x + 1;
                const int level = 31 - __clz(q);
                double base[4] = {st.pl[0], st.pl[1], st.pl[2], st.pl[3]};
                double cand[4];
                for (int g = 0; g <= level; g++) {
                    ransac_perturb(base, t0 + g, rg, k0, k1, cand);
                    if (g < level && ((q >> (level - 1 - g)) & 1)) { base[0] = cand[0]; base[1] = cand[1]; base[2] = cand[2]; base[3] = cand[3]; }
                }
#pragma unroll
                for (int e = 0; e < 4; e++) shpl[q - 1][e] = cand[e];
            }
            __syncthreads();
            int cnt[NODES];
            block_count_batch<RS_BLOCK, NODES>(pts, n, shpl, depth_abs, shb, cnt);
            // replay the G decisions in order (every thread, identically)
            int q = 1;
            for (int g = 0; g < g_here; g++) {
                int c = 0;
#pragma unroll
                for (int m = 0; m < NODES; m++) c = (m == q - 1) ? cnt[m] : c;
                const bool acc = c >= st.maximum;
                if (acc) {
#pragma unroll
                    for (int e = 0; e < 4; e++) st.pl[e] = shpl[q - 1][e];
                    st.maximum = c;
                }
                q = 2 * q + (acc ? 1 : 0);
            }
            __syncthreads();               // shpl is rewritten by the next group
        }
    }
    if (threadIdx.x == 0) {
        region_n4[rg] = make_float4((float)st.pl[0], (float)st.pl[1], (float)st.pl[2], (float)st.pl[3]);
        inlier_ratio[rg] = n > 0 ? (float)st.maximum / (float)n : 0.f;
    }
}

// Stage 2 as a speculative CHAIN of K steps.  Measured on full-size scenes, only 1.6-2 % of the 4 000 perturbation steps are
// accepted (and most of those at the finest scale, with an equal count), so the candidates of steps t0 .. t0 + K - 1 are all
// built from the current best plane — what they are if none of them is accepted —, counted in ONE pass over the points, and the
// decisions replayed in order: the first accepted step ends the pass (the later candidates started from the wrong plane) and the
// next pass begins right after it.  Same planes, counts and decisions as the sequential loop; ~K / (1 + 0.02 K) steps per pass
// instead of one, so the per-pass costs (two barriers, the cross-wave reduction, the perturbation) are paid ~7 times less often
// at K = 8 and what remains is the FP64 arithmetic of the counts.
template <int K>
__global__ __launch_bounds__(RS_BLOCK) void ransac_refine_chain_kernel(const float* __restrict__ pts_all, const int* __restrict__ pts_start,
                                                                       const int* __restrict__ pts_count, const int* __restrict__ region_of_slot,
                                                                       uint32_t k0, uint32_t k1, uint32_t flags, const RansacState* __restrict__ state,
                                                                       const int* __restrict__ cnt_all, float4* __restrict__ region_n4,
                                                                       float* __restrict__ inlier_ratio) {
    __shared__ int sh[RS_BLOCK / 64], sh2[RS_BLOCK / 64];
    __shared__ int shb[RS_BLOCK / 64][K];
    __shared__ double shpl[K][4];
    const int slot = blockIdx.x;
    const int rg = region_of_slot[slot];
    const int n = pts_count[slot];
    const float* __restrict__ pts = pts_all + 3 * (size_t)pts_start[slot];
    RansacState st = state[slot];
    if (n > 0) {
        {   // the last phase of stage 1 still has to be replayed
            int best_cnt;
            const int ix = replay_phase(cnt_all + (size_t)slot * RS_PHASE, st.maximum, sh, sh2, &best_cnt);
            if (ix >= 0) {
                ransac_hypothesis(pts, n, rg, 9 * RS_PHASE + ix, k0, k1, flags, st.pl);
                st.maximum = best_cnt;
            }
        }
        const double depth_abs = st.depth_abs_f;
        int t0 = 0;
        while (t0 < 4000) {                                   // every thread carries the same t0 and state
            const int k_here = min(K, 4000 - t0);
            if (threadIdx.x < K) {                            // (slots beyond k_here repeat the last step: counted, never read)
                double cand[4];
                ransac_perturb(st.pl, t0 + min((int)threadIdx.x, k_here - 1), rg, k0, k1, cand);
#pragma unroll
                for (int e = 0; e < 4; e++) shpl[threadIdx.x][e] = cand[e];
            }
            __syncthreads();
            int cnt[K];
            block_count_batch<RS_BLOCK, K>(pts, n, shpl, depth_abs, shb, cnt);
            int accepted = -1, count_acc = 0;                 // the first accepted step of the chain (main.cpp:1701: `>=`)
#pragma unroll
            for (int g = K - 1; g >= 0; g--)
                if (g < k_here && cnt[g] >= st.maximum) { accepted = g; count_acc = cnt[g]; }
            int advance = k_here;
            if (accepted >= 0) {                              // (one plane read with a dynamic index, not K of them selected among)
#pragma unroll
                for (int e = 0; e < 4; e++) st.pl[e] = shpl[accepted][e];
                st.maximum = count_acc;
                advance = accepted + 1;
            }
            t0 += advance;
            __syncthreads();                                  // shpl is rewritten by the next pass
        }
    }
    if (threadIdx.x == 0) {
        region_n4[rg] = make_float4((float)st.pl[0], (float)st.pl[1], (float)st.pl[2], (float)st.pl[3]);
        inlier_ratio[rg] = n > 0 ? (float)st.maximum / (float)n : 0.f;
    }
}

// The chain again, with a region's points split over W workgroups (W CUs) instead of one.  Every workgroup carries the same
// state and makes the same decisions; per pass each counts its share of the points, adds its K counts to the pass's accumulator
// (three rotate: workgroup 0 clears the next one before it arrives at the barrier, a pass after its last reader left) and waits
// at a per-region arrival counter until all W have added.  ~540 passes per region, a few microseconds of barrier each, against
// 33 us of counting per pass on one CU.  The W workgroups of a region must be resident together: the host launches at most
// one 1024-thread workgroup per CU AS A COOPERATIVE LAUNCH (hipLaunchCooperativeKernel: the runtime places the whole grid or
// refuses the launch, also when another context's kernels hold CUs of the device), and as a second line a wait that does not end
// within `poll_limit` polls (~10 ms by default; a pass is microseconds) raises `failed`: every workgroup then leaves and the host
// runs the single-workgroup kernel instead — a spin can never hang the device.  A workgroup that finds `failed` set when it
// starts leaves at once (TSAR_RANSAC_FORCE_FALLBACK=1 pre-sets it: the test of the give-up path).
// sync[slot]: [0] arrival counter, [1 .. 3 K] the three accumulators.
#define RS_SYNC_INTS 64
template <int K>
__global__ __launch_bounds__(RS_BLOCK) void ransac_refine_chain_mw_kernel(const float* __restrict__ pts_all, const int* __restrict__ pts_start,
                                                                          const int* __restrict__ pts_count, const int* __restrict__ region_of_slot,
                                                                          uint32_t k0, uint32_t k1, uint32_t flags, const RansacState* __restrict__ state,
                                                                          const int* __restrict__ cnt_all, float4* __restrict__ region_n4,
                                                                          float* __restrict__ inlier_ratio, int W, int* sync_all, int* failed, int poll_limit) {
    static_assert(1 + 3 * K <= RS_SYNC_INTS, "sync block too small");
    __shared__ int sh[RS_BLOCK / 64], sh2[RS_BLOCK / 64];
    __shared__ int shb[RS_BLOCK / 64][K];
    __shared__ double shpl[K][4];
    __shared__ int tot[K];
    __shared__ int give_up;
    const int slot = blockIdx.x / W, wg = blockIdx.x - slot * W;
    const int rg = region_of_slot[slot];
    const int n = pts_count[slot];
    const float* __restrict__ pts = pts_all + 3 * (size_t)pts_start[slot];
    const int i0 = (int)(((long long)n * wg) / W), i1 = (int)(((long long)n * (wg + 1)) / W);
    int* sync = sync_all + (size_t)slot * RS_SYNC_INTS;
    RansacState st = state[slot];
    bool ok = __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;   // (uniform: nobody waits for a workgroup that left)
    if (n > 0 && ok) {
        {   // the last phase of stage 1 still has to be replayed (every workgroup, identically)
            int best_cnt;
            const int ix = replay_phase(cnt_all + (size_t)slot * RS_PHASE, st.maximum, sh, sh2, &best_cnt);
            if (ix >= 0) {
                ransac_hypothesis(pts, n, rg, 9 * RS_PHASE + ix, k0, k1, flags, st.pl);
                st.maximum = best_cnt;
            }
        }
        const double depth_abs = st.depth_abs_f;
        int t0 = 0, pass = 0;
        while (t0 < 4000 && ok) {
            const int k_here = min(K, 4000 - t0);
            if (threadIdx.x < K) {
                double cand[4];
                ransac_perturb(st.pl, t0 + min((int)threadIdx.x, k_here - 1), rg, k0, k1, cand);
#pragma unroll
                for (int e = 0; e < 4; e++) shpl[threadIdx.x][e] = cand[e];
            }
            __syncthreads();
            int cnt[K];
            block_count_batch<RS_BLOCK, K>(pts + 3 * (size_t)i0, i1 - i0, shpl, depth_abs, shb, cnt);
            int* acc = sync + 1 + (pass % 3) * K;
            int* acc_next = sync + 1 + ((pass + 1) % 3) * K;
            if (threadIdx.x < K) {
                int mine = 0;
#pragma unroll
                for (int g = 0; g < K; g++) mine = ((int)threadIdx.x == g) ? cnt[g] : mine;
                __hip_atomic_fetch_add(&acc[threadIdx.x], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wg == 0) __hip_atomic_store(&acc_next[threadIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                                  // this workgroup's K additions (and the clear) are issued
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const int want = W * (pass + 1);
                int polls = 0, bad = 0;
                while (__hip_atomic_load(&sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(4);
                    if ((++polls & 255) == 0 && __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; break; }
                    if (polls > poll_limit) { __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad = 1; break; }
                }
                give_up = bad;
            }
            __syncthreads();
            if (give_up) { ok = false; break; }
            if (threadIdx.x < K) tot[threadIdx.x] = __hip_atomic_load(&acc[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            int accepted = -1, count_acc = 0;                 // the first accepted step of the chain (main.cpp:1701: `>=`)
#pragma unroll
            for (int g = K - 1; g >= 0; g--) {
                const int c = tot[g];
                if (g < k_here && c >= st.maximum) { accepted = g; count_acc = c; }
            }
            int advance = k_here;
            if (accepted >= 0) {
#pragma unroll
                for (int e = 0; e < 4; e++) st.pl[e] = shpl[accepted][e];
                st.maximum = count_acc;
                advance = accepted + 1;
            }
            t0 += advance;
            pass++;
            __syncthreads();                                  // shpl / tot are rewritten by the next pass
        }
    }
    if (threadIdx.x == 0 && wg == 0 && ok) {
        region_n4[rg] = make_float4((float)st.pl[0], (float)st.pl[1], (float)st.pl[2], (float)st.pl[3]);
        inlier_ratio[rg] = n > 0 ? (float)st.maximum / (float)n : 0.f;
    }
}

extern "C" int tsar_ransac_regions(tsar_ctx* ctx, float* region_planes_out, float* inlier_ratio_out) {
    if (!ctx) return TSAR_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return TSAR_ERR_HIP; }
    if (!ctx->have_views || ctx->n_regions < 1) { ctx->err = "tsar_set_views / tsar_set_regions have not been called"; return TSAR_ERR_STATE; }
    const int nreg = ctx->n_regions;
    const size_t np = (size_t)ctx->w * ctx->h;
    hipStream_t st = ctx->stream;
    const bool trace = ctx->trace_host;
    auto tr0 = std::chrono::steady_clock::now();
    auto TR = [&](const char* what) { if (trace) { hipStreamSynchronize(st); auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[ransac] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - tr0).count()); tr0 = n; } };
    std::vector<float> text(nreg);
    std::vector<int32_t> slot_of_region(nreg, -1), region_of_slot;
    if (hipMemcpy(text.data(), ctx->region_text, (size_t)nreg * 4, hipMemcpyDeviceToHost) != hipSuccess) { ctx->err = "D2H failed"; return TSAR_ERR_HIP; }
    for (int r = 0; r < nreg; r++)
        if (text[r] == -1.0f) { slot_of_region[r] = (int)region_of_slot.size(); region_of_slot.push_back(r); }
    const int nslot = (int)region_of_slot.size();
    TR("text D2H + slots");
    ScratchScope scratch(ctx);           // temporaries come out of the context's arena (tsar_dev.h)
    auto dmalloc = [&](size_t bytes) -> void* { return scratch.alloc(bytes); };
    auto done = [&](int rc, const char* msg) { if (msg) ctx->err = msg; hipStreamSynchronize(st); scratch.release(); return rc; };
    float* d_ratio = (float*)dmalloc((size_t)nreg * 4);
    if (!d_ratio) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
    hipMemsetAsync(d_ratio, 0, (size_t)nreg * 4, st);
    if (nslot > 0) {
        int32_t* d_slot_of_region = (int32_t*)dmalloc((size_t)nreg * 4);
        int32_t* d_region_of_slot = (int32_t*)dmalloc((size_t)nslot * 4);
        uint32_t* d_pix = (uint32_t*)dmalloc(np * 4);
        uint32_t* d_nsel = (uint32_t*)dmalloc(4);
        int* d_counts = (int*)dmalloc((size_t)nslot * 4);
        if (!d_slot_of_region || !d_region_of_slot || !d_pix || !d_nsel || !d_counts) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
        hipMemcpyAsync(d_slot_of_region, slot_of_region.data(), (size_t)nreg * 4, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_region_of_slot, region_of_slot.data(), (size_t)nslot * 4, hipMemcpyHostToDevice, st);
        hipMemsetAsync(d_counts, 0, (size_t)nslot * 4, st);
        TR("allocs + uploads");
        // (1) raster-order list of reliable pixels inside textureless regions (main.cpp:1527-1536)
        uint8_t* d_flag = (uint8_t*)dmalloc(np);
        if (!d_flag) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
        size_t tmp_bytes = 0;
        rocprim::counting_iterator<uint32_t> first(0);
        if (rocprim::select(nullptr, tmp_bytes, first, d_flag, d_pix, d_nsel, np, st) != hipSuccess) return done(TSAR_ERR_HIP, "rocprim::select sizing failed");
        void* d_tmp = dmalloc(tmp_bytes);
        if (!d_tmp) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
        {
            ScopedKernelTimer tm(ctx, "ransac_select");
            hipLaunchKernelGGL(ransac_flag_kernel, dim3(2048), dim3(256), 0, st, ctx->canny, ctx->scale, d_slot_of_region, np, d_flag);
            if (rocprim::select(d_tmp, tmp_bytes, first, d_flag, d_pix, d_nsel, np, st) != hipSuccess) return done(TSAR_ERR_HIP, "rocprim::select failed");
        }
        uint32_t nsel = 0;
        hipMemcpyAsync(&nsel, d_nsel, 4, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) return done(TSAR_ERR_HIP, "select failed");
        TR("select");
        std::vector<int> counts(nslot, 0), slot_start(nslot, 0), pts_start(nslot, 0), pts_count(nslot, 0);
        float* d_pts = nullptr;
        int *d_slot_start = (int*)dmalloc((size_t)nslot * 4), *d_pts_start = (int*)dmalloc((size_t)nslot * 4), *d_pts_count = (int*)dmalloc((size_t)nslot * 4);
        if (!d_slot_start || !d_pts_start || !d_pts_count) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
        if (nsel > 0) {
            uint32_t *d_keys = (uint32_t*)dmalloc((size_t)nsel * 4), *d_keys2 = (uint32_t*)dmalloc((size_t)nsel * 4), *d_pix2 = (uint32_t*)dmalloc((size_t)nsel * 4);
            if (!d_keys || !d_keys2 || !d_pix2) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
            hipLaunchKernelGGL(ransac_keys_kernel, dim3((nsel + 255) / 256), dim3(256), 0, st, d_pix, (int)nsel, ctx->canny, d_slot_of_region, d_keys, d_counts);
            // (2) stable sort by region keeps raster order inside each region
            int bits = 1;
            while ((1 << bits) < nslot) bits++;
            size_t sort_bytes = 0;
            if (rocprim::radix_sort_pairs(nullptr, sort_bytes, d_keys, d_keys2, d_pix, d_pix2, nsel, 0, bits, st) != hipSuccess) return done(TSAR_ERR_HIP, "radix sort sizing failed");
            void* d_sort_tmp = dmalloc(sort_bytes);
            if (!d_sort_tmp) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
            {
                ScopedKernelTimer tm(ctx, "ransac_sort");
                if (rocprim::radix_sort_pairs(d_sort_tmp, sort_bytes, d_keys, d_keys2, d_pix, d_pix2, nsel, 0, bits, st) != hipSuccess) return done(TSAR_ERR_HIP, "radix sort failed");
            }
            hipMemcpyAsync(counts.data(), d_counts, (size_t)nslot * 4, hipMemcpyDeviceToHost, st);
            if (hipStreamSynchronize(st) != hipSuccess) return done(TSAR_ERR_HIP, "sort failed");
            TR("keys + sort + counts D2H");
            int acc = 0, pacc = 0;
            for (int s = 0; s < nslot; s++) {
                slot_start[s] = acc; acc += counts[s];
                pts_count[s] = counts[s] > 50000 ? RS_MAXPTS : counts[s];      // main.cpp:1540-1549
                pts_start[s] = pacc; pacc += pts_count[s];
            }
            d_pts = (float*)dmalloc((size_t)(pacc > 0 ? pacc : 1) * 12);
            if (!d_pts) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
            hipMemcpyAsync(d_slot_start, slot_start.data(), (size_t)nslot * 4, hipMemcpyHostToDevice, st);
            hipMemcpyAsync(d_counts, counts.data(), (size_t)nslot * 4, hipMemcpyHostToDevice, st);
            hipMemcpyAsync(d_pts_start, pts_start.data(), (size_t)nslot * 4, hipMemcpyHostToDevice, st);
            {
                ScopedKernelTimer tm(ctx, "ransac_points");
                hipLaunchKernelGGL(ransac_points_kernel, dim3((nsel + 255) / 256), dim3(256), 0, st, ctx->dscene, d_pix2, d_keys2, (int)nsel, ctx->depth, d_slot_start,
                                   d_counts, d_pts_start, d_pts);
            }
        } else {
            d_pts = (float*)dmalloc(12);
            hipMemcpyAsync(d_pts_start, pts_start.data(), (size_t)nslot * 4, hipMemcpyHostToDevice, st);
        }
        hipMemcpyAsync(d_pts_count, pts_count.data(), (size_t)nslot * 4, hipMemcpyHostToDevice, st);
        TR("points");
        RansacState* d_state = (RansacState*)dmalloc((size_t)nslot * sizeof(RansacState));
        int* d_cnt = (int*)dmalloc((size_t)nslot * RS_PHASE * sizeof(int));
        if (!d_state || !d_cnt) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
        // stage 2, measured on six ~50 000-point regions: the history tree with lookahead 1 / 2 / 3 -> 25.9 / 26.4 / 33.7 ms (the passes
        // are bound by the CU's FP64 rate, so the extra planes of a tree cost what the saved passes return); the speculative chain of
        // 8 steps 17.5 ms; the chain on 8 CUs per region 5.3 ms (the default).
        const int lookahead = ctx->ransac_lookahead;
        // default: the speculative chain (ransac_refine_chain_kernel), K = 8; TSAR_RANSAC_CHAIN=4|8|16 picks the length,
        // TSAR_RANSAC_LOOKAHEAD=1|2|3 the history-tree kernel instead
        const int chain = ctx->ransac_chain;
        {
            ScopedKernelTimer tm(ctx, "ransac_fit");
            for (int phase = 0; phase < 10; phase++) {
                hipLaunchKernelGGL(ransac_adapt_kernel, dim3(nslot), dim3(RS_BLOCK), 0, st, d_pts, d_pts_start, d_pts_count, d_region_of_slot, ctx->region_size,
                                   ctx->hscene.seed_lo, ctx->hscene.seed_hi, ctx->hscene.flags, d_state, d_cnt, phase);
                hipLaunchKernelGGL(ransac_count_kernel, dim3(RS_BATCHES, nslot), dim3(RS_COUNT_BLOCK), 0, st, d_pts, d_pts_start, d_pts_count, d_region_of_slot,
                                   ctx->hscene.seed_lo, ctx->hscene.seed_hi, ctx->hscene.flags, d_state, d_cnt, phase);
            }
            auto refine = lookahead == 1 ? ransac_refine_kernel<1> : (lookahead == 3 ? ransac_refine_kernel<3> : ransac_refine_kernel<2>);
            if (lookahead < 1) refine = chain == 4 ? ransac_refine_chain_kernel<4> : (chain == 16 ? ransac_refine_chain_kernel<16> : ransac_refine_chain_kernel<8>);
            // W workgroups (CUs) per region in stage 2 (ransac_refine_chain_mw_kernel): 8 by default (measured on five ~50 000-point
            // regions: W = 1 / 2 / 4 / 8 / 16 -> 17.5 / 10.8 / 6.9 / 5.3 / 4.8 ms for the whole fit), at most one workgroup per CU so
            // that all of them are resident together; TSAR_RANSAC_WGS=W overrides, 1 = the single-workgroup kernel
            int wgs = ctx->ransac_wgs;
            int n_cu = 0;
            if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) n_cu = 0;
            if (wgs > n_cu / (nslot > 0 ? nslot : 1)) wgs = n_cu / (nslot > 0 ? nslot : 1);
            bool fitted = false;
            if (lookahead < 1 && wgs >= 2) {
                int* d_sync = (int*)dmalloc((size_t)nslot * RS_SYNC_INTS * 4 + 4);
                if (!d_sync) return done(TSAR_ERR_NOMEM, "hipMalloc failed");
                int* d_failed = d_sync + (size_t)nslot * RS_SYNC_INTS;
                hipMemsetAsync(d_sync, 0, (size_t)nslot * RS_SYNC_INTS * 4 + 4, st);
                // diagnostics: TSAR_RANSAC_FORCE_FALLBACK=1 pre-sets `failed` (the give-up path: every workgroup leaves, the
                // single-workgroup kernel below produces the result); TSAR_RANSAC_POLL_LIMIT=n bounds a wait (default 2^15 polls
                // of ~0.3 us: ~10 ms, three orders of magnitude above a pass)
                const bool force_fallback = ctx->ransac_force_fallback;
                int poll_limit = ctx->ransac_poll_limit;
                if (force_fallback) { const int one = 1; hipMemcpyAsync(d_failed, &one, 4, hipMemcpyHostToDevice, st); }
                const void* mw = chain == 4 ? (const void*)ransac_refine_chain_mw_kernel<4> : (chain == 16 ? (const void*)ransac_refine_chain_mw_kernel<16> : (const void*)ransac_refine_chain_mw_kernel<8>);
                const float* a_pts = d_pts; const int *a_ps = d_pts_start, *a_pc = d_pts_count, *a_rs = d_region_of_slot, *a_cnt = d_cnt;
                uint32_t a_k0 = ctx->hscene.seed_lo, a_k1 = ctx->hscene.seed_hi, a_fl = ctx->hscene.flags;
                const RansacState* a_state = d_state;
                float4* a_n4 = ctx->region_n4;
                float* a_ratio = d_ratio;
                int a_w = wgs;
                void* args[] = {&a_pts, &a_ps, &a_pc, &a_rs, &a_k0, &a_k1, &a_fl, &a_state, &a_cnt, &a_n4, &a_ratio, &a_w, &d_sync, &d_failed, &poll_limit};
                int coop = 0;
                if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, ctx->device) != hipSuccess) coop = 0;
                if (!ctx->ransac_cooperative) coop = 0;
                // One cooperative grid at a time per process: the contexts of one process (tsar_gipuma --workers=2, ranks' threads) would
                // otherwise compete for the CUs their grids must hold together — and two threads inside hipLaunchCooperativeKernel at
                // once leave the runtime (ROCm 7.2) in a state that crashes at process exit (seen in the two-contexts test).  The lock
                // is held until the grid has drained.
                static std::mutex coop_mutex;
                std::unique_lock<std::mutex> coop_lock(coop_mutex, std::defer_lock);
                if (coop) coop_lock.lock();
                hipError_t le = coop ? hipLaunchCooperativeKernel(mw, dim3(nslot * wgs), dim3(RS_BLOCK), args, 0, st)
                                     : hipLaunchKernel(mw, dim3(nslot * wgs), dim3(RS_BLOCK), args, 0, st);
                if (le == hipSuccess) {
                    int h_failed = 1;
                    hipMemcpyAsync(&h_failed, d_failed, 4, hipMemcpyDeviceToHost, st);
                    if (hipStreamSynchronize(st) != hipSuccess) return done(TSAR_ERR_HIP, "ransac kernel failed");
                    fitted = h_failed == 0;                   // else: the workgroups were not resident together; one per region below
                } else {
                    (void)hipGetLastError();                  // the grid could not be placed as a whole (or no cooperative launches here): one workgroup per region below
                }
                if (coop) coop_lock.unlock();
                if (trace) fprintf(stderr, "[ransac] stage 2 on %d workgroups per region: %s%s\n", wgs, coop ? "cooperative launch" : "plain launch", fitted ? "" : " -> fallback to one workgroup per region");
            }
            if (!fitted)
                hipLaunchKernelGGL(refine, dim3(nslot), dim3(RS_BLOCK), 0, st, d_pts, d_pts_start, d_pts_count, d_region_of_slot, ctx->hscene.seed_lo,
                                   ctx->hscene.seed_hi, ctx->hscene.flags, d_state, d_cnt, ctx->region_n4, d_ratio);
        }
        if (hipGetLastError() != hipSuccess) return done(TSAR_ERR_HIP, "ransac launch failed");
        TR("fit");
    }
    if (region_planes_out) hipMemcpyAsync(region_planes_out, ctx->region_n4, (size_t)nreg * 16, hipMemcpyDeviceToHost, st);
    if (inlier_ratio_out) hipMemcpyAsync(inlier_ratio_out, d_ratio, (size_t)nreg * 4, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) return done(TSAR_ERR_HIP, "ransac kernel failed");
    TR("outputs D2H");
    return done(TSAR_OK, nullptr);
}
