// pm_sweep_lds.hip — OPT-IN (TSAR_LDS_SWEEP=1) form of the red/black half-iteration (propagation +
// refinement fused) for the configuration every reference script uses: 8-bit imagery, --blocksize=11,
// n_best <= 2, at most LDS_VIEWS selected source views.  Same arithmetic and results as pm_sweep.hip (the
// default form): the whole GPU parity suite passes bit for bit with it.  What differs is where the bytes
// come from.
//
// STATUS (round 1, profiles/r01/README.md): not yet faster.  Measured on MI355X at 6048x4032 / 10 views:
// default form 65 ms per launch; this form 112 ms with the patch test (48 % of wave evaluations qualify
// over the first 4 iterations), and 72 ms when EVERY evaluation is forced onto the LDS path (timing
// experiment, wrong results).  So removing the L1 bottleneck alone does not pay: at 250 VGPRs this kernel
// runs 2 waves/SIMD and is bound by per-wave issue/latency, not by LDS or L1.  Capping it at 168 VGPRs
// (3 waves) spills ~100 registers.  Kept as the starting point for the next round (leaner tap body,
// separable weight table to free registers, robust patch placement).
//
// Why (profiles/r01): in the generic kernel every source tap is a per-lane 4-byte gather.  The vector L1
// retires about one 32-byte sector access per clock per CU and a 64-lane gather costs ~30 of them even
// when neighbouring pixels hold nearly the same plane, so the kernel is L1-access bound (TCP 0.8-0.9
// accesses/clk/CU) with the VALU ~45 % busy.
//
// What:
//  * The source-view patches a workgroup's 32x16 region lands in (56x36 texels each, stored as horizontal
//    texel pairs, 4.2 KiB per view) are staged into LDS once per launch with coalesced loads.
//  * A hypothesis x view evaluation first maps the four corner taps.  If every lane of the wave stays
//    inside its patch (projective maps keep the window convex), all 36 taps are served by ds_read_u16
//    pairs with NO per-tap clamp, bounds test or fallback code: ~30 VALU instructions per tap instead of
//    ~42.  Otherwise (random planes of the first sweeps, the two wide refinement steps, borders) the wave
//    takes the global-gather path of the generic kernel.  Both paths read the same texels, so the choice
//    (like the patch placement) cannot change a result.
//  * The 36 bilateral weights live in registers (the tap loops are fully unrolled), not in 36 KiB of LDS:
//    LDS per workgroup is 52 KiB -> three workgroups per CU (3 waves/SIMD).
//  * One rolled loop walks the 8 propagation arms and then the refinement steps, so the unrolled cost
//    routine exists once in the code object.
#include "pm_core.h"

#define SWEEP_RH 16
#define WIN_W 56
#define WIN_H 36
#define WIN_P 57                 // odd pitch: rows fall on different LDS banks
#define WIN_ROWS (WIN_H + 1)     // one extra row of pairs so that row y+1 exists for the last window row
#define WIN_ENTRIES (WIN_P * WIN_ROWS)
#define LDS_VIEWS 10             // source views whose patches are resident together
#define NTAP 36

// 8-arm adaptive candidate selection, gipuma.cu:874-1042 (same routine as pm_sweep.hip; device code is
// per translation unit).  out[a] = neighbour pixel index | (same colour ? 1 << 31 : 0), or -1.
DEVFN void select_arms(const DevScene* __restrict__ sc, const float* __restrict__ c_same, const float* __restrict__ c_other, int x, int y,
                       int* out, int stride) {
    const int col = sc->w, row = sc->h;
    const int p = y * col + x;
    const bool fix_seed = sc->flags & TSAR_FLAG_FIX_DOWN_FAR_SEED, fix_cmp = sc->flags & TSAR_FLAG_FIX_RIGHT_FAR_CMP;
    const int SAME = (int)0x80000000u;
    float cmin;
    int cp;
#pragma unroll
    for (int k = 0; k < 8; k++) out[k * stride] = -1;
    if (y > 2) {
        cp = p - 3 * col; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y > 2 + 2 * i) { const int q = p - (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        out[0] = cp;
    }
    if (y < row - 3) {
        cp = p + 3 * col;
        cmin = (fix_seed || y <= 2) ? c_other[cp] : c_other[p - 3 * col];   // gipuma.cu:906 seeds with c[up_far]
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y < row - 3 - 2 * i) { const int q = p + (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        out[1 * stride] = cp;
    }
    if (x > 2) {
        cp = p - 3; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (x > 2 + 2 * i) { const int q = p - 3 - 2 * i; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        out[2 * stride] = cp;
    }
    if (x < col - 3) {
        cp = p + 3; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (x < col - 3 - 2 * i) {
                const int q = p + 3 + 2 * i;
                const float v = c_other[q];
                const bool take = fix_cmp ? (v < cmin) : (cmin < v);             // gipuma.cu:943 is inverted
                if (take) { cmin = v; cp = q; }
            }
        out[3 * stride] = cp;
    }
    if (y > 0) {
        cp = p - col; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y > 1 + i && x > i) { const int q = p - (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
            if (y > 1 + i && x < col - 1 - i) { const int q = p - (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
        }
        out[4 * stride] = cp;
    }
    if (y < row - 1) {
        cp = p + col; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y < row - 2 - i && x > i) { const int q = p + (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
            if (y < row - 2 - i && x < col - 1 - i) { const int q = p + (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
        }
        out[5 * stride] = cp;
    }
    if (x > 0) {
        cp = p - 1; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x > 1 + i && y > i) { const int q = p - (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
            if (x > 1 + i && y < row - 1 - i) { const int q = p - (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
        }
        out[6 * stride] = cp;
    }
    if (x < col - 1) {
        cp = p + 1; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x < col - 2 - i && y > i) { const int q = p + (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
            if (x < col - 2 - i && y < row - 1 - i) { const int q = p + (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q | SAME; } }
        }
        out[7 * stride] = cp;
    }
}

DEVFN bool same_plane_bits(const float4& a, const float4& b) {
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
           __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

// (u, v) of reference tap (xf, yf) under H; the perspective divide is the only place fast and strict differ
template <bool STRICT>
DEVFN void warp(const float* H, float bx, float by, float bz, float yj, float& u, float& v, float& Z) {
    const float X = fma_(H[1], yj, bx), Y = fma_(H[4], yj, by);
    Z = fma_(H[7], yj, bz);
    if (STRICT) {
        u = X / Z;
        v = Y / Z;
    } else {
        const float rz = __builtin_amdgcn_rcpf(Z);
        u = X * rz;
        v = Y * rz;
    }
}

// pmCost (gipuma.cu:229-298) for one source view.  wt[] = the pixel's 36 bilateral weights (registers),
// tile = reference window (LDS), win = this view's staged patch with origin (ox, oy) in quad coordinates.
template <bool STRICT>
DEVFN float view_cost_lds(const DevScene* __restrict__ sc, const DevView& vw, const unsigned char* tile, int own, const float (&wt)[NTAP],
                          const PixelRef& pr, int x, int y, const float4& n4, const unsigned short* win, int ox, int oy, unsigned long long* dbg) {
    constexpr int tw = PM_RW + 10;
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    float H[9];
    plane_homography(sc->ref, vw, n4, H);
    // ---- do all 36 taps of every lane stay inside the patch?  corners decide (convexity) --------------
    bool lane_in = true;
    {
        const float fw = (float)w, fh = (float)h;
#pragma unroll
        for (int ci = -5; ci <= 5; ci += 10) {
            const float xi = (float)(x + ci);
            const float bx = fma_(H[0], xi, H[2]), by = fma_(H[3], xi, H[5]), bz = fma_(H[6], xi, H[8]);
#pragma unroll
            for (int cj = -5; cj <= 5; cj += 10) {
                float u, v, Z;
                warp<STRICT>(H, bx, by, bz, (float)(y + cj), u, v, Z);
                // one texel of slack for rounding of the interior taps; inside the image so that the clamp is a no-op
                const int wx = (int)floorf(fminf(fmaxf(u, -1.0f), fw)) + 1 - ox, wy = (int)floorf(fminf(fmaxf(v, -1.0f), fh)) + 1 - oy;
                lane_in = lane_in && Z > 0.0f && u > -1.0f && u < fw && v > -1.0f && v < fh && wx >= 1 && wx <= WIN_W - 2 && wy >= 1 && wy <= WIN_H - 2;
            }
        }
    }
    float sum_src = 0.f, sum_src_src = 0.f, sum_ref_src = 0.f;
    const bool fast = __all(lane_in);
    if (dbg) {   // TSAR_DEBUG_COUNTERS=1: [0] wave evaluations on the LDS path, [1] on the gather path, [2] lanes that vetoed
        const unsigned long long act = __ballot(1);
        if ((int)(__ffsll((long long)act) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&dbg[fast ? 0 : 1], 1ull);
        if (!lane_in) atomicAdd(&dbg[2], 1ull);
    }
    if (fast) {
        // ---- fast path: every tap from LDS, no clamp, no bounds test ------------------------------------
        const int org = __mul24(oy, WIN_P) + ox;
#pragma unroll
        for (int ii = 0; ii < 6; ii++) {
            const int i = 2 * ii - 5;
            const float xi = (float)(x + i);
            const float bx = fma_(H[0], xi, H[2]), by = fma_(H[3], xi, H[5]), bz = fma_(H[6], xi, H[8]);
#pragma unroll
            for (int jj = 0; jj < 6; jj++) {
                const int j = 2 * jj - 5;
                float u, v, Z;
                warp<STRICT>(H, bx, by, bz, (float)(y + j), u, v, Z);
                const float fu = floorf(u), fv = floorf(v);
                const float ax = u - fu, ay = v - fv;
                const int e = __mul24((int)fv + 1, WIN_P) + (int)fu + 1 - org;
                const uint32_t a = win[e], b = win[e + WIN_P];       // (T(x,y), T(x+1,y)) and the pair one row below
                const float t00 = (float)(a & 0xffu), t10 = (float)(a >> 8), t01 = (float)(b & 0xffu), t11 = (float)(b >> 8);
                const float top = fma_(ax, t10 - t00, t00);
                const float bot = fma_(ax, t11 - t01, t01);
                const float s = fma_(ay, bot - top, top);
                const float r = (float)tile[own + j * tw + i];
                const float wr = wt[ii * 6 + jj] * r, ws = wt[ii * 6 + jj] * s;
                sum_src += ws;
                sum_src_src = fma_(ws, s, sum_src_src);
                sum_ref_src = fma_(wr, s, sum_ref_src);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one column's reads/temporaries live at a time (register pressure)
        }
    } else {
        // ---- general path: per-lane gathers from the quad image in HBM/L2 (as pm_sweep.hip) -------------
#pragma unroll
        for (int ii = 0; ii < 6; ii++) {
            const int i = 2 * ii - 5;
            const float xi = (float)(x + i);
            const float bx = fma_(H[0], xi, H[2]), by = fma_(H[3], xi, H[5]), bz = fma_(H[6], xi, H[8]);
#pragma unroll
            for (int jj = 0; jj < 6; jj++) {
                const int j = 2 * jj - 5;
                float u, v, Z;
                warp<STRICT>(H, bx, by, bz, (float)(y + j), u, v, Z);
                const float s = sample_bilinear<true>(vw, w, h, qp, u, v);
                const float r = (float)tile[own + j * tw + i];
                const float wr = wt[ii * 6 + jj] * r, ws = wt[ii * 6 + jj] * s;
                sum_src += ws;
                sum_src_src = fma_(ws, s, sum_src_src);
                sum_ref_src = fma_(wr, s, sum_ref_src);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    sum_src *= pr.inv_wsum;
    sum_src_src *= pr.inv_wsum;
    sum_ref_src *= pr.inv_wsum;
    const float var_src = sum_src_src - sum_src * sum_src;
    if (var_src < 1e-5f) return TSAR_MAXCOST;
    const float covar = sum_ref_src - pr.mean_ref * sum_src;
    const float vrs = sqrtf(pr.var_ref * var_src);
    return fmaxf(0.0f, fminf(TSAR_MAXCOST, 1.0f - covar / vrs));
}

template <bool STRICT>
__global__ __launch_bounds__(PM_BLOCK) void pm_sweep_lds_kernel(const DevScene* __restrict__ sc, int colour, const float* __restrict__ c_same,
                                                                const float4* __restrict__ n_same, const float* __restrict__ c_other,
                                                                const float4* __restrict__ n_other, float* c_out, float4* n_out,
                                                                float* __restrict__ ratio_out, int32_t* __restrict__ beview_out,
                                                                uint32_t stream_id, int do_prop, int do_refine, int tiles_x, int n_tiles,
                                                                int cost_consistent, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int hr = 5, vr = 5;
    constexpr int tw = PM_RW + 2 * hr, th = SWEEP_RH + 2 * vr;
    unsigned char* tile = lds_raw;
    int* arms = (int*)(lds_raw + tile_bytes<true>(tw, th)) + threadIdx.x;                 // [8][PM_BLOCK]
    unsigned short* wins = (unsigned short*)(lds_raw + tile_bytes<true>(tw, th) + sizeof(int) * 8 * PM_BLOCK);
    __shared__ int box[4 * LDS_VIEWS];
    __shared__ int org[2 * LDS_VIEWS];

    const int t = xcd_tile(blockIdx.x, n_tiles);
    const int ty0 = (t / tiles_x) * SWEEP_RH, tx0 = (t % tiles_x) * PM_RW;
    const int n_sel = sc->n_sel;
    stage_ref_tile<SWEEP_RH, unsigned char>(sc, tile, tx0, ty0, hr, vr);
    if (threadIdx.x < n_sel) { box[4 * threadIdx.x] = 0x7fffffff; box[4 * threadIdx.x + 1] = 0x7fffffff; box[4 * threadIdx.x + 2] = -0x7fffffff; box[4 * threadIdx.x + 3] = -0x7fffffff; }
    __syncthreads();

    const int ly = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int y = ty0 + ly;
    const int lx = 2 * k + ((colour + y) & 1);       // (x + y) & 1 == colour; gipuma.cu:1099-1103 / :1121-1125
    const int x = tx0 + lx;
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    const bool in_image = x < w && y < h;
    const int p = in_image ? y * w + x : 0;
    const int own = (ly + vr) * tw + lx + hr;
    const DevRef& rf = sc->ref;

    // ---- per-pixel hoist: weights (registers), reference moments (gipuma.cu:247-277) -------------------
    float wt[NTAP];
    PixelRef pr;
    {
        const float cen = (float)tile[own];
        float sum_ref = 0.f, sum_ref_ref = 0.f, wsum = 0.f;
#pragma unroll
        for (int ii = 0; ii < 6; ii++) {
#pragma unroll
            for (int jj = 0; jj < 6; jj++) {
                const int i = 2 * ii - 5, j = 2 * jj - 5;
                const float r = (float)tile[own + j * tw + i];
                const float sd = sqrtf((float)(i * i + j * j));
                const float cd = fabsf(r - cen);
                const float wv = tsar_expf(-sd / 50.0f - cd / 18.0f);
                wt[ii * 6 + jj] = wv;
                const float wr = wv * r;
                sum_ref += wr;
                sum_ref_ref = fma_(wr, r, sum_ref_ref);
                wsum += wv;
            }
        }
        pr.inv_wsum = 1.0f / wsum;
        sum_ref *= pr.inv_wsum;
        sum_ref_ref *= pr.inv_wsum;
        pr.mean_ref = sum_ref;
        pr.var_ref = sum_ref_ref - sum_ref * sum_ref;
        pr.textured = !(pr.var_ref < 1e-5f);
    }
    const bool active = in_image && pr.textured;

    float cost_now = 0.f;
    float4 n_now = make_float4(0.f, 0.f, -1.f, 1.f);
    if (in_image) { cost_now = c_same[p]; n_now = n_same[p]; }

    // ---- patch placement (a hint only) and the propagation arms -----------------------------------------
    if (active) {
        for (int i = 0; i < n_sel; i++) {
            float H[9];
            plane_homography(rf, sc->view[sc->sel[i]], n_now, H);
            const float xf = (float)x, yf = (float)y;
            const float rz = 1.0f / fma_(H[7], yf, fma_(H[6], xf, H[8]));
            const float u = fminf(fmaxf(fma_(H[1], yf, fma_(H[0], xf, H[2])) * rz, -1.0f), (float)w);
            const float v = fminf(fmaxf(fma_(H[4], yf, fma_(H[3], xf, H[5])) * rz, -1.0f), (float)h);
            const int iu = (int)floorf(u) + 1, iv = (int)floorf(v) + 1;
            atomicMin(&box[4 * i], iu); atomicMin(&box[4 * i + 1], iv);
            atomicMax(&box[4 * i + 2], iu); atomicMax(&box[4 * i + 3], iv);
        }
        if (do_prop) select_arms(sc, c_same, c_other, x, y, arms, PM_BLOCK);
    }
    __syncthreads();
    if (threadIdx.x < n_sel) {
        const int i = threadIdx.x;
        const int x0 = box[4 * i], y0 = box[4 * i + 1], x1 = box[4 * i + 2], y1 = box[4 * i + 3];
        int ox = 0, oy = 0;
        if (x1 >= x0) { ox = (x0 + x1) / 2 - WIN_W / 2; oy = (y0 + y1) / 2 - WIN_H / 2; }
        org[2 * i] = max(0, min(ox, w + 2 - WIN_W));
        org[2 * i + 1] = max(0, min(oy, h + 2 - WIN_H));
    }
    __syncthreads();
    for (int s = 0; s < n_sel; s++) {
        const DevView& vw = sc->view[sc->sel[s]];
        const int ox = org[2 * s], oy = org[2 * s + 1];
        const global_u32_ptr src = (global_u32_ptr)vw.quad;
        unsigned short* dst = wins + s * WIN_ENTRIES;
        for (int e = threadIdx.x; e < WIN_W * WIN_ROWS; e += PM_BLOCK) {
            const int wy = e / WIN_W, wx = e - wy * WIN_W;
            // rows 0..WIN_H-1: low half (row y pair) of the quad at that row; extra row: high half of the last quad row
            const uint32_t q = src[(uint32_t)((oy + min(wy, WIN_H - 1)) * qp + ox + wx)];
            dst[wy * WIN_P + wx] = (unsigned short)(wy < WIN_H ? (q & 0xffffu) : (q >> 16));
        }
    }
    __syncthreads();
    if (!active) {                                   // no barrier below this point
        if (in_image) { c_out[p] = cost_now; n_out[p] = n_now; }
        return;
    }

    // ---- one rolled loop over the 8 propagation arms, then the refinement steps -------------------------
    bool wrote = false;
    float ratio_w = 0.f;
    int beview_w = 0;
    float depth_now = plane_depth(rf, n_now, x, y);
    const float4 n_first = n_now;
    float vv[3];
    view_vector(rf, x, y, vv);
    float deltaN = 1.0f, deltaZ = sc->max_disp / 2.0f;
    const float fb = rf.f * rf.baseline;
    const int h0 = do_prop ? 0 : 8, h1 = do_refine ? 8 + sc->refine_steps : 8;
#pragma unroll 1
    for (int hyp = h0; hyp < h1; hyp++) {
        float4 n_t = n_now;
        float depth_t = depth_now;
        bool valid = false;
        if (hyp < 8) {                               // spatialPropagation_cu gipuma.cu:524-566
            const int a = arms[hyp * PM_BLOCK];
            if (a != -1) {
                const int idx = a & 0x7fffffff;
                n_t = (a < 0) ? n_same[idx] : n_other[idx];
                depth_t = plane_depth(rf, n_t, x, y);
                // out-of-range planes are never accepted (:553); a plane bit-identical to the one this pixel holds or
                // held at launch start re-scores to a cost that is not lower (c[p] is the score of norm4[p])
                valid = depth_t >= rf.depthMin && depth_t <= rf.depthMax &&
                        !(cost_consistent && (same_plane_bits(n_t, n_now) || same_plane_bits(n_t, n_first)));
            }
        } else {                                     // planeRefinement_cu gipuma.cu:621-676, getRndDispAndUnitVector_cu :582-619
            const Rand4 rn = philox_uniform4((uint32_t)p, stream_id, (uint32_t)(hyp - 8), sc->seed_lo, sc->seed_hi);
            const float disp = fb / depth_now;
            const float minDelta = -fminf(deltaZ, sc->min_disp + disp);      // "+" as written, :601
            const float maxDelta = fminf(deltaZ, sc->max_disp - disp);
            const float dz = between(rn.u[0], minDelta, maxDelta);
            const float dispOut = fminf(fmaxf(disp + dz, sc->min_disp), sc->max_disp);
            depth_t = fb / dispOut;
            float nt[3];
            nt[0] = n_now.x + between(rn.u[1], -deltaN, deltaN);
            nt[1] = n_now.y + between(rn.u[2], -deltaN, deltaN);
            nt[2] = n_now.z + between(rn.u[3], -deltaN, deltaN);
            const float inv = 1.0f / sqrtf(dot3(nt, nt));
            nt[0] *= inv; nt[1] *= inv; nt[2] *= inv;
            if (dot3(nt, vv) > 0.0f) { nt[0] = -nt[0]; nt[1] = -nt[1]; nt[2] = -nt[2]; }
            n_t.x = nt[0]; n_t.y = nt[1]; n_t.z = nt[2];
            n_t.w = plane_offset(rf, nt, x, y, depth_t);
            deltaN = deltaN / 4.0f;
            deltaZ = deltaZ / 10.0f;
            valid = true;
        }
        if (valid) {
            // pmCostMultiview_cu gipuma.cu:455-518, best-2 kept in registers
            float b0 = __builtin_inff(), b1 = __builtin_inff(), cmin = __builtin_inff();
            int nvalid = 0, bv = -1;
#pragma unroll 1
            for (int i = 0; i < n_sel; i++) {
                const int vi = sc->sel[i];
                float c = view_cost_lds<STRICT>(sc, sc->view[vi], tile, own, wt, pr, x, y, n_t, wins + i * WIN_ENTRIES, org[2 * i], org[2 * i + 1], dbg);
                if (c < TSAR_MAXCOST) nvalid++; else c = TSAR_MAXCOST;
                if (c <= cmin) { cmin = c; bv = vi; }
                const float lo = fminf(b0, c), hi = fmaxf(b0, c);
                b0 = lo;
                b1 = fminf(b1, hi);
            }
            int nb = nvalid;
            if (sc->cost_comb == TSAR_COMB_BEST_N) nb = min(nb, sc->n_best);
            if (nb > 0) {
                float cost = 0.f + b0;
                if (nb > 1) cost += b1;
                cost = cost / (float)nb;
                if (cost < cost_now) {
                    cost_now = cost; n_now = n_t; depth_now = depth_t;
                    ratio_w = n_sel >= 2 ? b0 / b1 : 0.f;
                    beview_w = bv;
                    wrote = true;
                }
            }
        }
    }
    c_out[p] = cost_now;
    n_out[p] = n_now;
    if (wrote) { ratio_out[p] = ratio_w; beview_out[p] = beview_w; }
}

// launches the LDS form when it applies (sets *launched), otherwise leaves the work to the generic form
int launch_pm_sweep_lds(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out, uint32_t stream_id,
                        int do_prop, int do_refine, int* launched) {
    const DevScene& hs = ctx->hscene;
    *launched = 0;
    const int need = hs.cost_comb == TSAR_COMB_BEST_N ? (hs.n_best < hs.n_sel ? hs.n_best : hs.n_sel) : hs.n_sel;
    if (!(hs.use_quad && hs.hrad == 5 && hs.vrad == 5 && need <= 2 && hs.n_sel <= LDS_VIEWS && hs.w + 2 >= WIN_W && hs.h + 2 >= WIN_H)) return TSAR_OK;
    const int tiles_x = (hs.w + PM_RW - 1) / PM_RW, tiles_y = (hs.h + SWEEP_RH - 1) / SWEEP_RH;
    const int n_tiles = tiles_x * tiles_y;
    const size_t lds = tile_bytes<true>(PM_RW + 10, SWEEP_RH + 10) + sizeof(int) * 8 * PM_BLOCK + sizeof(unsigned short) * WIN_ENTRIES * hs.n_sel;
    const bool strict = hs.flags & TSAR_FLAG_STRICT_DIV;
    {
        ScopedKernelTimer tm(ctx, "pm_sweep");
        if (strict)
            hipLaunchKernelGGL(pm_sweep_lds_kernel<true>, dim3(n_tiles), dim3(PM_BLOCK), lds, ctx->stream, ctx->dscene, colour, same_in.c, same_in.n4, other.c,
                               other.n4, same_out.c, same_out.n4, ctx->ratio, ctx->beview, stream_id, do_prop, do_refine, tiles_x, n_tiles, ctx->cost_consistent ? 1 : 0, ctx->dbg);
        else
            hipLaunchKernelGGL(pm_sweep_lds_kernel<false>, dim3(n_tiles), dim3(PM_BLOCK), lds, ctx->stream, ctx->dscene, colour, same_in.c, same_in.n4, other.c,
                               other.n4, same_out.c, same_out.n4, ctx->ratio, ctx->beview, stream_id, do_prop, do_refine, tiles_x, n_tiles, ctx->cost_consistent ? 1 : 0, ctx->dbg);
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    *launched = 1;
    return TSAR_OK;
}
