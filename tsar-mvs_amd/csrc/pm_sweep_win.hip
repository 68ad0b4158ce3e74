// pm_sweep_win.hip — the production form of the red/black half-iteration for 8-bit imagery, box 11,
// n_best <= 2 (the configuration every reference script uses, scripts/*.sh: --blocksize=11 --n_best=1).
// Same arithmetic and results as pm_sweep.hip (the generic form), different data movement.
//
// Why: rocprofv3 PMC on the generic kernel (profiles/r01) shows the vector L1 — not HBM, not VALU — as the
// limiter: TCP_TOTAL_CACHE_ACCESSES ~0.8-0.9 per clock per CU, ~38 tag accesses per 64-lane gather, VALU
// ~44 % busy.  Every source tap is a per-lane 4-byte gather and the L1 retires about one access per clock.
//
// What: once a hypothesis family is coherent (propagation candidates, late refinement steps) the taps of
// a workgroup's 32x16 region land in one compact patch of each source view.  That patch (56x36 texel
// quads, 8 KiB) is staged into LDS with coalesced loads and taps inside it are served by ds_read; taps
// outside (random init-like planes, image borders, outliers) fall back to the global gather.  Where the
// patch sits is only a performance hint, so results do not depend on it.
// To stage one view at a time the propagation loop is turned inside out: the 8 candidates of a pixel do
// not depend on each other's acceptance (gipuma.cu:874-1042 reads only the launch-start state), so all 8
// are scored against view v before moving to view v+1, and the accept chain (:553-563) runs afterwards
// in arm order on the finished costs.  Refinement stays hypothesis-major (each step perturbs the plane
// the previous step accepted, :644-675) and re-stages per view.
#include "pm_core.h"

#define SWEEP_RH 16
#define WIN_W 56
#define WIN_H 36
#define WIN_P 57   // odd pitch: rows fall on different LDS banks



struct Candidate {
    int idx;
    int same;
};
// defined in pm_sweep.hip as a DEVFN; duplicated here because device functions are per-TU (no rdc)
DEVFN void select_candidates_w(const DevScene* __restrict__ sc, const float* __restrict__ c_same, const float* __restrict__ c_other, int x,
                               int y, Candidate cand[8]) {
    const int col = sc->w, row = sc->h;
    const int p = y * col + x;
    const bool fix_seed = sc->flags & TSAR_FLAG_FIX_DOWN_FAR_SEED, fix_cmp = sc->flags & TSAR_FLAG_FIX_RIGHT_FAR_CMP;
    float cmin;
    int cp, cs;
#pragma unroll
    for (int k = 0; k < 8; k++) { cand[k].idx = -1; cand[k].same = 0; }
    if (y > 2) {
        cp = p - 3 * col; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y > 2 + 2 * i) { const int q = p - (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[0].idx = cp;
    }
    if (y < row - 3) {
        cp = p + 3 * col;
        cmin = (fix_seed || y <= 2) ? c_other[cp] : c_other[p - 3 * col];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y < row - 3 - 2 * i) { const int q = p + (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[1].idx = cp;
    }
    if (x > 2) {
        cp = p - 3; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (x > 2 + 2 * i) { const int q = p - 3 - 2 * i; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[2].idx = cp;
    }
    if (x < col - 3) {
        cp = p + 3; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (x < col - 3 - 2 * i) {
                const int q = p + 3 + 2 * i;
                const float v = c_other[q];
                const bool take = fix_cmp ? (v < cmin) : (cmin < v);
                if (take) { cmin = v; cp = q; }
            }
        cand[3].idx = cp;
    }
    if (y > 0) {
        cp = p - col; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y > 1 + i && x > i) { const int q = p - (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (y > 1 + i && x < col - 1 - i) { const int q = p - (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[4].idx = cp; cand[4].same = cs;
    }
    if (y < row - 1) {
        cp = p + col; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y < row - 2 - i && x > i) { const int q = p + (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (y < row - 2 - i && x < col - 1 - i) { const int q = p + (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[5].idx = cp; cand[5].same = cs;
    }
    if (x > 0) {
        cp = p - 1; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x > 1 + i && y > i) { const int q = p - (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (x > 1 + i && y < row - 1 - i) { const int q = p - (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[6].idx = cp; cand[6].same = cs;
    }
    if (x < col - 1) {
        cp = p + 1; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x < col - 2 - i && y > i) { const int q = p + (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (x < col - 2 - i && y < row - 1 - i) { const int q = p + (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[7].idx = cp; cand[7].same = cs;
    }
}

DEVFN bool same_bits_w(const float4& a, const float4& b) {
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
           __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

// pmCost (gipuma.cu:229-298) for one source view; taps inside the staged window come from LDS.
// Identical arithmetic to view_cost<5, STRICT, true> in pm_core.h.
template <bool STRICT>
DEVFN float view_cost_win(const DevScene* __restrict__ sc, const DevView& vw, const unsigned char* tile, int tw, int own, const float* wts,
                          const PixelRef& pr, int x, int y, const float4& n4, const uint32_t* win, int ox, int oy) {
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    float H[9];
    plane_homography(sc->ref, vw, n4, H);
    float sum_src = 0.f, sum_src_src = 0.f, sum_ref_src = 0.f;
    int tap = 0;
    const float fw = (float)w, fh = (float)h;
#pragma unroll 1
    for (int i = -5; i <= 5; i += 2) {
        const float xi = (float)(x + i);
        const float bx = fma_(H[0], xi, H[2]), by = fma_(H[3], xi, H[5]), bz = fma_(H[6], xi, H[8]);
#pragma unroll
        for (int j = -5; j <= 5; j += 2) {
            const float yj = (float)(y + j);
            const float X = fma_(H[1], yj, bx), Y = fma_(H[4], yj, by), Z = fma_(H[7], yj, bz);
            float u, v;
            if (STRICT) {
                u = X / Z;
                v = Y / Z;
            } else {
                const float rz = __builtin_amdgcn_rcpf(Z);
                u = X * rz;
                v = Y * rz;
            }
            u = fminf(fmaxf(u, -1.0f), fw);
            v = fminf(fmaxf(v, -1.0f), fh);
            const float fu = floorf(u), fv = floorf(v);
            const float ax = u - fu, ay = v - fv;
            const int qx = (int)fu + 1, qy = (int)fv + 1;          // coordinates in the quad image
            const int wx = qx - ox, wy = qy - oy;
            uint32_t q;
            if ((unsigned)wx < (unsigned)WIN_W && (unsigned)wy < (unsigned)WIN_H) {
                q = win[__mul24(wy, WIN_P) + wx];
            } else {
                const uint32_t off = (uint32_t)(__mul24(qy, qp) + qx) * 4u;
                q = *(global_u32_ptr)((const char __attribute__((address_space(1)))*)vw.quad + off);
            }
            const float t00 = (float)(q & 0xffu), t10 = (float)((q >> 8) & 0xffu), t01 = (float)((q >> 16) & 0xffu), t11 = (float)(q >> 24);
            const float top = fma_(ax, t10 - t00, t00);
            const float bot = fma_(ax, t11 - t01, t01);
            const float s = fma_(ay, bot - top, top);
            const float r = (float)tile[own + j * tw + i];
            const float wt = wts[tap * PM_BLOCK];
            const float wr = wt * r, ws = wt * s;
            sum_src += ws;
            sum_src_src = fma_(ws, s, sum_src_src);
            sum_ref_src = fma_(wr, s, sum_ref_src);
            ++tap;
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

// running best-2 / valid count / best view of pmCostMultiview_cu (gipuma.cu:455-518)
struct ViewAcc {
    float b0, b1, cmin;
    int valid, bv;
};
DEVFN void acc_reset(ViewAcc& a) { a.b0 = __builtin_inff(); a.b1 = __builtin_inff(); a.cmin = __builtin_inff(); a.valid = 0; a.bv = -1; }
DEVFN void acc_add(ViewAcc& a, float c, int vi) {
    if (c < TSAR_MAXCOST) a.valid++; else c = TSAR_MAXCOST;
    if (c <= a.cmin) { a.cmin = c; a.bv = vi; }
    const float lo = fminf(a.b0, c), hi = fmaxf(a.b0, c);
    a.b0 = lo;
    a.b1 = fminf(a.b1, hi);
}
DEVFN float acc_finish(const DevScene* __restrict__ sc, const ViewAcc& a, int& bv, float& ratio) {
    int nb = a.valid;
    if (sc->cost_comb == TSAR_COMB_BEST_N) nb = min(nb, sc->n_best);
    if (nb <= 0) { bv = -1; ratio = 0.f; return TSAR_MAXCOST; }
    float cost = 0.f + a.b0;
    if (nb > 1) cost += a.b1;
    cost = cost / (float)nb;
    ratio = sc->n_sel >= 2 ? a.b0 / a.b1 : 0.f;
    bv = a.bv;
    return cost;
}

template <bool STRICT>
__global__ __launch_bounds__(PM_BLOCK) void pm_sweep_win_kernel(const DevScene* __restrict__ sc, int colour, const float* __restrict__ c_same,
                                                                const float4* __restrict__ n_same, const float* __restrict__ c_other,
                                                                const float4* __restrict__ n_other, float* c_out, float4* n_out,
                                                                float* __restrict__ ratio_out, int32_t* __restrict__ beview_out,
                                                                uint32_t stream_id, int do_prop, int do_refine, int tiles_x, int n_tiles,
                                                                int cost_consistent) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int hr = 5, vr = 5;
    constexpr int tw = PM_RW + 2 * hr, th = SWEEP_RH + 2 * vr;
    unsigned char* tile = lds_raw;
    float* wts = (float*)(lds_raw + tile_bytes<true>(tw, th)) + threadIdx.x;
    uint32_t* win = (uint32_t*)(lds_raw + tile_bytes<true>(tw, th) + sizeof(float) * 36 * PM_BLOCK);
    int* box = (int*)(win + WIN_P * WIN_H);          // per selected view: min_x, min_y, max_x, max_y of the projected pixels
    __shared__ int org[2 * TSAR_MAX_VIEWS];

    const int t = xcd_tile(blockIdx.x, n_tiles);
    const int ty0 = (t / tiles_x) * SWEEP_RH, tx0 = (t % tiles_x) * PM_RW;
    const int n_sel = sc->n_sel;
    stage_ref_tile<SWEEP_RH, unsigned char>(sc, tile, tx0, ty0, hr, vr);
    for (int k = threadIdx.x; k < n_sel; k += PM_BLOCK) { box[4 * k] = 0x7fffffff; box[4 * k + 1] = 0x7fffffff; box[4 * k + 2] = -0x7fffffff; box[4 * k + 3] = -0x7fffffff; }
    __syncthreads();

    const int ly = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int y = ty0 + ly;
    const int lx = 2 * k + ((colour + y) & 1);
    const int x = tx0 + lx;
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    const bool in_image = x < w && y < h;
    const int p = in_image ? y * w + x : 0;
    const int own = (ly + vr) * tw + lx + hr;
    const DevRef& rf = sc->ref;

    float cost_now = 0.f;
    float4 n_now = make_float4(0.f, 0.f, -1.f, 1.f);
    PixelRef pr;
    pr.inv_wsum = 0.f; pr.mean_ref = 0.f; pr.var_ref = 0.f; pr.textured = false;
    if (in_image) {
        cost_now = c_same[p];
        n_now = n_same[p];
        pr = hoist_reference<5, unsigned char>(tile, tw, own, wts, hr, vr);
    }
    const bool active = in_image && pr.textured;     // inactive threads only help staging and keep the barriers matched

    // ---- where does this workgroup's region land in each source view? (placement hint only) ----
    if (active) {
        for (int i = 0; i < n_sel; i++) {
            float H[9];
            plane_homography(rf, sc->view[sc->sel[i]], n_now, H);
            const float xf = (float)x, yf = (float)y;
            const float Z = fma_(H[7], yf, fma_(H[6], xf, H[8]));
            const float rz = 1.0f / Z;
            const float u = fminf(fmaxf(fma_(H[1], yf, fma_(H[0], xf, H[2])) * rz, -1.0f), (float)w);
            const float v = fminf(fmaxf(fma_(H[4], yf, fma_(H[3], xf, H[5])) * rz, -1.0f), (float)h);
            const int iu = (int)floorf(u) + 1, iv = (int)floorf(v) + 1;
            atomicMin(&box[4 * i], iu); atomicMin(&box[4 * i + 1], iv);
            atomicMax(&box[4 * i + 2], iu); atomicMax(&box[4 * i + 3], iv);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_sel; i += PM_BLOCK) {
        const int x0 = box[4 * i], y0 = box[4 * i + 1], x1 = box[4 * i + 2], y1 = box[4 * i + 3];
        int ox = 0, oy = 0;
        if (x1 >= x0) {
            // centre the window on the bounding box of the projected pixel centres
            ox = (x0 + x1) / 2 - WIN_W / 2;
            oy = (y0 + y1) / 2 - WIN_H / 2;
        }
        ox = max(0, min(ox, w + 2 - WIN_W));
        oy = max(0, min(oy, h + 2 - WIN_H));
        org[2 * i] = ox; org[2 * i + 1] = oy;
    }
    __syncthreads();

    auto stage_window = [&](int i) {
        const DevView& vw = sc->view[sc->sel[i]];
        const int ox = org[2 * i], oy = org[2 * i + 1];
        const global_u32_ptr src = (global_u32_ptr)vw.quad;
        for (int e = threadIdx.x; e < WIN_W * WIN_H; e += PM_BLOCK) {
            const int wy = e / WIN_W, wx = e - wy * WIN_W;
            const int gx = min(ox + wx, w + 1), gy = min(oy + wy, h + 1);
            win[wy * WIN_P + wx] = src[(uint32_t)(gy * qp + gx)];
        }
    };

    bool wrote = false;
    float ratio_w = 0.f;
    int beview_w = 0;
    float depth_now = active ? plane_depth(rf, n_now, x, y) : 1.0f;
    const float4 n_first = n_now;

    // ---- propagation: view-major over the 8 candidates -------------------------------------------
    if (do_prop) {
        float4 nb[8];
        float depth_b[8];
        bool use[8];
        ViewAcc acc[8];
        {
            Candidate cand[8];
#pragma unroll
            for (int a = 0; a < 8; a++) { cand[a].idx = -1; cand[a].same = 0; }
            if (active) select_candidates_w(sc, c_same, c_other, x, y, cand);
#pragma unroll
            for (int a = 0; a < 8; a++) {
                use[a] = false;
                nb[a] = n_now;
                depth_b[a] = 0.f;
                acc_reset(acc[a]);
                if (cand[a].idx >= 0) {
                    nb[a] = cand[a].same ? n_same[cand[a].idx] : n_other[cand[a].idx];
                    depth_b[a] = plane_depth(rf, nb[a], x, y);
                    // same early-outs as pm_sweep.hip: out-of-range planes are never accepted (gipuma.cu:553);
                    // a plane identical to the one this pixel held at launch start re-scores to its own cost
                    use[a] = depth_b[a] >= rf.depthMin && depth_b[a] <= rf.depthMax && !(cost_consistent && same_bits_w(nb[a], n_first));
                }
            }
        }
        for (int i = 0; i < n_sel; i++) {
            __syncthreads();
            stage_window(i);
            __syncthreads();
            const int vi = sc->sel[i];
            const DevView& vw = sc->view[vi];
            const int ox = org[2 * i], oy = org[2 * i + 1];
#pragma unroll
            for (int a = 0; a < 8; a++) {
                if (use[a]) {
                    const float c = view_cost_win<STRICT>(sc, vw, tile, tw, own, wts, pr, x, y, nb[a], win, ox, oy);
                    acc_add(acc[a], c, vi);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 8; a++) {
            if (use[a]) {
                int bv; float rt;
                const float cost_b = acc_finish(sc, acc[a], bv, rt);
                if (cost_b < cost_now) {                         // spatialPropagation_cu gipuma.cu:555-563, arm order
                    cost_now = cost_b; n_now = nb[a]; depth_now = depth_b[a];
                    ratio_w = rt; beview_w = bv; wrote = true;
                }
            }
        }
    }

    // ---- refinement: hypothesis-major (each step starts from what the previous one accepted) -----
    if (do_refine) {
        float vv[3] = {0.f, 0.f, 1.f};
        if (active) view_vector(rf, x, y, vv);
        float deltaN = 1.0f;
        float deltaZ = sc->max_disp / 2.0f;
        const float fb = rf.f * rf.baseline;
        for (int step = 0; step < sc->refine_steps; step++) {
            float4 n_t = n_now;
            float depthOut = depth_now;
            if (active) {
                const Rand4 rn = philox_uniform4((uint32_t)p, stream_id, (uint32_t)step, sc->seed_lo, sc->seed_hi);
                const float disp = fb / depth_now;
                const float minDelta = -fminf(deltaZ, sc->min_disp + disp);
                const float maxDelta = fminf(deltaZ, sc->max_disp - disp);
                const float dz = between(rn.u[0], minDelta, maxDelta);
                const float dispOut = fminf(fmaxf(disp + dz, sc->min_disp), sc->max_disp);
                depthOut = fb / dispOut;
                float nt[3];
                nt[0] = n_now.x + between(rn.u[1], -deltaN, deltaN);
                nt[1] = n_now.y + between(rn.u[2], -deltaN, deltaN);
                nt[2] = n_now.z + between(rn.u[3], -deltaN, deltaN);
                const float inv = 1.0f / sqrtf(dot3(nt, nt));
                nt[0] *= inv; nt[1] *= inv; nt[2] *= inv;
                if (dot3(nt, vv) > 0.0f) { nt[0] = -nt[0]; nt[1] = -nt[1]; nt[2] = -nt[2]; }
                n_t.x = nt[0]; n_t.y = nt[1]; n_t.z = nt[2];
                n_t.w = plane_offset(rf, nt, x, y, depthOut);
            }
            ViewAcc acc;
            acc_reset(acc);
            for (int i = 0; i < n_sel; i++) {
                __syncthreads();
                stage_window(i);
                __syncthreads();
                if (active) {
                    const int vi = sc->sel[i];
                    const float c = view_cost_win<STRICT>(sc, sc->view[vi], tile, tw, own, wts, pr, x, y, n_t, win, org[2 * i], org[2 * i + 1]);
                    acc_add(acc, c, vi);
                }
            }
            if (active) {
                int bv; float rt;
                const float cost_t = acc_finish(sc, acc, bv, rt);
                if (cost_t < cost_now) {
                    cost_now = cost_t; n_now = n_t; depth_now = depthOut;
                    ratio_w = rt; beview_w = bv; wrote = true;
                }
            }
            deltaN = deltaN / 4.0f;
            deltaZ = deltaZ / 10.0f;
        }
    }
    if (in_image) {
        c_out[p] = cost_now;
        n_out[p] = n_now;
        if (wrote) { ratio_out[p] = ratio_w; beview_out[p] = beview_w; }
    }
}

// returns 1 if this specialised form applies and was launched, 0 if the caller should use the generic form
int launch_pm_sweep_win(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out, uint32_t stream_id,
                        int do_prop, int do_refine, int* launched) {
    const DevScene& hs = ctx->hscene;
    *launched = 0;
    const int need = hs.cost_comb == TSAR_COMB_BEST_N ? (hs.n_best < hs.n_sel ? hs.n_best : hs.n_sel) : hs.n_sel;
    if (!(hs.use_quad && hs.hrad == 5 && hs.vrad == 5 && need <= 2 && hs.w + 2 >= WIN_W && hs.h + 2 >= WIN_H)) return TSAR_OK;
    const int tiles_x = (hs.w + PM_RW - 1) / PM_RW, tiles_y = (hs.h + SWEEP_RH - 1) / SWEEP_RH;
    const int n_tiles = tiles_x * tiles_y;
    const size_t lds = tile_bytes<true>(PM_RW + 10, SWEEP_RH + 10) + sizeof(float) * 36 * PM_BLOCK + sizeof(uint32_t) * WIN_P * WIN_H + sizeof(int) * 4 * TSAR_MAX_VIEWS;
    const bool strict = hs.flags & TSAR_FLAG_STRICT_DIV;
    {
        ScopedKernelTimer tm(ctx, "pm_sweep");
        if (strict)
            hipLaunchKernelGGL(pm_sweep_win_kernel<true>, dim3(n_tiles), dim3(PM_BLOCK), lds, ctx->stream, ctx->dscene, colour, same_in.c, same_in.n4, other.c,
                               other.n4, same_out.c, same_out.n4, ctx->ratio, ctx->beview, stream_id, do_prop, do_refine, tiles_x, n_tiles, ctx->cost_consistent ? 1 : 0);
        else
            hipLaunchKernelGGL(pm_sweep_win_kernel<false>, dim3(n_tiles), dim3(PM_BLOCK), lds, ctx->stream, ctx->dscene, colour, same_in.c, same_in.n4, other.c,
                               other.n4, same_out.c, same_out.n4, ctx->ratio, ctx->beview, stream_id, do_prop, do_refine, tiles_x, n_tiles, ctx->cost_consistent ? 1 : 0);
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    *launched = 1;
    return TSAR_OK;
}
