// pm_sweep_impl.h — the fused propagation + refinement kernel of one red/black half-iteration and its launcher, as templates:
// pm_sweep.hip instantiates the production (box 11) configurations, pm_sweep_lut*.hip the general-window ones.
#pragma once
#include "pm_core.h"

// Workgroup = BLK threads = a region of 32 x BLK/16 pixels, one thread per pixel of the active colour.  BLK = 256 (32 x 16) is
// the production shape; BLK = 128 (32 x 8) is used for small images, where 256-thread tiles number fewer than the ~1000
// workgroup slots of the chip and leave CUs idle or unevenly loaded (640 x 480: 600 tiles of 256, 1200 of 128).
#define SWEEP_SMALL_IMAGE_TILES 3072   // below this many 256-thread tiles the 128-thread shape is launched

// A candidate is the pixel index of the neighbour whose plane is tried, with bit 30 set if that neighbour has the
// active colour (its plane is read from same_in); -1 = arm skipped.
#define CAND_SAME (1 << 30)

// 8-arm adaptive candidate selection, gipuma.cu:874-1042.
DEVFN void select_candidates(const DevScene* __restrict__ sc, const float* __restrict__ c_same, const float* __restrict__ c_other,
                             int x, int y, int cand[8]) {
    const int col = sc->w, row = sc->h;
    const int p = y * col + x;
    const bool fix_seed = sc->flags & TSAR_FLAG_FIX_DOWN_FAR_SEED, fix_cmp = sc->flags & TSAR_FLAG_FIX_RIGHT_FAR_CMP;
    float cmin;
    int cp, cs;
#pragma unroll
    for (int k = 0; k < 8; k++) cand[k] = -1;
    // far arms: offsets 3, 5, ..., 23 along the axis -> always the other colour
    if (y > 2) {
        cp = p - 3 * col; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y > 2 + 2 * i) { const int q = p - (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[0] = cp;
    }
    if (y < row - 3) {
        cp = p + 3 * col;
        cmin = (fix_seed || y <= 2) ? c_other[cp] : c_other[p - 3 * col];   // gipuma.cu:906 seeds with c[up_far]
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (y < row - 3 - 2 * i) { const int q = p + (3 + 2 * i) * col; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[1] = cp;
    }
    if (x > 2) {
        cp = p - 3; cmin = c_other[cp];
#pragma unroll
        for (int i = 1; i < 11; ++i)
            if (x > 2 + 2 * i) { const int q = p - 3 - 2 * i; const float v = c_other[q]; if (v < cmin) { cmin = v; cp = q; } }
        cand[2] = cp;
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
        cand[3] = cp;
    }
    // near arms: the 4-neighbour (other colour) and three V pairs (same colour)
    if (y > 0) {
        cp = p - col; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y > 1 + i && x > i) { const int q = p - (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (y > 1 + i && x < col - 1 - i) { const int q = p - (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[4] = cs ? (cp | CAND_SAME) : cp;
    }
    if (y < row - 1) {
        cp = p + col; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (y < row - 2 - i && x > i) { const int q = p + (2 + i) * col - i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (y < row - 2 - i && x < col - 1 - i) { const int q = p + (2 + i) * col + i; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[5] = cs ? (cp | CAND_SAME) : cp;
    }
    if (x > 0) {
        cp = p - 1; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x > 1 + i && y > i) { const int q = p - (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (x > 1 + i && y < row - 1 - i) { const int q = p - (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[6] = cs ? (cp | CAND_SAME) : cp;
    }
    if (x < col - 1) {
        cp = p + 1; cs = 0; cmin = c_other[cp];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (x < col - 2 - i && y > i) { const int q = p + (2 + i) - i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
            if (x < col - 2 - i && y < row - 1 - i) { const int q = p + (2 + i) + i * col; const float v = c_same[q]; if (v < cmin) { cmin = v; cp = q; cs = 1; } }
        }
        cand[7] = cs ? (cp | CAND_SAME) : cp;
    }
}

DEVFN bool same_bits(const float4& a, const float4& b) {
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
           __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

// The propagation memo (tsar_dev.h tsar_ctx::memo_cand): passed by value.
// Why dropping an arm changes nothing.  spatialPropagation_cu (gipuma.cu:524-566) takes a neighbour's plane iff its score at this
// pixel is STRICTLY below the pixel's cost, and a sweep changes a pixel's state in no other way than by such a take (refinement
// :621-676 likewise), so (1) a pixel's cost never rises.  (2) The score of a plane at a pixel is a function of the two, the views and
// the parameters — nothing the sweeps change — and every form of the tap loop returns the same bits for it.  Say pixel p tried the
// plane of neighbour q in launch M: whatever happened then — scored and rejected, scored and taken, not scored because p held that
// very plane or because its depth was out of range, or dropped by this memo (induction) — p's cost after M is at most that score, or
// the plane can never be taken.  If q's plane has not changed since (changed[q] < M: q's plane at the start of M is its plane now)
// and q is the arm's candidate again, the reference scores it again and rejects it again by (1) and (2); the memo skips the scoring.
// It holds while c[p] is the score of n4[p] under the current views (cost_consistent) and only among the launches of one
// tsar_pm_iterate call (valid_from), so that nothing but sweeps has touched the state between M and now.
struct SweepMemo {
    int32_t* cand;                   // [pixel][8]: the candidates of the pixel's previous propagation launch
    uint32_t* seq;                   // [pixel]: the launch that wrote them
    uint32_t* changed;               // [pixel]: the last launch in which the pixel's plane changed
    uint32_t launch, valid_from;     // this launch's number; memos written before valid_from are void
    int mode;                        // 0 = off, 1 = kept and applied (a memo-skipped arm is an arm skipped)
    int slot_off;                    // CMP kernels: byte offset in dynamic LDS of the waves' hand-over slots (6 bytes per lane)
};

// One hypothesis of the propagation / refinement loop scored for pixel (x, y): what the loop body shares between its forms.
// CMP = false: the rolled loop over the eight arms and the refinement steps, each lane scoring its own pixel's hypotheses (a lane
// whose arm is skipped idles through that arm).
// CMP = true (launches whose memo removes a good part of the arms — from the fourth iteration of a run on: TSAR_COMPACT_FROM): the surviving
// (pixel, arm) pairs of a WAVE are packed, 64 per trip, whichever lanes' pixels they belong to: trip t scores pairs 64 t .. 64 t + 63
// in arm-major order, lane j scoring pair 64 t + j for its owner, who publishes the pair (its lane, the arm, the candidate) in the
// wave's LDS slots before the trip and collects the cost (ds_bpermute) after it.  Arms arrive at an owner in increasing order and
// the accept test is the reference's strict `<` (gipuma.cu:555), so the state after the last trip is the state after the
// reference's eight sequential calls.  The refinement steps follow, one per trip, every lane on its own pixel as before.
template <int NB, int HR, bool STRICT, bool QUAD, int V = 0, int BLK = PM_BLOCK, bool CMP = false>
__global__ __launch_bounds__(BLK) void pm_sweep_kernel(const DevScene* __restrict__ sc, int colour,
                                                            const float* __restrict__ c_same, const float4* __restrict__ n_same,
                                                            const float* __restrict__ c_other, const float4* __restrict__ n_other,
                                                            float* c_out, float4* n_out, float* __restrict__ ratio_out,
                                                            int32_t* __restrict__ beview_out, uint32_t stream_id, int do_prop,
                                                            int do_refine, int tiles_x, int n_tiles, int cost_consistent, int strip_w,
                                                            const float* __restrict__ final_text, SweepMemo memo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename TileOf<QUAD>::type TileT;
    // region of a workgroup: 32 pixels wide, 2 BLK / 32 high (one thread per pixel of the active colour)
    constexpr int RW = PM_RW;
    constexpr int SWEEP_RH = 2 * BLK / RW;
    const int hr = HR > 0 ? HR : sc->hrad, vr = HR > 0 ? HR : sc->vrad;
    const int tw = RW + 2 * hr, th = SWEEP_RH + 2 * vr;
    // LDS: [reference window][S weights per thread], or (variant bit 10) [shared weight table][reference window]
    constexpr bool LUTW = (V & 1024) != 0;
    const size_t lut_bytes = LUTW ? (size_t)(sc->lut_classes + 1) * 1024 : 0;
    TileT* tile = (TileT*)(lds_raw + lut_bytes);
    float* const wts_base = LUTW ? (float*)lds_raw : (float*)(lds_raw + tile_bytes<QUAD>(tw, th));
    float* wts = LUTW ? wts_base : wts_base + threadIdx.x;
    if constexpr (LUTW) build_weight_lut<BLK>(sc, wts);

    const int t = xcd_tile(blockIdx.x, n_tiles);
    int tix, tiy;
    strip_tile(t, tiles_x, n_tiles / tiles_x, strip_w, tix, tiy);
    const int ty0 = tiy * SWEEP_RH, tx0 = tix * RW;
    stage_ref_tile<SWEEP_RH, TileT, BLK, RW>(sc, tile, tx0, ty0, hr, vr, LUTW ? LUT_TILE_PAD_ROWS : 0);
    __syncthreads();

    // lane -> pixel: a wave covers 4 rows x 16 pixels of the active colour (a 32 x 4 pixel strip).  The other shapes the round-1
    // review asked to have measured lose: 8 x 8 pixels per wave +9.6 %, 16 rows x 4 +61 %, 2 rows x 32 +0.7 % (profiles/r02/ab_lane_maps.json)
    const int ly = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int y = ty0 + ly;
    const int lx = 2 * k + ((colour + y) & 1);            // (x + y) & 1 == colour; gipuma.cu:1099-1103 / :1121-1125
    const int x = tx0 + lx;
    const int w = sc->w, h = sc->h;
    // (CMP: every lane of a wave stays to the end — it scores other lanes' hypotheses and serves their ds_bpermute reads)
    bool mine = x < w && y < h;                            // this lane has a pixel to update
    if (!CMP && !mine) return;
    const int p = mine ? y * w + x : 0;
    const int own = (ly + vr) * tw + lx + hr;

    float cost_now = TSAR_MAXCOST;
    float4 n_now = {0.f, 0.f, 0.f, 0.f};
    if (mine) { cost_now = c_same[p]; n_now = n_same[p]; }
    // the kernels' `final == true` mode (gipuma.cu:856, :1063): pixels whose lines->text is -1 keep their state
    // (copied across the ping-pong), and no accepted hypothesis writes ratio / beview (:559-562, :669-672)
    if (mine && final_text && final_text[p] == -1.0f) {
        c_out[p] = cost_now; n_out[p] = n_now;
        if (!CMP) return;
        mine = false;
    }
    PixelRef pr;
    pr.inv_wsum = 0.f; pr.mean_ref = 0.f; pr.var_ref = 0.f; pr.textured = false;
    if (mine) {
        if constexpr (LUTW) pr = hoist_reference_lut(sc, tile, tw, own, wts);
        else pr = hoist_reference<HR, TileT, BLK>(tile, tw, own, wts, hr, vr);
    }
    bool wrote = false;
    float ratio_w = 0.f;
    int beview_w = 0;
    const DevRef& rf = sc->ref;
    const bool work = mine && pr.textured;
    int cand[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // neighbour pixel index | same-colour flag << 30, -1 = arm skipped
    if (work && do_prop) {
        select_candidates(sc, c_same, c_other, x, y, cand);
        if (memo.mode) {
            // a candidate that is the neighbour this arm tried in the pixel's previous propagation launch, with a plane unchanged since
            const uint32_t m_seq = memo.seq[p];
            const bool have = m_seq >= memo.valid_from && cost_consistent;
            const int4* mc = (const int4*)(memo.cand + (size_t)p * 8);
            int prev[8];
            if (have) {
                const int4 a0 = mc[0], a1 = mc[1];
                prev[0] = a0.x; prev[1] = a0.y; prev[2] = a0.z; prev[3] = a0.w; prev[4] = a1.x; prev[5] = a1.y; prev[6] = a1.z; prev[7] = a1.w;
            }
            int4 w0, w1;
            w0.x = cand[0]; w0.y = cand[1]; w0.z = cand[2]; w0.w = cand[3]; w1.x = cand[4]; w1.y = cand[5]; w1.z = cand[6]; w1.w = cand[7];
            ((int4*)(memo.cand + (size_t)p * 8))[0] = w0;
            ((int4*)(memo.cand + (size_t)p * 8))[1] = w1;
            memo.seq[p] = memo.launch;
            if (have) {
#pragma unroll
                for (int a = 0; a < 8; a++)
                    if (cand[a] >= 0 && cand[a] == prev[a] && memo.changed[cand[a] & 0x3fffffff] < m_seq) cand[a] = -1;
            }
        }
    }
    float depth_now = 0.f;
    float vv[3] = {0.f, 0.f, 0.f};
    if (work) { depth_now = plane_depth(rf, n_now, x, y); view_vector(rf, x, y, vv); }
    const float4 n_first = n_now;
    float deltaN = 1.0f;
    float deltaZ = sc->max_disp / 2.0f;
    const float fb = rf.f * rf.baseline;
    const int n_ref = do_refine ? sc->refine_steps : 0;

    if constexpr (!CMP) {
        if (work) {
            // One rolled loop over the hypotheses of this pixel: h = 0..7 the propagation arms in the reference's
            // order (gipuma.cu:874-1042), h = 8.. the refinement steps (:1066-1090).  The loop counter is wave-uniform,
            // so the arm/step switch is a scalar branch and the multi-view cost (the whole tap loop) exists once in the
            // binary instead of nine times: ~6 KB of hot code instead of ~45 KB, and fewer live registers.
            const int h_end = 8 + n_ref;
#pragma unroll 1
            for (int h = do_prop ? 0 : 8; h < h_end; h++) {
                float4 n_t;
                float depth_t;
                if (h < 8) {
                    int ci = cand[0];
#pragma unroll
                    for (int a = 1; a < 8; a++) ci = (h == a) ? cand[a] : ci;
                    if (ci < 0) continue;
                    const int idx = ci & 0x3fffffff;
                    n_t = (ci >> 30) ? n_same[idx] : n_other[idx];
                    // A neighbour often carries the very plane this pixel already holds (or held when the launch
                    // started): planes spread by verbatim copies.  While c[p] is the score of norm4[p] (true for
                    // every state produced by init / sweeps) re-scoring it returns a cost that is not smaller
                    // than cost_now, so the reference's `cost_before < *cost_now` (gipuma.cu:555) rejects it.
                    if (cost_consistent && (same_bits(n_t, n_now) || same_bits(n_t, n_first))) continue;
                    depth_t = plane_depth(rf, n_t, x, y);
                    // spatialPropagation_cu gipuma.cu:524-566; the range test is done first: a
                    // hypothesis outside [depthMin, depthMax] is never accepted, so it is not scored.
                    if (!(depth_t >= rf.depthMin && depth_t <= rf.depthMax)) continue;
                } else {
                    // planeRefinement_cu gipuma.cu:621-676 + getRndDispAndUnitVector_cu :582-619
                    const Rand4 rn = philox_uniform4((uint32_t)p, stream_id, (uint32_t)(h - 8), sc->seed_lo, sc->seed_hi);
                    const float disp = fb / depth_now;
                    const float minDelta = -fminf(deltaZ, sc->min_disp + disp);   // "+" as written, gipuma.cu:601
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
                }
                int bv; float rt;
                const float cost_t = multiview_cost<NB, HR, STRICT, QUAD, V, BLK>(sc, tile, tw, own, wts, pr, x, y, n_t, bv, rt);
                if (cost_t < cost_now) {
                    cost_now = cost_t; n_now = n_t; depth_now = depth_t;
                    ratio_w = rt; beview_w = bv; wrote = true;
                }
            }
        }
    } else {
        // ---- the packed form ----
        // Register diet: what a lane keeps across the scoring is its pixel's running state (cost, plane, ratio, best view), the
        // hoisted reference terms other lanes read by ds_bpermute, and the mask of its arms.  Everything else is recomputed where it
        // is needed — the pixel's coordinates from the thread index, the candidates from the memo this lane wrote in the prologue,
        // the view vector and the step widths of a refinement step from the pixel and the step's number — so that the kernel
        // keeps the four waves per SIMD of the rolled form.
        const int lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63;
        volatile int32_t* slot_ci = (volatile int32_t*)(lds_raw + memo.slot_off) + wave0;                                   // [BLK] candidates
        volatile unsigned short* slot_src = (volatile unsigned short*)(lds_raw + memo.slot_off + 4 * BLK) + wave0;         // [BLK] owner lane | arm << 6
        auto pixel_of = [&](int tid, int& px, int& py, int& pown) {
            const int sly = tid >> 4, sk = tid & 15;
            py = ty0 + sly;
            const int slx = 2 * sk + ((colour + py) & 1);
            px = tx0 + slx;
            pown = (sly + vr) * tw + slx + hr;
        };
        auto my_tid = []() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; };   // (opaque: what is derived from it is recomputed, not kept)
        // which arms of this pixel are to be scored: every arm that has a candidate the memo did not remove.  The rolled loop's other two
        // tests wait for the lane that scores the pair and reads the candidate's plane anyway: a depth out of range scores as never
        // accepted; a plane the pixel already holds is scored like any other and rejected by the strict `<` (2 % of the pairs, against
        // eight scattered 16-byte reads per pixel up front: measured, profiles/r05/README.md section 12)
        uint32_t fresh = 0;
        if (work && do_prop) {
#pragma unroll
            for (int a = 0; a < 8; a++)
                if (cand[a] >= 0) fresh |= 1u << a;
        }
        if (!work) fresh = 0;
        uint32_t flags = fresh | (work ? 256u : 0u) | (mine ? 512u : 0u);      // bits 0-7 arms, 8 work, 9 mine
        // wave-uniform: how many pairs the wave has.  (Which lanes have which arm — eight ballots — and where each arm's pairs start
        // in the packed order are recomputed from the mask in every trip, a v_cmp and an s_bcnt1 per arm: kept across the scoring they
        // would cost 25 scalar registers the kernel does not have.)
        int total = 0;
#pragma unroll
        for (int a = 0; a < 8; a++) total += __popcll(__ballot((flags >> a) & 1u));
        const int trips = (total + 63) >> 6;
        // for every arm with pairs in trip [lo, lo + 64): f(a, j, has) — j = the slot of this lane's pair of that arm, has = it has one
        auto arms_of_trip = [&](int lo, auto&& f) {
            asm volatile("" : "+v"(flags));               // (no common subexpressions with the trip's other phase across the scoring)
            int b0 = 0;
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const uint64_t bal = __ballot((flags >> a) & 1u);
                const int n = __popcll(bal);
                if (b0 + n > lo && b0 < lo + 64) {        // (wave-uniform)
                    const int j = b0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)) - lo;
                    f(a, j, ((flags >> a) & 1u) && j >= 0 && j < 64);
                }
                b0 += n;
            }
        };
        int acc_arm = -1;                                  // the arm whose hypothesis the pixel holds after the propagation trips
        const float start_dz = sc->max_disp / 2.0f;
#pragma unroll 1
        for (int it = 0; it < trips + n_ref; it++) {
            float4 n_t = {0.f, 0.f, 0.f, 0.f};
            float depth_t = 0.f;
            int ex, ey, eown, etid = my_tid();
            PixelRef epr = pr;
            bool go;
            const int lo = it << 6;
            if (it < trips) {
                // owners publish this trip's pairs (the candidate from the memo written in the prologue)
                {
                    int mx, my, mown;
                    pixel_of(my_tid(), mx, my, mown);
                    const int32_t* mc = memo.cand + (size_t)(my * w + mx) * 8;
                    arms_of_trip(lo, [&](int a, int j, bool has) {
                        if (has) { slot_ci[j] = mc[a]; slot_src[j] = (unsigned short)(lane | (a << 6)); }
                    });
                }
                __builtin_amdgcn_wave_barrier();
                go = lane < total - lo;
                int src = 0, ci = 0;
                if (go) { src = slot_src[lane]; ci = slot_ci[lane]; }
                __builtin_amdgcn_wave_barrier();
                const int s = src & 63;
                epr.inv_wsum = __int_as_float(__builtin_amdgcn_ds_bpermute(s << 2, __float_as_int(pr.inv_wsum)));
                epr.mean_ref = __int_as_float(__builtin_amdgcn_ds_bpermute(s << 2, __float_as_int(pr.mean_ref)));
                epr.var_ref = __int_as_float(__builtin_amdgcn_ds_bpermute(s << 2, __float_as_int(pr.var_ref)));
                if (go) {
                    const int idx = ci & 0x3fffffff;
                    n_t = (ci >> 30) ? n_same[idx] : n_other[idx];
                    etid = wave0 + s;
                }
                pixel_of(etid, ex, ey, eown);
                if (go) {                                   // spatialPropagation_cu gipuma.cu:553: a hypothesis outside [depthMin, depthMax] is never accepted
                    const float d = plane_depth(rf, n_t, ex, ey);
                    go = d >= rf.depthMin && d <= rf.depthMax;
                }
            } else {
                // planeRefinement_cu gipuma.cu:621-676 + getRndDispAndUnitVector_cu :582-619, every lane on its own pixel
                pixel_of(etid, ex, ey, eown);
                if (it == trips && acc_arm >= 0) {          // what the propagation trips left: the accepted arm's plane
                    const int ci = memo.cand[(size_t)(ey * w + ex) * 8 + acc_arm];
                    const int idx = ci & 0x3fffffff;
                    n_now = (ci >> 30) ? n_same[idx] : n_other[idx];
                    depth_now = plane_depth(rf, n_now, ex, ey);
                    acc_arm = -1;
                }
                go = (flags >> 8) & 1u;
                if (go) {
                    const int step = it - trips;
                    float dN = 1.0f, dZ = start_dz;          // the widths of this step: the reference's running divisions, redone
                    for (int q = 0; q < step; q++) { dN = dN / 4.0f; dZ = dZ / 10.0f; }
                    float ev[3];
                    view_vector(rf, ex, ey, ev);
                    const Rand4 rn = philox_uniform4((uint32_t)(ey * w + ex), stream_id, (uint32_t)step, sc->seed_lo, sc->seed_hi);
                    const float disp = fb / depth_now;         // (the depth the accepted hypothesis was made with, not the plane's: gipuma.cu:664)
                    const float minDelta = -fminf(dZ, sc->min_disp + disp);   // "+" as written, gipuma.cu:601
                    const float maxDelta = fminf(dZ, sc->max_disp - disp);
                    const float dz = between(rn.u[0], minDelta, maxDelta);
                    const float dispOut = fminf(fmaxf(disp + dz, sc->min_disp), sc->max_disp);
                    depth_t = fb / dispOut;
                    float nt[3];
                    nt[0] = n_now.x + between(rn.u[1], -dN, dN);
                    nt[1] = n_now.y + between(rn.u[2], -dN, dN);
                    nt[2] = n_now.z + between(rn.u[3], -dN, dN);
                    const float inv = 1.0f / sqrtf(dot3(nt, nt));
                    nt[0] *= inv; nt[1] *= inv; nt[2] *= inv;
                    if (dot3(nt, ev) > 0.0f) { nt[0] = -nt[0]; nt[1] = -nt[1]; nt[2] = -nt[2]; }
                    n_t.x = nt[0]; n_t.y = nt[1]; n_t.z = nt[2];
                    n_t.w = plane_offset(rf, nt, ex, ey, depth_t);
                }
            }
            const float* ewts = LUTW ? wts_base : wts_base + etid;
            int bv = -1;
            float rt = 0.f, cost_t = TSAR_MAXCOST;
            if (go) cost_t = multiview_cost<NB, HR, STRICT, QUAD, V, BLK>(sc, tile, tw, eown, ewts, epr, ex, ey, n_t, bv, rt);
            if (it < trips) {
                // owners collect, arms in increasing order: the reference's eight calls one after the other
                arms_of_trip(lo, [&](int a, int j, bool has) {
                    const float c = __int_as_float(__builtin_amdgcn_ds_bpermute((j & 63) << 2, __float_as_int(cost_t)));
                    const float r = __int_as_float(__builtin_amdgcn_ds_bpermute((j & 63) << 2, __float_as_int(rt)));
                    const int b = __builtin_amdgcn_ds_bpermute((j & 63) << 2, bv);
                    if (has && c < cost_now) {
                        cost_now = c; acc_arm = a;
                        ratio_w = r; beview_w = b;
                    }
                });
            } else if (go && cost_t < cost_now) {
                cost_now = cost_t; n_now = n_t; depth_now = depth_t;
                ratio_w = rt; beview_w = bv;
            }
        }
        {
            int mx, my, mown;
            pixel_of(my_tid(), mx, my, mown);
            if (!((flags >> 9) & 1u)) return;
            const int pp = my * w + mx;
            if (acc_arm >= 0) {                            // no refinement step followed the propagation trips
                const int ci = memo.cand[(size_t)pp * 8 + acc_arm];
                const int idx = ci & 0x3fffffff;
                n_now = (ci >> 30) ? n_same[idx] : n_other[idx];
            }
            const bool changed = cost_now < c_same[pp];    // every accepted hypothesis lowered the cost, and nothing else did
            c_out[pp] = cost_now;
            n_out[pp] = n_now;
            if (changed && !final_text) { ratio_out[pp] = ratio_w; beview_out[pp] = beview_w; }
            if (changed && memo.mode) memo.changed[pp] = memo.launch;
            return;
        }
    }
    c_out[p] = cost_now;
    n_out[p] = n_now;
    if (wrote && !final_text) { ratio_out[p] = ratio_w; beview_out[p] = beview_w; }
    if (wrote && memo.mode) memo.changed[p] = memo.launch;
}


// the tap loops of 8-bit imagery — box 11's own and the general-window one — have a packed form (CMP) beside the rolled one
template <int HR, bool QUAD, int V>
constexpr bool sweep_has_packed_form() { return QUAD && ((HR == 5 && r5_production_variant(V)) || (V & 1024) != 0); }

template <int NB, int HR, bool STRICT, bool QUAD, int V = 0, int BLK = PM_BLOCK>
static int launch_sweep_t(tsar_ctx* ctx, int colour, const PlaneBuf& same_in, const PlaneBuf& other, const PlaneBuf& same_out,
                          uint32_t stream_id, int do_prop, int do_refine) {
    const DevScene& hs = ctx->hscene;
    constexpr int RW = PM_RW;
    constexpr int SWEEP_RH = 2 * BLK / RW;
    const int tiles_x = (hs.w + RW - 1) / RW, tiles_y = (hs.h + SWEEP_RH - 1) / SWEEP_RH;
    const int n_tiles = tiles_x * tiles_y;
#ifdef TSAR_EXPERIMENTS
    const size_t lds_pad = ctx->lds_pad;       // occupancy experiments: unused LDS per workgroup
#else
    constexpr size_t lds_pad = 0;
#endif
    size_t lds = tile_bytes<QUAD>(RW + 2 * hs.hrad, SWEEP_RH + 2 * hs.vrad + ((V & 1024) ? LUT_TILE_PAD_ROWS : 0)) + lds_pad +
                 ((V & 1024) ? (size_t)(hs.lut_classes + 1) * 1024 : sizeof(float) * (size_t)(hs.hrad + 1) * (hs.vrad + 1) * BLK);
    SweepMemo memo;
    memo.cand = ctx->memo_cand; memo.seq = ctx->memo_seq; memo.changed = ctx->changed_seq;
    memo.launch = ctx->launch_seq; memo.valid_from = ctx->memo_valid_from;
    memo.mode = (ctx->memo_mode && ctx->memo_cand && ctx->cost_consistent && !ctx->final_text) ? 1 : 0;
    memo.slot_off = 0;
    bool packed = false;
    if constexpr (sweep_has_packed_form<HR, QUAD, V>())
        packed = memo.mode && do_prop && ctx->compact_from >= 0 && ctx->call_launch >= ctx->compact_from;
    auto kern = pm_sweep_kernel<NB, HR, STRICT, QUAD, V, BLK, false>;
    if constexpr (sweep_has_packed_form<HR, QUAD, V>()) {
        if (packed) {
            kern = pm_sweep_kernel<NB, HR, STRICT, QUAD, V, BLK, true>;
            memo.slot_off = (int)lds;
            lds += 6 * (size_t)BLK;
        }
    }
    if (lds > 64 * 1024) TSAR_HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ScopedKernelTimer tm(ctx, "pm_sweep");
        ScopedKernelTimer tm_packed(ctx, packed ? "pm_sweep_packed" : nullptr);      // (the packed launches a second time under their own name)
        hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(BLK), lds, ctx->stream, ctx->dscene, colour, same_in.c, same_in.n4, other.c,
                           other.n4, same_out.c, same_out.n4, ctx->ratio, ctx->beview, stream_id, do_prop, do_refine, tiles_x, n_tiles,
                           ctx->cost_consistent ? 1 : 0, strip_width(ctx->strip_w, tiles_x), ctx->final_text, memo);
    }
    TSAR_HIP_TRY(ctx, hipGetLastError());
    return TSAR_OK;
}
