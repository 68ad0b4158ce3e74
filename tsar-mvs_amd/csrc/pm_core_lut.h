// pm_core_lut.h — the matching cost for ANY window (box 1..63, square or not) on 8-bit imagery: the production tap loop of
// pm_core.h (view_cost, variant 250 / 122) is written for the scripts' box 11; every other box — the reference binary's own
// default is 19 (algorithmparameters.h:25) — used to fall to a one-tap-at-a-time loop whose S hoisted weights per thread
// (100 KiB of LDS per workgroup at box 19) left one wave per SIMD.  Included by pm_core.h.
//
// What differs from the box-11 loop:
//   * weights.  exp(-sqrt(i^2 + j^2) / 50 - |r - centre| / 18) (gipuma.cu:268) takes its second argument from the integers
//     0..255 on 8-bit imagery and its first from the few distinct tap distances of the window (14 at box 19), so the
//     workgroup keeps ONE table [distance class][0..255] in LDS — evaluated with the same expression, so the values are the
//     oracle's bit for bit — and a tap looks its weight up with |r - centre| as the index.  LDS per workgroup: 15 KiB at box 19
//     instead of 100, occupancy is bounded by registers again (four waves per SIMD).  Cost: three VALU instructions and one
//     LDS gather per tap.
//   * shape.  A line of the window (a row in fast mode, a column — the oracle's summation order — in strict mode) is walked
//     in chunks of CH = 4, 5 or 6 taps, each chunk in the three phases of the box-11 loop (positions, gathers, blend), CH
//     chosen by the host so that the line length rounds up with the fewest padding slots.  A padding slot repeats the line's
//     last tap with a weight from the table's all-zero row: it adds +0 to the three sums, so the result does not depend on CH.
#pragma once

// The table: (classes + 1) rows of 256 floats at the start of the workgroup's LDS.
template <int BLK>
DEVFN void build_weight_lut(const DevScene* __restrict__ sc, float* lut) {
    const int nc = sc->lut_classes;
    for (int k = threadIdx.x; k < (nc + 1) * 256; k += BLK) {
        const int cls = k >> 8;
        float wt = 0.0f;
        if (cls < nc) {
            const float sd = sqrtf((float)sc->lut_d2[cls]);
            const float cd = (float)(k & 255);
            wt = tsar_expf(-sd / 50.0f - cd / 18.0f);      // the expression of hoist_reference (pm_core.h), gipuma.cu:268
        }
        lut[k] = wt;
    }
}

DEVFN float lut_weight(const float* lut, uint32_t row_bytes, float r, float cen) {
    const int idx = (int)fabsf(r - cen);                    // integer-valued: texels are 0..255
    return *(const float*)((const char*)lut + row_bytes + (uint32_t)idx * 4u);
}

// hoist_reference (pm_core.h) with the weights read from the table; the oracle's tap order (columns).
DEVFN PixelRef hoist_reference_lut(const DevScene* __restrict__ sc, const unsigned short* tile, int tw, int own, const float* lut) {
    const int hr = sc->hrad, vr = sc->vrad;
    const bool rowm = sc->lut_row_major != 0;
    const int pt = sc->lut_pad_taps;
    const float cen = tile_value(tile[own]);
    float sum_ref = 0.f, sum_ref_ref = 0.f, wsum = 0.f;
    for (int ii = 0; ii <= hr; ii++) {
        const int i = 2 * ii - hr;
        for (int jj = 0; jj <= vr; jj++) {
            const int j = 2 * jj - vr;
            const float r = tile_value(tile[own + j * tw + i]);
            const float wt = lut_weight(lut, rowm ? sc->tap_row[jj * pt + ii] : sc->tap_row[ii * pt + jj], r, cen);
            const float wr = wt * r;
            sum_ref += wr;
            sum_ref_ref = fma_(wr, r, sum_ref_ref);
            wsum += wt;
        }
    }
    PixelRef pr;
    pr.inv_wsum = 1.0f / wsum;
    sum_ref *= pr.inv_wsum;
    sum_ref_ref *= pr.inv_wsum;
    pr.mean_ref = sum_ref;
    pr.var_ref = sum_ref_ref - sum_ref * sum_ref;
    pr.textured = !(pr.var_ref < 1e-5f);
    return pr;
}

// pmCost (gipuma.cu:229-298) for one source view, any window, 8-bit quad textures.
// Fast mode (rows): the window of the workgroup is staged with one extra row (LUT_TILE_PAD_ROWS), so that the padding slots of a
// row's last chunk may read the texels that follow the row (finite values, zero weight) with immediate offsets.
#define LUT_TILE_PAD_ROWS 1
// The chunk's D16 window loads (issued by asm, invisible to the compiler's counters) have returned.  `after`: the byte offset of
// the chunk's last gather — an input, so that the wait cannot be scheduled above the tap-position arithmetic.
template <int CH>
DEVFN void lut_wait_lds(float (&r)[CH], uint32_t after) {
    if constexpr (CH == 4) asm("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(after));
    else if constexpr (CH == 5) asm("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "v"(after));
    else asm("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "v"(after));
}
template <bool STRICT, int CH, bool BUF = false, bool MIX = false>
DEVFN float view_cost_lut(const DevScene* __restrict__ sc, const DevView& vw, const unsigned short* tile, int tw, int own, const float* lut,
                          const PixelRef& pr, int x, int y, const float4& n4) {
    constexpr bool ROW = !STRICT;                           // fast mode walks window rows (see pm_core.h, variant bit 7)
    const int hr = sc->hrad, vr = sc->vrad;
    const int rt = ROW ? hr : vr, rl = ROW ? vr : hr;       // radius along / across the lines
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    float H[9];
    if (STRICT) plane_homography(sc->ref, vw, n4, H, sc->k_sparse != 0);
    else plane_homography_fast(sc->ref, vw, n4, H);
    // clamp-free loop when the four corner taps of every active lane land inside the source image with Z > 0 and a pixel of
    // margin (pm_core.h, variant bit 4): wave-uniform, identical results.  In fast mode the padding slots of a row's last chunk
    // are sampled where they fall, beyond the window's right edge: the corners include them.
    const int xr = ROW ? 2 * (sc->lut_pad_taps - 1) - hr : hr;
    bool inside = true;
    float zmin = __builtin_inff(), zmax = 0.0f;
    if (!STRICT) {
        // fast mode: the decision from the window's centre and a bound on its extent (pm_tap_r5.h): one reciprocal instead of four.
        // |dx| <= max(hr, xr) (the padding slots reach further right than the window), |dy| <= vr
        const float ex = (float)max(hr, xr), ey = (float)vr;
        const float xc = (float)x, yc = (float)y;
        const float Xc = fma_(H[1], yc, fma_(H[0], xc, H[2])), Yc = fma_(H[4], yc, fma_(H[3], xc, H[5])), Zc = fma_(H[7], yc, fma_(H[6], xc, H[8]));
        const float a = fma_(ex, fabsf(H[0]), ey * fabsf(H[1])), b = fma_(ex, fabsf(H[3]), ey * fabsf(H[4])), c = fma_(ex, fabsf(H[6]), ey * fabsf(H[7]));
        const float Zmin = Zc - c;
        const float r = __builtin_amdgcn_rcpf(Zmin * Zc);
        const float rc = Zmin * r;                                  // 1 / Zc
        const float du = fma_(a, Zc, fabsf(Xc) * c) * r, dv = fma_(b, Zc, fabsf(Yc) * c) * r;
        const float uc = Xc * rc, vc = Yc * rc;
        inside = Zmin > 0.0f && fminf(uc - du, vc - dv) >= 1.5f && uc + du <= (float)(w - 1) - 1.5f && vc + dv <= (float)(h - 1) - 1.5f;
    } else
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float xi = (float)(x + ((c & 1) ? xr : -hr)), yj = (float)(y + ((c & 2) ? vr : -vr));
        const float X = fma_(H[1], yj, fma_(H[0], xi, H[2])), Y = fma_(H[4], yj, fma_(H[3], xi, H[5])), Z = fma_(H[7], yj, fma_(H[6], xi, H[8]));
        const float rz = __builtin_amdgcn_rcpf(Z);
        const float u = X * rz, v = Y * rz;
        inside = inside && Z > 0.0f && u >= 1.0f && u <= (float)(w - 2) && v >= 1.0f && v <= (float)(h - 2);
        if (STRICT) { zmin = fminf(zmin, Z); zmax = fmaxf(zmax, Z); }
    }
    if (STRICT) {       // the clamp-free strict loop runs persp_divide_exact without its per-tap guard: see view_cost (pm_core.h) for the bounds
        const float cm = (float)(max(w, h) + 32);
        const float sz = fma_(fabsf(H[6]) + fabsf(H[7]), cm, fabsf(H[8]));
        const float sx = fma_(fabsf(H[0]) + fabsf(H[1]), cm, fabsf(H[2]));
        const float sy = fma_(fabsf(H[3]) + fabsf(H[4]), cm, fabsf(H[5]));
        inside = inside && zmin >= 3.814697265625e-06f && zmax <= 131072.0f && sz * cm <= 524288.0f * zmin && fmaxf(sx, sy) <= 262144.0f * zmin;
    }
    const bool need_clamp = !__all(inside);
    // quad-texture base with the border offset folded in, pinned in an SGPR pair for the whole view (pm_core.h, variant bit 6):
    // offsets are unsigned from entry (1, 1), so positions are clamped to [0, w - 1] x [0, h - 1] — the same samples, bit for
    // bit, as the oracle's clamp to [-1, w] (edge replication)
    const uint64_t qa = (uint64_t)(uintptr_t)vw.quad + (uint32_t)((qp + 1) << 2);
    uint32_t qb_lo = __builtin_amdgcn_readfirstlane((uint32_t)qa), qb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(qa >> 32));
    asm volatile("" : "+s"(qb_lo), "+s"(qb_hi));
    // BUF (fast mode, from the second sweep of a run on): the gathers as structured buffer loads through a stride-4 resource
    // descriptor — see pm_core.h, variant bit 17
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    u32x4s rsrc = {0u, 0u, 0u, 0u};
    static_assert(!MIX || (BUF && !STRICT), "the half-float difference texture serves the fast arithmetic's blend through buffer loads");
    if constexpr (MIX) {                                    // 8-byte entries of the difference texture (pm_tap_r5.h MIX), same pitch and border
        const uint64_t da = (uint64_t)(uintptr_t)vw.dquad + 2 * (uint64_t)(uint32_t)((qp + 1) << 2);
        rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)da);
        rsrc.y = __builtin_amdgcn_readfirstlane(((uint32_t)(da >> 32) & 0xffffu) | (8u << 16));
        rsrc.z = __builtin_amdgcn_readfirstlane((uint32_t)(qp * (h + 1) - 1));
        rsrc.w = 0x00020000u;
        asm volatile("" : "+s"(rsrc));
    } else if constexpr (BUF) {
        rsrc.x = qb_lo;
        rsrc.y = __builtin_amdgcn_readfirstlane((qb_hi & 0xffffu) | (4u << 16));
        rsrc.z = __builtin_amdgcn_readfirstlane((uint32_t)(qp * (h + 1) - 1));
        rsrc.w = 0x00020000u;
        asm volatile("" : "+s"(rsrc));
    }
    const float cen = tile_value(tile[own]);
    const float fa = (float)(ROW ? x : y), fl = (float)(ROW ? y : x);
    const float uhi = (float)(w - 1), vhi = (float)(h - 1);
    const bool q8 = (sc->flags & TSAR_FLAG_TEX_FILTER_8BIT) != 0;   // bilinear weights with 8 fractional bits (wave-uniform)
    float sum_src = 0.f, sum_src_src = 0.f, sum_ref_src = 0.f;
    // rows: the table rows of this chunk's taps (scalars, loaded while the previous chunk ran)
    auto chunk = [&](const unsigned short* trow, const uint32_t (&rows)[CH], int c0, float bx, float by, float bz, auto clamp_tag) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        float r[CH], ax[CH], ay[CH], wv[CH];
        uint32_t q[CH], off_last = 0;
        uint64_t q2[CH];                                    // MIX: the tap's four halfs
        float yj0 = 0.f;
        if constexpr (ROW) {
            // the chunk's reference texels, each loaded into bits 31:16 of a register = its fp32 value (pm_core.h, variant bit 3);
            // the wait for them is lut_wait_lds below, after the gathers are on their way
            const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned short*)(trow + 2 * c0);
#pragma unroll
            for (int jj = 0; jj < CH; jj++) asm("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(r[jj]) : "v"(a0), "n"(jj * 4), "v"(bz));
            yj0 = fa + (float)(2 * c0 - rt);
        } else {
#pragma unroll
            for (int jj = 0; jj < CH; jj++) r[jj] = tile_value(trow[min(c0 + jj, rt) * 2 * tw]);
        }
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int jj = 0; jj < CH; jj++) {                   // phase 1: tap positions -> byte offsets; phase 2: gathers
            const float yj = ROW ? yj0 + (float)(2 * jj) : fa + (float)(2 * min(c0 + jj, rt) - rt);
            float X = fma_(H[ROW ? 0 : 1], yj, bx), Y = fma_(H[ROW ? 3 : 4], yj, by), Z = fma_(H[ROW ? 6 : 7], yj, bz);
            if (STRICT) { X += H[2]; Y += H[5]; Z += H[8]; }      // strict: (m[0] x + m[1] y) + m[2], the constant last (pm_core.h view_cost_generic)
            float u, v;
            int iu, iv;
            if (STRICT) {                                   // the oracle's operations: IEEE divides, min/max clamp, floor / subtract
                persp_divide_exact<CLAMP>(X, Y, Z, u, v);   // = X / Z, Y / Z bit for bit (tsar_device_math.h); clamp-free: guard shown by the corner test
                if (CLAMP) {
                    u = fminf(fmaxf(u, 0.0f), uhi);
                    v = fminf(fmaxf(v, 0.0f), vhi);
                }
                // u, v >= 0 (clamped, or inside the image by the corner test): fract = u - floor(u) exactly
                ax[jj] = __builtin_amdgcn_fractf(u);
                ay[jj] = __builtin_amdgcn_fractf(v);
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu) : "v"(u));
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv) : "v"(v));
            } else {
                const float rz = __builtin_amdgcn_rcpf(Z);
                u = X * rz;
                v = Y * rz;
                if (CLAMP) {
                    u = __builtin_amdgcn_fmed3f(u, 0.0f, uhi);
                    v = __builtin_amdgcn_fmed3f(v, 0.0f, vhi);
                }
                ax[jj] = __builtin_amdgcn_fractf(u);
                ay[jj] = __builtin_amdgcn_fractf(v);
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu) : "v"(u));
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv) : "v"(v));
            }
            int lin;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(iv), "s"(qp), "v"(iu));
            if constexpr (MIX) {
                off_last = (uint32_t)lin;
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 idxen" : "=v"(q2[jj]) : "v"(lin), "s"(rsrc));
            } else if constexpr (BUF) {
                off_last = (uint32_t)lin;
                asm volatile("buffer_load_dword %0, %1, %2, 0 idxen" : "=v"(q[jj]) : "v"(lin), "s"(rsrc));
            } else {
                off_last = (uint32_t)lin << 2;
                q[jj] = *(global_u32_ptr)((const char __attribute__((address_space(1)))*)(uintptr_t)(((uint64_t)qb_hi << 32) | qb_lo) + off_last);
            }
        }
        if constexpr (ROW) lut_wait_lds<CH>(r, off_last);
#pragma unroll
        for (int jj = 0; jj < CH; jj++) wv[jj] = lut_weight(lut, rows[jj], r[jj], cen);
        if (q8) {                                           // one scalar branch per chunk; the oracle's rounding (sample_bilinear, pm_core.h)
#pragma unroll
            for (int jj = 0; jj < CH; jj++) {
                ax[jj] = rintf(ax[jj] * 256.0f) * 0.00390625f;
                ay[jj] = rintf(ay[jj] * 256.0f) * 0.00390625f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                  // nothing of phase 3 may move above the last gather
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < CH; jj++) {                   // phase 3: unpack, blend, accumulate
            float t00, t10, t01, t11;
            float s;
            if constexpr (MIX) {
                asm("s_waitcnt vmcnt(%3)" : "+v"(q2[jj]), "+v"(sum_src_src) : "v"(q2[CH - 1]), "n"(CH - 1 - jj));
                const uint32_t lo = (uint32_t)q2[jj], hi = (uint32_t)(q2[jj] >> 32);
                float ta, tb;
                asm("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(ta) : "v"(ax[jj]), "v"(lo));          // ax * d1 + t00: the top row's interpolation
                asm("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(tb) : "v"(ax[jj]), "v"(hi));          // ax * d3 + d2: bottom row minus top row, rounded once
                s = fma_(ay[jj], tb, ta);
            } else {
            if constexpr (BUF)      // the asm-issued gathers return in order: tap jj has CH - 1 - jj behind it (waits chained, pm_core.h)
                asm("s_waitcnt vmcnt(%3)" : "+v"(q[jj]), "+v"(sum_src_src) : "v"(q[CH - 1]), "n"(CH - 1 - jj));
            asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(t00) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t10) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(t01) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(t11) : "v"(q[jj]));
            if (STRICT) {
                const float top = fma_(ax[jj], t10 - t00, t00);
                const float bot = fma_(ax[jj], t11 - t01, t01);
                s = fma_(ay[jj], bot - top, top);
            } else {                                            // fast arithmetic (oracle S7 (6)), see pm_tap_r5.h
                const float d1 = t10 - t00, d2 = t01 - t00, d3 = (t11 - t01) - d1;
                s = fma_(ay[jj], fma_(ax[jj], d3, d2), fma_(ax[jj], d1, t00));
            }
            }
            const float wt = wv[jj];
            const float ws = wt * s;
            sum_src += ws;
            sum_src_src = fma_(ws, s, sum_src_src);
            if (STRICT) sum_ref_src = fma_(wt * r[jj], s, sum_ref_src);   // (w r) s, the oracle's order
            else sum_ref_src = fma_(ws, r[jj], sum_ref_src);              // (w s) r: one multiply fewer per tap
        }
    };
    auto lines = [&](auto clamp_tag) {
        const uint32_t* rp = sc->tap_row;                   // walked linearly: chunks are consecutive in the table
        uint32_t rows[CH];
#pragma unroll
        for (int jj = 0; jj < CH; jj++) rows[jj] = rp[jj];
#pragma unroll 1
        for (int l = 0; l <= rl; l++) {
            const int ol = 2 * l - rl;
            const float xi = fl + (float)ol;
            const float bx = STRICT ? H[0] * xi : fma_(H[ROW ? 1 : 0], xi, H[2]), by = STRICT ? H[3] * xi : fma_(H[ROW ? 4 : 3], xi, H[5]),
                        bz = STRICT ? H[6] * xi : fma_(H[ROW ? 7 : 6], xi, H[8]);
            const unsigned short* trow = tile + own + (ROW ? ol * tw - rt : ol - rt * tw);
#pragma unroll 1
            for (int c0 = 0; c0 <= rt; c0 += CH) {
                rp += CH;
                uint32_t nxt[CH];                           // (the table has slack after its last chunk)
#pragma unroll
                for (int jj = 0; jj < CH; jj++) nxt[jj] = rp[jj];
                chunk(trow, rows, c0, bx, by, bz, clamp_tag);
#pragma unroll
                for (int jj = 0; jj < CH; jj++) rows[jj] = nxt[jj];
            }
        }
    };
    if (need_clamp) lines(std::true_type());
    else lines(std::false_type());
    sum_src *= pr.inv_wsum;
    sum_src_src *= pr.inv_wsum;
    sum_ref_src *= pr.inv_wsum;
    const float var_src = sum_src_src - sum_src * sum_src;
    if (var_src < 1e-5f) return TSAR_MAXCOST;
    const float covar = sum_ref_src - pr.mean_ref * sum_src;
    // both variances are >= 1e-5 here and at most 255^2 (8-bit imagery): their product lies inside sqrt_rsq_exact's range by
    // construction, so the correctly rounded root needs no guard (and none of the six v_cndmask of the compiler's sqrtf)
    const float vrs = sqrt_rsq_exact(pr.var_ref * var_src);
    return fmaxf(0.0f, fminf(TSAR_MAXCOST, 1.0f - covar / vrs));
}
