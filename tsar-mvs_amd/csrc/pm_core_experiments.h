// pm_core_experiments.h — the tap-loop variants that were measured and rejected, and the diagnostic ones (deliberately wrong
// results: the VALU floor without gathers, the instruction mix of paired gathers, all taps from LDS), kept for the record.
// Compiled only with `make TSAR_EXPERIMENTS=1` and reached through TSAR_VARIANT; the product library carries pm_tap_r5.h only.
// Included by pm_core.h.
#pragma once

// pmCost gipuma.cu:229-298 for one source view, given the hoisted reference terms.
// V selects a code-generation variant of the tap loop (identical arithmetic unless noted):
//   (bit 0, two tap columns per trip, was measured and removed: 135 VGPRs, one wave of occupancy lost)
//   bit 1: fast mode only — clamp with v_med3_f32 and take the fraction with v_fract_f32
//          (differs from floor/subtract only for u in (-2^-24, 0), where fract saturates below 1)
//   bit 2: experiment — no gather (texel bits synthesised from the address): the VALU floor of the kernel
//   bit 3: fast mode, radius 5 — reference-window texels loaded with ds_read_u16_d16_hi (no convert instruction)
//   bit 4: fast mode, radius 5 — clamp-free tap loop for waves whose windows project inside the source image (-1 %)
//   bit 5: fast mode, radius 5 — s_setprio 3 while a wave computes tap positions and issues its gathers, 0 while it
//          blends: gathers enter the memory system earlier (-1.1 %; the opposite assignment costs +2.6 %)
//   bit 6: radius 5 — the view's quad-texture base (border offset folded in) is pinned in an SGPR pair for the whole view
//          (the compiler otherwise re-loads it with s_load in every column and waits for it, and for the column's LDS
//          loads, right before issuing the gathers), the tap's byte offset is a plain shift, and the column's six
//          bilateral weights are loaded at the top of the column with its reference texels instead of one LDS round trip
//          per pair of taps inside the blend phase
//   bit 7: fast mode, radius 5 — the window is walked ROW by row (six taps along x per trip) instead of column by column.  A
//          row's six taps of one lane fall into one or two cache lines of the source texture, so in the random-plane regime
//          (init, the first sweep: neighbouring lanes' footprints are unrelated and L2 bandwidth is the bound) the six gathers
//          of a trip reuse the lines the first one brought into L1.  Changes the summation order of the three tap sums, hence
//          fast mode only; strict keeps the oracle's column order.
//   bit 8: experiment (wrong results): the instruction mix of pairing two taps into one 16-byte gather
//   bit 9: radius 5, with bits 3 and 6 — gathers of line t+1 issued before line t is blended (two register sets)
//   bit 17: with bit 6 — gathers as structured buffer loads (idxen, stride 4): the addresser scales the element index, the per-tap
//           shift goes away (-0.65 %); issued by asm, so their vmcnt waits are written out
//   bit 10: any window, 8-bit imagery — view_cost_lut (pm_core_lut.h) instead of this function; bits 11-13 = taps per chunk
// BLK: threads per workgroup = stride, in floats, between the weights of consecutive taps of one thread ([tap][thread])
template <int HR, bool STRICT, bool QUAD, int V = 0, int BLK = PM_BLOCK>
DEVFN float view_cost_variants(const DevScene* __restrict__ sc, const DevView& vw, const typename TileOf<QUAD>::type* tile, int tw, int own, const float* wts,
                      const PixelRef& pr, int x, int y, const float4& n4) {
    const int hr = HR > 0 ? HR : sc->hrad, vr = HR > 0 ? HR : sc->vrad;
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    const int qorg = (qp + 1) << 2;          // byte offset of quad entry (0 + 1, 0 + 1)
    float H[9];
    if (STRICT) plane_homography(sc->ref, vw, n4, H);
    else plane_homography_fast(sc->ref, vw, n4, H);
    float sum_src = 0.f, sum_src_src = 0.f, sum_ref_src = 0.f;
    int tap = 0;
    constexpr bool FAST6 = QUAD && (V & 2) && HR == 5;   // the production tap loop: 8-bit quad texture, radius 5 (both arithmetic modes)
    // Variant bit 4: if the four corner taps of every active lane's window land inside the source image with Z > 0
    // (the window then maps into the convex quadrilateral they span), no tap needs the clamp and the wave runs a tap
    // loop without the two v_med3_f32.  Wave-uniform decision, identical results.
    bool need_clamp = true;
    if (FAST6 && (V & 16) && (V & 524288)) {
        // variant bit 19: the same decision without the four reciprocals.  For Z > 0, lo <= X / Z <= hi  <=>  X - lo Z >= 0 and
        // hi Z - X >= 0, so each corner contributes five margins (four fused multiply-adds and Z itself) and the window is inside
        // when the smallest of the twenty is positive.  v_min drops NaN operands, so non-finite homographies are caught up front:
        // sum |H_i| < 1e30 also rules out overflow of the corner terms (|x|, |y| < 2^23).
        const float mg = (V & 64) ? 1.0f : 0.0f;
        const float uh = (float)(w - 1) - mg, vh = (float)(h - 1) - mg;
        float sh = fabsf(H[0]);
#pragma unroll
        for (int e = 1; e < 9; e++) sh += fabsf(H[e]);
        float m = __builtin_inff();
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float yj = (float)(y + (r ? 5 : -5));
            const float rx = fma_(H[1], yj, H[2]), ry = fma_(H[4], yj, H[5]), rz = fma_(H[7], yj, H[8]);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float xi = (float)(x + (c ? 5 : -5));
                const float X = fma_(H[0], xi, rx), Y = fma_(H[3], xi, ry), Z = fma_(H[6], xi, rz);
                const float a = fma_(-mg, Z, X), b = fma_(uh, Z, -X), c2 = fma_(-mg, Z, Y), d = fma_(vh, Z, -Y);
                m = fminf(fminf(m, fminf(a, b)), fminf(fminf(c2, d), Z));
            }
        }
        need_clamp = !__all(m > 0.0f && sh < 1e30f);
    } else if (FAST6 && (V & 16)) {
        bool inside = true;
        float zmin = __builtin_inff(), zmax = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float xi = (float)(x + ((c & 1) ? 5 : -5)), yj = (float)(y + ((c & 2) ? 5 : -5));
            const float X = fma_(H[1], yj, fma_(H[0], xi, H[2])), Y = fma_(H[4], yj, fma_(H[3], xi, H[5])), Z = fma_(H[7], yj, fma_(H[6], xi, H[8]));
            const float rz = __builtin_amdgcn_rcpf(Z);
            const float u = X * rz, v = Y * rz;
            // variant bit 6 addresses the texture from entry (1, 1) with an unsigned offset: its clamp-free loop must never see
            // floor(u) = -1, so the corners keep one pixel of margin (rounding moves a tap by ~1e-4 pixel at most)
            const float mg = (V & 64) ? 1.0f : 0.0f;
            inside = inside && Z > 0.0f && u >= mg && u <= (float)(w - 1) - mg && v >= mg && v <= (float)(h - 1) - mg;
            if (STRICT) { zmin = fminf(zmin, Z); zmax = fmaxf(zmax, Z); }
        }
        if (STRICT && (V & 64)) {
            // The clamp-free loop of strict mode also drops the per-tap operand guard of persp_divide_exact, so "inside" must imply
            // that X, Y, Z of EVERY tap lie in [2^-20, 2^38].  With cm >= |x|, |y| of any tap: Z is affine in the tap position, so at
            // every tap it lies between the corner values up to the rounding of its three-term evaluation, dZ <= 3 * 2^-24 * sz with
            // sz = (|H6| + |H7|) cm + |H8|.  sz cm <= 2^19 zmin bounds dZ / Z by 3 * 2^-5 / cm <= 1.2 % (cm >= 8), so Z stays in
            // [2^-19, 2^18] for zmin >= 2^-18, zmax <= 2^17.  u = X / Z of a tap lies in the hull of the corners' true u (Z > 0: the
            // map is projective), which are >= 1 - 0.15: computed u >= 1, and a computed corner is off by u dZ / Z <= cm * 3 * 2^-24
            // * 2^19 / cm = 0.094 plus dX / Z <= 3 * 2^-24 * sx / zmin <= 0.047 for sx = (|H0| + |H1|) cm + |H2| <= 2^18 zmin.  Hence
            // X >= 0.8 zmin >= 2^-20 and |X| <= sx <= 2^35; the same for Y.
            const float cm = (float)(max(w, h) + 32);
            const float sz = fma_(fabsf(H[6]) + fabsf(H[7]), cm, fabsf(H[8]));
            const float sx = fma_(fabsf(H[0]) + fabsf(H[1]), cm, fabsf(H[2]));
            const float sy = fma_(fabsf(H[3]) + fabsf(H[4]), cm, fabsf(H[5]));
            inside = inside && zmin >= 3.814697265625e-06f && zmax <= 131072.0f && sz * cm <= 524288.0f * zmin && fmaxf(sx, sy) <= 262144.0f * zmin;
        }
        need_clamp = !__all(inside);
    }
    // One window column (six taps) of the production loop, written in three explicit phases — all six tap positions,
    // then all six gathers, then unpack / blend / accumulate — so that six gathers are in flight per wave whatever
    // the instruction scheduler decides (it keeps source order when a reordering would cost registers).
    // variant bit 6: quad base + border offset, opaque to the optimiser so that it stays in two SGPRs across the view
    uint32_t qb_lo = 0, qb_hi = 0;
    if (V & 64) {
        const uint64_t qa = (uint64_t)(uintptr_t)vw.quad + (uint32_t)qorg;
        qb_lo = __builtin_amdgcn_readfirstlane((uint32_t)qa);
        qb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(qa >> 32));
        asm volatile("" : "+s"(qb_lo), "+s"(qb_hi));
    }
    // variant bit 17 (with bit 6; production since round 2): the gather is a structured buffer load (buffer_load_dword ... idxen) through a
    // resource descriptor of stride 4 — the texture addresser multiplies the element index, so the per-tap shift goes away
    // (one VALU instruction of ~26).  The loads are issued by asm (no compiler builtin reaches idxen), so their vmcnt waits are
    // written out in phase 3.
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    u32x4s rsrc = {0u, 0u, 0u, 0u};
    if (V & 131072) {
        const uint64_t qa = (uint64_t)(uintptr_t)vw.quad + (uint32_t)qorg;
        rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)qa);
        rsrc.y = __builtin_amdgcn_readfirstlane(((uint32_t)(qa >> 32) & 0xffffu) | (4u << 16));      // base[47:32] | stride 4
        rsrc.z = __builtin_amdgcn_readfirstlane((uint32_t)(qp * (h + 1) - 1));                       // records from entry (1, 1) on
        rsrc.w = 0x00020000u;                                                                         // 32-bit data format (gfx9 family)
        asm volatile("" : "+s"(rsrc));
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr bool ROW = (V & 128) && !STRICT;     // bit 7: `i` below is then the row offset and the six taps run along x
    auto column_fast = [&](int i, auto clamp_tag) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        const float xi = (float)((ROW ? y : x) + i);
        const float bx = fma_(H[ROW ? 1 : 0], xi, H[2]), by = fma_(H[ROW ? 4 : 3], xi, H[5]), bz = fma_(H[ROW ? 7 : 6], xi, H[8]);
        const int line = (i + 5) >> 1;                // 0..5: which column (or row) this is
        float rcol[6];
        f32x2 wcol[3];
        if (V & 64) {
            // the line's six weights, [tap][thread] layout, tap = 6 * column + row: taps are BLK floats apart = BLK / 64 units of
            // ds_read2st64's 256-byte stride; along a row consecutive taps are 6 taps apart
            constexpr int U = BLK / 64, S = ROW ? 6 : 1;
            const uint32_t wa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)(wts + (ROW ? line : 6 * line) * BLK);
#pragma unroll
            for (int k = 0; k < 3; k++)
                asm("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(wcol[k]) : "v"(wa), "n"(2 * U * S * k), "n"(2 * U * S * k + U * S), "v"(bz));
        }
        if (V & 8) {
            // the column's six reference texels, each loaded into bits 31:16 of a register = its fp32 value.  gfx950 runs
            // with SRAM ECC, where a D16 load writes the whole register (zeros in the other half); tsar_create probes
            // this once and falls back to the variant without bit 3 if it does not hold.  The loads are invisible to the
            // compiler's waitcnt bookkeeping, which stays correct (LDS returns in order, its own waits only get more
            // conservative); the wait for these six is the asm before their first use below.  Neither asm is volatile
            // (a volatile one fences the gathers and serialises the taps); the unused bz operand keeps the loads inside
            // the column loop instead of being hoisted out of the view and hypothesis loops into 36 live registers.
            const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned short*)(ROW ? tile + own + i * tw - 5 : tile + own + i - 5 * tw);
#pragma unroll
            for (int jj = 0; jj < 6; jj++)
                asm("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(rcol[jj]) : "v"(a0), "n"(ROW ? jj * 4 : jj * 2 * (PM_RW + 10) * 2), "v"(bz));
        }
        float ax[6], ay[6];
        uint32_t q[6];
        if (V & 32) __builtin_amdgcn_s_setprio(3);               // a wave computing tap positions / issuing gathers goes ahead of waves that are blending
#pragma unroll
        for (int jj = 0; jj < 6; jj++) {                        // phase 1: tap positions -> byte offsets; phase 2: gathers
            const float yj = (float)((ROW ? x : y) + 2 * jj - 5);
            const float X = fma_(H[ROW ? 0 : 1], yj, bx), Y = fma_(H[ROW ? 3 : 4], yj, by), Z = fma_(H[ROW ? 6 : 7], yj, bz);
            float u, v;
            int iu, iv;
            // Clamp range.  The oracle clamps to [-1, w] (tex2D at u + .5 with clamp addressing).  With variant bit 6 the byte
            // offset is unsigned from entry (1, 1), so floor(u) must be >= 0: clamp to [0, w - 1] instead.  The sample is the same
            // bit for bit: for u in [-1, 0) both texels of the pair are T(0) (edge replication), so the blend returns T(0) whatever
            // the fraction — exactly what u = 0 returns (fraction 0); likewise beyond w - 1, and per axis.
            const float ulo = (V & 64) ? 0.0f : -1.0f, uhi = (V & 64) ? (float)(w - 1) : (float)w, vhi = (V & 64) ? (float)(h - 1) : (float)h;
            if (STRICT) {                                       // the oracle's values: correctly rounded quotients, min/max clamp, floor / subtract
                persp_divide_exact<CLAMP || !(V & 64) || !(V & 16)>(X, Y, Z, u, v);   // clamp-free loop: guard shown by the corner test
                if (CLAMP) {
                    u = fminf(fmaxf(u, ulo), uhi);
                    v = fminf(fmaxf(v, ulo), vhi);
                }
                if (V & 64) {
                    // u, v >= 0 here (clamped to [0, w - 1], or inside the image by the corner test): v_fract_f32 = u - floor(u)
                    // exactly (the difference is representable), v_cvt_flr_i32_f32 = (int)floor(u)
                    ax[jj] = __builtin_amdgcn_fractf(u);
                    ay[jj] = __builtin_amdgcn_fractf(v);
                    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu) : "v"(u));
                    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv) : "v"(v));
                } else {
                    const float fu = floorf(u), fv = floorf(v);
                    ax[jj] = u - fu;
                    ay[jj] = v - fv;
                    iu = (int)fu;
                    iv = (int)fv;
                }
            } else {
                const float rz = __builtin_amdgcn_rcpf(Z);
                u = X * rz;
                v = Y * rz;
                if (CLAMP) {
                    u = __builtin_amdgcn_fmed3f(u, ulo, uhi);
                    v = __builtin_amdgcn_fmed3f(v, ulo, vhi);
                }
                ax[jj] = __builtin_amdgcn_fractf(u);
                ay[jj] = __builtin_amdgcn_fractf(v);
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu) : "v"(u));   // floor + convert in one instruction each
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv) : "v"(v));
            }
            // byte offset of quad entry (iv + 1, iu + 1): one 24-bit multiply-add, one shift-add; the two +1 are
            // folded into the uniform constant (qp + 1) * 4
            int lin;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(iv), "s"(qp), "v"(iu));
            uint32_t off = 0;
            if (!(V & 64)) asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(lin), "s"(qorg));
            if ((V & 131072) && (V & 262144)) {
                q[jj] = (uint32_t)lin;                          // experiment (bit 18): the six loads issued back to back after the loop
            } else if (V & 131072) {
                asm volatile("buffer_load_dword %0, %1, %2, 0 idxen" : "=v"(q[jj]) : "v"(lin), "s"(rsrc));
            } else if (V & 64) {                                // base already holds the border offset: the byte offset is a plain shift
                const uint32_t off2 = (uint32_t)lin << 2;
                if ((V & 256) && (jj & 1)) {
                    // EXPERIMENT (TSAR_VARIANT=506, wrong results): the upper bound of pairing two taps of a row into one wide gather
                    // — odd taps issue no load, a 16-byte load replaces the even tap's; the odd tap's offset arithmetic stands
                    // in for the dword-select instructions a real pairing would need
                    q[jj] = q[jj - 1] + off2;
                } else if (V & 256) {
                    typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
                    const u32x4a4 wide = *(const u32x4a4 __attribute__((address_space(1)))*)((const char __attribute__((address_space(1)))*)(uintptr_t)(((uint64_t)qb_hi << 32) | qb_lo) + off2);
                    q[jj] = wide.x ^ (wide.y & wide.z & wide.w & 0x01010101u);
                } else if (V & 4) {
                    q[jj] = off2 * 2654435761u;                 // EXPERIMENT (TSAR_VARIANT=254, wrong results): no gather at all -> the VALU floor of the production loop
                } else if (V & 1048576) {
                    // EXPERIMENT (TSAR_VARIANT=1048826, wrong results): every gather replaced by a 4-byte LDS read at an address derived
                    // from the tap's offset (inside the workgroup's weight table) -> the ceiling of a source patch staged in LDS,
                    // before any staging cost
                    const uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)wts + (off2 & 0x1ffcu);
                    asm volatile("ds_read_b32 %0, %1" : "=v"(q[jj]) : "v"(la));
                } else
                q[jj] = *(global_u32_ptr)((const char __attribute__((address_space(1)))*)(uintptr_t)(((uint64_t)qb_hi << 32) | qb_lo) + off2);
            } else if (V & 4) q[jj] = off * 2654435761u;        // experiment only (TSAR_VARIANT=6): no gather, same arithmetic -> the VALU floor
            else q[jj] = *(global_u32_ptr)((const char __attribute__((address_space(1)))*)vw.quad + off);
        }
        if ((V & 131072) && (V & 262144)) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 6; jj++) asm volatile("buffer_load_dword %0, %0, %1, 0 idxen" : "+v"(q[jj]) : "s"(rsrc));
        }
        if (V & 16) __builtin_amdgcn_sched_barrier(0);           // nothing of phase 3 may move above the last gather
        if (V & 32) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int jj = 0; jj < 6; jj++) {                        // phase 3: unpack, blend, accumulate
            float t00, t10, t01, t11;                           // the four texels: one convert each, no shifts/masks
            if (V & 131072) {
                // the asm-issued gathers return in order: tap jj has 5 - jj behind it.  Not volatile (a volatile wait is
                // scheduled with the loads, ahead of every blend); the q[5] input keeps each wait behind the issue of the last load, the
                // accumulator behind the previous tap's blend
                if (jj == 0) asm("s_waitcnt vmcnt(5)" : "+v"(q[0]) : "v"(q[5]));
                if (jj == 1) asm("s_waitcnt vmcnt(4)" : "+v"(q[1]), "+v"(sum_src_src) : "v"(q[5]));
                if (jj == 2) asm("s_waitcnt vmcnt(3)" : "+v"(q[2]), "+v"(sum_src_src) : "v"(q[5]));
                if (jj == 3) asm("s_waitcnt vmcnt(2)" : "+v"(q[3]), "+v"(sum_src_src) : "v"(q[5]));
                if (jj == 4) asm("s_waitcnt vmcnt(1)" : "+v"(q[4]), "+v"(sum_src_src) : "v"(q[5]));
                if (jj == 5) asm("s_waitcnt vmcnt(0)" : "+v"(q[5]), "+v"(sum_src_src));
            }
            if ((V & 64) && (V & 1048576))     // the LDS-read experiment: LDS returns in order, tap jj has 5 - jj reads behind it
                asm("s_waitcnt lgkmcnt(%2)" : "+v"(q[jj]), "+v"(sum_src_src) : "n"(5 - jj), "v"(q[5]));
            float s;
            {
            asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(t00) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t10) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(t01) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(t11) : "v"(q[jj]));
            const float top = fma_(ax[jj], t10 - t00, t00);
            const float bot = fma_(ax[jj], t11 - t01, t01);
            s = fma_(ay[jj], bot - top, top);
            }
            float r;
            if (V & 64) {
                // one wait per column, at its first tap: every LDS load of the column (six texels when they are D16 loads,
                // three weight pairs) was issued before the gathers, in order, and has long returned when the first gather does
                if (jj == 0) {
                    if (V & 8)
                        asm("s_waitcnt lgkmcnt(0)" : "+v"(rcol[0]), "+v"(rcol[1]), "+v"(rcol[2]), "+v"(rcol[3]), "+v"(rcol[4]), "+v"(rcol[5]),
                            "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(s));
                    else
                        asm("s_waitcnt lgkmcnt(0)" : "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(s));
                }
                r = (V & 8) ? rcol[jj] : tile_value(ROW ? tile[own + i * tw + (2 * jj - 5)] : tile[own + (2 * jj - 5) * tw + i]);
            } else if (V & 8) {
                // tied to s so that the wait cannot be scheduled ahead of the gather's return, by which time the LDS
                // loads issued at the top of the column have long completed
                asm("s_waitcnt lgkmcnt(0)" : "+v"(rcol[jj]), "+v"(s));
                r = rcol[jj];
            } else {
                r = tile_value(tile[own + (2 * jj - 5) * tw + i]);
            }
            const float wt = (V & 64) ? wcol[jj >> 1][jj & 1] : wts[(tap + jj) * BLK];
            const float ws = wt * s;
            sum_src += ws;
            sum_src_src = fma_(ws, s, sum_src_src);
            if (STRICT) sum_ref_src = fma_(wt * r, s, sum_ref_src);   // (w r) s, the oracle's order
            else sum_ref_src = fma_(ws, r, sum_ref_src);              // (w s) r: one multiply fewer per tap
        }
        tap += 6;
    };
    // variant bit 9 (EXPERIMENT, built with TSAR_EXPERIMENTS only; correct results, slower): the same line, split in two so that
    // the six gathers of line t+1 are issued BEFORE line t is blended — a wave then always has six to twelve gathers in flight and
    // its own address arithmetic covers part of their latency, instead of leaving all of it to the other three waves of the SIMD.
    // Two register sets (Trip) alternate.  Measured (profiles/r02): 128 VGPRs only with 56 spills around the tap loop, 40.5 ms
    // (six lines written out) / 41.5 ms (rolled, two lines per trip) against 38.45 ms: at four waves per SIMD the gather latency is
    // already covered, and the second register set costs more than it hides.  Needs bits 3 and 6.
    struct Trip { uint32_t q[6]; float ax[6], ay[6]; };
    auto issue_line = [&](int i, Trip& T, auto clamp_tag) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        const float xi = (float)((ROW ? y : x) + i);
        const float bx = fma_(H[ROW ? 1 : 0], xi, H[2]), by = fma_(H[ROW ? 4 : 3], xi, H[5]), bz = fma_(H[ROW ? 7 : 6], xi, H[8]);
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int jj = 0; jj < 6; jj++) {
            const float yj = (float)((ROW ? x : y) + 2 * jj - 5);
            const float X = fma_(H[ROW ? 0 : 1], yj, bx), Y = fma_(H[ROW ? 3 : 4], yj, by), Z = fma_(H[ROW ? 6 : 7], yj, bz);
            float u, v;
            int iu, iv;
            if (STRICT) {
                u = X / Z;
                v = Y / Z;
                if (CLAMP) {
                    u = fminf(fmaxf(u, 0.0f), (float)(w - 1));
                    v = fminf(fmaxf(v, 0.0f), (float)(h - 1));
                }
                const float fu = floorf(u), fv = floorf(v);
                T.ax[jj] = u - fu;
                T.ay[jj] = v - fv;
                iu = (int)fu;
                iv = (int)fv;
            } else {
                const float rz = __builtin_amdgcn_rcpf(Z);
                u = X * rz;
                v = Y * rz;
                if (CLAMP) {
                    u = __builtin_amdgcn_fmed3f(u, 0.0f, (float)(w - 1));
                    v = __builtin_amdgcn_fmed3f(v, 0.0f, (float)(h - 1));
                }
                T.ax[jj] = __builtin_amdgcn_fractf(u);
                T.ay[jj] = __builtin_amdgcn_fractf(v);
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu) : "v"(u));
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv) : "v"(v));
            }
            int lin;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(iv), "s"(qp), "v"(iu));
            T.q[jj] = *(global_u32_ptr)((const char __attribute__((address_space(1)))*)(uintptr_t)(((uint64_t)qb_hi << 32) | qb_lo) + ((uint32_t)lin << 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto blend_line = [&](int i, Trip& T) {
        const int line = (i + 5) >> 1;
        float rcol[6];
        f32x2 wcol[3];
        {   // the line's nine LDS loads; H[2] (changes with every view and hypothesis) pins them to this evaluation
            constexpr int U = BLK / 64, S = ROW ? 6 : 1;
            const uint32_t wa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)(wts + (ROW ? line : 6 * line) * BLK);
#pragma unroll
            for (int k = 0; k < 3; k++)
                asm("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(wcol[k]) : "v"(wa), "n"(2 * U * S * k), "n"(2 * U * S * k + U * S), "v"(H[2]));
            const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned short*)(ROW ? tile + own + i * tw - 5 : tile + own + i - 5 * tw);
#pragma unroll
            for (int jj = 0; jj < 6; jj++)
                asm("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(rcol[jj]) : "v"(a0), "n"(ROW ? jj * 4 : jj * 2 * (PM_RW + 10) * 2), "v"(H[2]));
        }
#pragma unroll
        for (int jj = 0; jj < 6; jj++) {
            float t00, t10, t01, t11;
            asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(t00) : "v"(T.q[jj]));
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t10) : "v"(T.q[jj]));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(t01) : "v"(T.q[jj]));
            asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(t11) : "v"(T.q[jj]));
            const float top = fma_(T.ax[jj], t10 - t00, t00);
            const float bot = fma_(T.ax[jj], t11 - t01, t01);
            float s = fma_(T.ay[jj], bot - top, top);
            if (jj == 0)
                asm("s_waitcnt lgkmcnt(0)" : "+v"(rcol[0]), "+v"(rcol[1]), "+v"(rcol[2]), "+v"(rcol[3]), "+v"(rcol[4]), "+v"(rcol[5]),
                    "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(s));
            const float r = rcol[jj];
            const float wt = wcol[jj >> 1][jj & 1];
            const float ws = wt * s;
            sum_src += ws;
            sum_src_src = fma_(ws, s, sum_src_src);
            if (STRICT) sum_ref_src = fma_(wt * r, s, sum_ref_src);
            else sum_ref_src = fma_(ws, r, sum_ref_src);
        }
    };
    auto pipelined_lines = [&](auto clamp_tag) {
        Trip A, B;
        issue_line(-5, A, clamp_tag);
#pragma unroll 1
        for (int i = -5; i <= 3; i += 4) {          // two lines per trip: the hot code stays ~2.5 KB (six unrolled lines cost the i-cache more than they saved)
            issue_line(i + 2, B, clamp_tag);
            blend_line(i, A);
            if (i < 3) issue_line(i + 4, A, clamp_tag);
            blend_line(i + 2, B);
        }
    };
    // any window, both arithmetic modes, float or quad images: one tap at a time in the oracle's order
    auto column = [&](int i) {
        const float xi = (float)(x + i);
        const float bx = fma_(H[0], xi, H[2]), by = fma_(H[3], xi, H[5]), bz = fma_(H[6], xi, H[8]);
#pragma unroll
        for (int j = -vr; j <= vr; j += 2) {
            const float yj = (float)(y + j);
            const float X = fma_(H[1], yj, bx), Y = fma_(H[4], yj, by), Z = fma_(H[7], yj, bz);
            float u, v;
            if (STRICT) {
                persp_divide_exact<true>(X, Y, Z, u, v);
            } else {
                const float rz = __builtin_amdgcn_rcpf(Z);
                u = X * rz;
                v = Y * rz;
            }
            const float s = sample_bilinear<QUAD>(vw, w, h, qp, u, v, (sc->flags & TSAR_FLAG_TEX_FILTER_8BIT) != 0);
            const float r = tile_value(tile[own + j * tw + i]);
            const float wt = wts[tap * BLK];
            const float ws = wt * s;
            sum_src += ws;
            sum_src_src = fma_(ws, s, sum_src_src);
            if (STRICT) sum_ref_src = fma_(wt * r, s, sum_ref_src);      // (w r) s, the oracle's order
            else sum_ref_src = fma_(ws, r, sum_ref_src);
            ++tap;
        }
    };
    if (FAST6 && (V & 512)) {
        if (need_clamp) pipelined_lines(std::true_type());
        else pipelined_lines(std::false_type());
    } else if (FAST6) {
        if (need_clamp) {
#pragma unroll 1
            for (int i = -5; i <= 5; i += 2) column_fast(i, std::true_type());
        } else {
#pragma unroll 1
            for (int i = -5; i <= 5; i += 2) column_fast(i, std::false_type());
        }
    } else {
#pragma unroll 1
        for (int i = -hr; i <= hr; i += 2) column(i);
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

