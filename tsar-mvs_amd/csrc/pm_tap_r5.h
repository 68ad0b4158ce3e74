// pm_tap_r5.h — the production tap loop: pmCost (gipuma.cu:229-298) of one source view for the scripts' window (--blocksize=11:
// radius 5, taps at {-5,-3,-1,1,3,5}^2, scripts/courtyard.sh:10-15) on 8-bit imagery (quad textures), given the hoisted reference
// terms of pm_core.h.  Both arithmetic modes run this one function; included by pm_core.h.
//
// Template switches (the kernels' variant number V keeps naming them in profiles: 114 = none, 122 = D16, 250 = D16 + ROW,
// + 131072 = BUF):
//   STRICT  the oracle's values bit for bit: correctly rounded quotients (persp_divide_exact), min/max clamp, (w r) s, and the
//           oracle's summation order (window columns).  Fast mode: v_rcp_f32 + 2 multiplies, v_med3_f32 clamp, (w s) r.
//   ROW     fast mode only: the window is walked row by row (six taps along x per trip).  A row's six taps of one lane fall into
//           one or two cache lines of the source texture, so where neighbouring lanes' footprints are unrelated (random planes:
//           init, the first sweep) the six gathers of a trip reuse the lines the first one brought into L1; -4 % converged,
//           -18..25 % on random planes.  Changes the summation order of the three tap sums, hence not in strict mode.
//   D16     the line's six reference texels are loaded with ds_read_u16_d16_hi straight into the upper half of a register
//           (= their fp32 value, no convert).  gfx950 runs with SRAM ECC, where a D16 load writes the whole register;
//           tsar_create probes this once (pm_sweep.hip) and the library falls back to D16 = false if it does not hold.
//   BUF     the gathers as structured buffer loads (buffer_load_dword ... idxen) through a stride-4 resource descriptor: the texture
//           addresser scales the element index, the per-tap shift goes away (-0.65 % on a converged launch, +4 ms on the first
//           sweep of a view, so the launcher uses it from the second sweep on in fast mode, the third in strict mode).  No compiler builtin reaches idxen: the loads are
//           issued by asm and their vmcnt waits are written out.
//   MIX     with BUF, fast mode: the gather reads 8 bytes from the view's half-float difference texture (t00, t10 - t00, t01 - t00,
//           t11 - t10 - t01 + t00; plane_kernels.hip build_dquad_kernel) and the fast arithmetic's blend (t00 + ax d1) + ay (d2 + ax d3)
//           is two v_fma_mix_f32 on the halfs in place and one v_fma_f32: no byte converts, no subtractions (-9 issue units of a tap's 34).  Same
//           values bit for bit as the byte-texture form of that blend, which the global-load launches (init, the first two sweeps) keep.
//           (Strict mode can form the reference's blend from the same halfs bit-exactly — t10 - t00 is stored, t01 and t11 - t01 are
//           one exact v_fma_mix_f32 each, -4.5 issue units per tap — and was measured SLOWER, 48.1 -> 51.6 ms per launch: its
//           column-order walk touches six texture rows per lane and trip, and 8-byte entries double that footprint.  Not kept.)
// Always on (each measured, profiles/r01-r02): a line (six taps) per trip in three explicit phases — all six tap positions, all six
// gathers, then unpack / blend / accumulate — so that six gathers are in flight per wave whatever the scheduler decides; the view's
// quad-texture base (border offset folded in) pinned in SGPRs for the whole view; the line's six weights loaded at the top of the
// line with its reference texels; s_setprio 3 while a wave computes tap positions and issues gathers, 0 while it blends; a
// clamp-free loop for waves whose windows project inside the source image.
// BLK: threads per workgroup = stride, in floats, between the weights of consecutive taps of one thread ([tap][thread]).
#pragma once

// DIAG (experiments build only, WRONG RESULTS by construction; the ceilings of profiles/r04): 1 = the MIX body with every gather
// removed (the tap's halfs are synthesised from its element index: the VALU floor of the shipping body), 2 = every gather replaced
// by an 8-byte LDS read at a per-lane address (ds_read_b64: the instruction mix of source patches staged in LDS, before any staging).
template <bool STRICT, bool ROW, bool D16, bool BUF, bool MIX, int BLK, int DIAG = 0>
DEVFN float view_cost_r5(const DevScene* __restrict__ sc, const DevView& vw, const unsigned short* tile, int tw, int own, const float* wts,
                         const PixelRef& pr, int x, int y, const float4& n4) {
    static_assert(!(STRICT && ROW), "the row-wise walk changes the summation order: fast mode only");
    static_assert(!MIX || (BUF && !STRICT), "the half-float difference texture serves the fast arithmetic's blend through buffer loads");
    const int w = sc->w, h = sc->h, qp = sc->quad_pitch;
    const int qorg = (qp + 1) << 2;          // byte offset of quad entry (0 + 1, 0 + 1)
    float H[9];
    if (STRICT) plane_homography(sc->ref, vw, n4, H, sc->k_sparse != 0);
    else plane_homography_fast(sc->ref, vw, n4, H);
    float sum_src = 0.f, sum_src_src = 0.f, sum_ref_src = 0.f;
    // Clamp-free loop: if the four corner taps of every active lane's window land inside the source image with Z > 0 (the window
    // then maps into the convex quadrilateral they span), no tap needs the clamp and the wave runs a tap loop without it.
    // Wave-uniform decision, identical results.  The texture is addressed from entry (1, 1) with an unsigned offset, so the
    // clamp-free loop must never see floor(u) = -1: the corners keep one pixel of margin (rounding moves a tap by ~1e-4 pixel).
    bool need_clamp;
    if (!STRICT) {
        // Fast mode: the same decision from the window's CENTRE and a bound on its extent — one reciprocal instead of four and ~20
        // instructions fewer per hypothesis and view.  With (Xc, Yc, Zc) the centre's homogeneous position, a tap is at
        // (Xc + dX, Yc + dY, Zc + dZ) with |dX| <= a = 5 (|H0| + |H1|), |dY| <= b = 5 (|H3| + |H4|), |dZ| <= c = 5 (|H6| + |H7|); for
        // Zmin = Zc - c > 0 every tap has Z >= Zmin and |u_tap - u_c| = |dX Zc - Xc dZ| / (Z_tap Zc) <= (a Zc + |Xc| c) / (Zmin Zc).
        // The window is inside when the centre keeps that distance (+ the pixel of margin the unsigned addressing needs, + half a
        // pixel for the rounding of this bound itself: its terms are evaluated to ~1e-6 relative on distances of a few pixels and
        // positions of a few thousand).  More conservative than the corner test by the slack of the bound: waves whose windows come
        // within ~2 extents of the border take the clamp loop, which returns the same bits.
        const float xc = (float)x, yc = (float)y;
        const float Xc = fma_(H[1], yc, fma_(H[0], xc, H[2])), Yc = fma_(H[4], yc, fma_(H[3], xc, H[5])), Zc = fma_(H[7], yc, fma_(H[6], xc, H[8]));
        const float a = fabsf(H[0]) + fabsf(H[1]), b = fabsf(H[3]) + fabsf(H[4]), c = fabsf(H[6]) + fabsf(H[7]);
        const float Zmin = fma_(-5.0f, c, Zc);
        const float r = __builtin_amdgcn_rcpf(Zmin * Zc);
        const float r5 = 5.0f * r, rc = Zmin * r;                 // 5 / (Zmin Zc), 1 / Zc
        const float du = fma_(a, Zc, fabsf(Xc) * c) * r5, dv = fma_(b, Zc, fabsf(Yc) * c) * r5;
        const float uc = Xc * rc, vc = Yc * rc;
        const bool inside = Zmin > 0.0f && fminf(uc - du, vc - dv) >= 1.5f && uc + du <= (float)(w - 1) - 1.5f && vc + dv <= (float)(h - 1) - 1.5f;
        need_clamp = !__all(inside);
    } else {
        bool inside = true;
        float zmin = __builtin_inff(), zmax = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float xi = (float)(x + ((c & 1) ? 5 : -5)), yj = (float)(y + ((c & 2) ? 5 : -5));
            const float X = fma_(H[1], yj, fma_(H[0], xi, H[2])), Y = fma_(H[4], yj, fma_(H[3], xi, H[5])), Z = fma_(H[7], yj, fma_(H[6], xi, H[8]));
            const float rz = __builtin_amdgcn_rcpf(Z);
            const float u = X * rz, v = Y * rz;
            inside = inside && Z > 0.0f && u >= 1.0f && u <= (float)(w - 1) - 1.0f && v >= 1.0f && v <= (float)(h - 1) - 1.0f;
            if (STRICT) { zmin = fminf(zmin, Z); zmax = fmaxf(zmax, Z); }
        }
        if (STRICT) {
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
    // quad base + border offset, opaque to the optimiser so that it stays in two SGPRs across the view (the compiler otherwise
    // re-loads it with s_load in every line and waits for it, and for the line's LDS loads, right before issuing the gathers)
    uint32_t qb_lo = 0, qb_hi = 0;
    if (!BUF) {                              // (the buffer-load forms address through the descriptor below: no second pointer load per view)
        const uint64_t qa = (uint64_t)(uintptr_t)vw.quad + (uint32_t)qorg;
        qb_lo = __builtin_amdgcn_readfirstlane((uint32_t)qa);
        qb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(qa >> 32));
        asm volatile("" : "+s"(qb_lo), "+s"(qb_hi));
    }
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    u32x4s rsrc = {0u, 0u, 0u, 0u};
    if (BUF) {
        if (MIX) {                           // 8-byte entries of the difference texture, same pitch and border
            const uint64_t da = (uint64_t)(uintptr_t)vw.dquad + 2 * (uint64_t)(uint32_t)qorg;
            rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)da);
            rsrc.y = __builtin_amdgcn_readfirstlane(((uint32_t)(da >> 32) & 0xffffu) | (8u << 16));  // base[47:32] | stride 8
        } else {
        const uint64_t qa = (uint64_t)(uintptr_t)vw.quad + (uint32_t)qorg;
        rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)qa);
        rsrc.y = __builtin_amdgcn_readfirstlane(((uint32_t)(qa >> 32) & 0xffffu) | (4u << 16));      // base[47:32] | stride 4
        }
        rsrc.z = __builtin_amdgcn_readfirstlane((uint32_t)(qp * (h + 1) - 1));                       // records from entry (1, 1) on
        rsrc.w = 0x00020000u;                                                                         // 32-bit data format (gfx9 family)
        asm volatile("" : "+s"(rsrc));
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // one line of the window: `i` is the column offset (the row offset when ROW) and the six taps run along the other axis
    // unr_tag: the caller's loop over the six lines is fully unrolled (`i` is a constant after inlining): the line's LDS loads then
    // take their line offset as an immediate from loop-invariant base addresses instead of two address additions per line
    auto line6 = [&](int i, auto clamp_tag, auto unr_tag) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        constexpr bool UNR = decltype(unr_tag)::value;
        static_assert(!UNR || (ROW && D16), "the unrolled form is written for the row-wise walk with D16 window loads");
        const float xi = (float)((ROW ? y : x) + i);
        // strict mode: the text's association (m[0] x + m[1] y) + m[2] of getCorrespondingPoint_cu (gipuma.cu:161-171, config.h:150-162),
        // the constant added last (pm_core.h view_cost_generic); fast: folded into the line term (oracle S7 (7))
        const float bx = STRICT ? H[0] * xi : fma_(H[ROW ? 1 : 0], xi, H[2]), by = STRICT ? H[3] * xi : fma_(H[ROW ? 4 : 3], xi, H[5]),
                    bz = STRICT ? H[6] * xi : fma_(H[ROW ? 7 : 6], xi, H[8]);
        const int line = (i + 5) >> 1;                // 0..5: which column (or row) this is
        float rcol[6];
        f32x2 wcol[3];
        {
            // the line's six weights, [tap][thread] layout, tap = 6 * column + row: taps are BLK floats apart = BLK / 64 units of
            // ds_read2st64's 256-byte stride; along a row consecutive taps are 6 taps apart
            constexpr int U = BLK / 64, S = ROW ? 6 : 1;
            const uint32_t wa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)(wts + (UNR ? 0 : (ROW ? line : 6 * line) * BLK));
#pragma unroll
            for (int k = 0; k < 3; k++)
                asm("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(wcol[k]) : "v"(wa), "n"(2 * U * S * k + (UNR ? U * line : 0)), "n"(2 * U * S * k + U * S + (UNR ? U * line : 0)), "v"(bz));
        }
        if (D16) {
            // The loads are invisible to the compiler's waitcnt bookkeeping, which stays correct (LDS returns in order, its own
            // waits only get more conservative); the wait for these six is the asm before their first use below.  Neither asm is
            // volatile (a volatile one fences the gathers and serialises the taps); the unused bz operand keeps the loads inside
            // the line loop instead of being hoisted out of the view and hypothesis loops into 36 live registers.
            // (UNR: base = the window's first row; the window is PM_RW + 10 texels wide in every kernel that runs this loop)
            const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned short*)(UNR ? tile + own - 5 * tw - 5 : ROW ? tile + own + i * tw - 5 : tile + own + i - 5 * tw);
#pragma unroll
            for (int jj = 0; jj < 6; jj++)
                asm("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(rcol[jj]) : "v"(a0), "n"((ROW ? jj * 4 : jj * 2 * (PM_RW + 10) * 2) + (UNR ? line * 2 * (PM_RW + 10) * 2 : 0)), "v"(bz));
        }
        float ax[6], ay[6];
        uint32_t q[6];
        uint64_t q2[6];                              // MIX: the tap's four halfs
        __builtin_amdgcn_s_setprio(3);               // a wave computing tap positions / issuing gathers goes ahead of waves that are blending
#pragma unroll
        for (int jj = 0; jj < 6; jj++) {                        // phase 1: tap positions -> element index; phase 2: gathers
            const float yj = (float)((ROW ? x : y) + 2 * jj - 5);
            float X = fma_(H[ROW ? 0 : 1], yj, bx), Y = fma_(H[ROW ? 3 : 4], yj, by), Z = fma_(H[ROW ? 6 : 7], yj, bz);
            if (STRICT) { X += H[2]; Y += H[5]; Z += H[8]; }
            float u, v;
            // Clamp range.  The oracle clamps to [-1, w] (tex2D at u + .5 with clamp addressing).  The offset is unsigned from entry
            // (1, 1), so floor(u) must be >= 0: clamp to [0, w - 1] instead.  The sample is the same bit for bit: for u in [-1, 0)
            // both texels of the pair are T(0) (edge replication), so the blend returns T(0) whatever the fraction — exactly what
            // u = 0 returns (fraction 0); likewise beyond w - 1, and per axis.
            const float uhi = (float)(w - 1), vhi = (float)(h - 1);
            if (STRICT) {                                       // the oracle's values: correctly rounded quotients, min/max clamp
                persp_divide_exact<CLAMP>(X, Y, Z, u, v);       // clamp-free loop: the operand guard is shown by the corner test
                if (CLAMP) {
                    u = fminf(fmaxf(u, 0.0f), uhi);
                    v = fminf(fmaxf(v, 0.0f), vhi);
                }
            } else {
                const float rz = __builtin_amdgcn_rcpf(Z);
                u = X * rz;
                v = Y * rz;
                if (CLAMP) {
                    u = __builtin_amdgcn_fmed3f(u, 0.0f, uhi);
                    v = __builtin_amdgcn_fmed3f(v, 0.0f, vhi);
                }
            }
            // u, v >= 0 here (clamped to [0, w - 1], or inside the image by the corner test): v_fract_f32 = u - floor(u) exactly (the
            // difference is representable), v_cvt_flr_i32_f32 = (int)floor(u) — the oracle's floor / subtract / convert
            int iu, iv;
            ax[jj] = __builtin_amdgcn_fractf(u);
            ay[jj] = __builtin_amdgcn_fractf(v);
            asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu) : "v"(u));
            asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv) : "v"(v));
            // element index of quad entry (iv + 1, iu + 1) from entry (1, 1): one 24-bit multiply-add
            int lin;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(iv), "s"(qp), "v"(iu));
            if (MIX && DIAG == 1) {
                q2[jj] = ((uint64_t)(uint32_t)~lin << 32) | (uint32_t)lin;      // (two different dwords: identical ones would let the compiler merge the blend's two mixed FMAs)
            } else if (MIX && DIAG == 2) {
                uint32_t la;                                    // inside the workgroup's first 32 KiB of LDS, lanes 16 bytes apart like a staged patch
                asm("v_bfe_u32 %0, %1, 0, 12" : "=v"(la) : "v"(lin));
                asm volatile("ds_read_b64 %0, %1" : "=v"(q2[jj]) : "v"(la << 3));
            } else if (MIX) {
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 idxen" : "=v"(q2[jj]) : "v"(lin), "s"(rsrc));
            } else if (BUF) {
                asm volatile("buffer_load_dword %0, %1, %2, 0 idxen" : "=v"(q[jj]) : "v"(lin), "s"(rsrc));
            } else {                                            // base already holds the border offset: the byte offset is a plain shift
                const uint32_t off2 = (uint32_t)lin << 2;
                q[jj] = *(global_u32_ptr)((const char __attribute__((address_space(1)))*)(uintptr_t)(((uint64_t)qb_hi << 32) | qb_lo) + off2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);           // nothing of phase 3 may move above the last gather
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 6; jj++) {                        // phase 3: unpack, blend, accumulate
            float t00, t10, t01, t11;                           // the four texels: one convert each, no shifts/masks
            if (MIX && DIAG == 1) {
            } else if (MIX && DIAG == 2) {                      // LDS returns in order: tap jj has 5 - jj reads behind it
                if (jj == 0) asm("s_waitcnt lgkmcnt(5)" : "+v"(q2[0]), "+v"(rcol[0]), "+v"(rcol[1]), "+v"(rcol[2]), "+v"(rcol[3]), "+v"(rcol[4]), "+v"(rcol[5]),
                                 "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]) : "v"(q2[5]));
                else asm("s_waitcnt lgkmcnt(%2)" : "+v"(q2[jj]), "+v"(sum_src_src) : "n"(5 - jj), "v"(q2[5]));
            } else if (MIX) {
                if (jj == 0) asm("s_waitcnt vmcnt(5)" : "+v"(q2[0]) : "v"(q2[5]));
                if (jj == 1) asm("s_waitcnt vmcnt(4)" : "+v"(q2[1]), "+v"(sum_src_src) : "v"(q2[5]));
                if (jj == 2) asm("s_waitcnt vmcnt(3)" : "+v"(q2[2]), "+v"(sum_src_src) : "v"(q2[5]));
                if (jj == 3) asm("s_waitcnt vmcnt(2)" : "+v"(q2[3]), "+v"(sum_src_src) : "v"(q2[5]));
                if (jj == 4) asm("s_waitcnt vmcnt(1)" : "+v"(q2[4]), "+v"(sum_src_src) : "v"(q2[5]));
                if (jj == 5) asm("s_waitcnt vmcnt(0)" : "+v"(q2[5]), "+v"(sum_src_src));
            } else if (BUF) {
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
            float s;
            if (MIX) {
                // (t00 + ax d1) + ay (d2 + ax d3) with the halfs read in place: two mixed-precision FMAs on the gathered dwords and one
                // plain FMA, one fp32 rounding each — the same values as the fp32 chain below (the halfs are exact integers)
                const uint32_t lo = (uint32_t)q2[jj], hi = (uint32_t)(q2[jj] >> 32);
                float ta, tb;
                asm("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(ta) : "v"(ax[jj]), "v"(lo));          // ax * d1 + t00: the top row's interpolation
                asm("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(tb) : "v"(ax[jj]), "v"(hi));          // ax * d3 + d2: bottom row minus top row, rounded once
                s = fma_(ay[jj], tb, ta);
            } else {
            asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(t00) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t10) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(t01) : "v"(q[jj]));
            asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(t11) : "v"(q[jj]));
            if (STRICT) {                                       // the reference's blend: two horizontal interpolations, one vertical
                const float top = fma_(ax[jj], t10 - t00, t00);
                const float bot = fma_(ax[jj], t11 - t01, t01);
                s = fma_(ay[jj], bot - top, top);
            } else {                                            // fast arithmetic (oracle S7 (6)): (t00 + ax d1) + ay (d2 + ax d3), exact integer differences
                const float d1 = t10 - t00, d2 = t01 - t00, d3 = (t11 - t01) - d1;
                s = fma_(ay[jj], fma_(ax[jj], d3, d2), fma_(ax[jj], d1, t00));
            }
            }
            // one wait per line, at its first tap: every LDS load of the line (six texels when they are D16 loads, three weight
            // pairs) was issued before the gathers, in order, and has long returned when the first gather does
            if (jj == 0 && DIAG != 2) {
                if (D16)
                    asm("s_waitcnt lgkmcnt(0)" : "+v"(rcol[0]), "+v"(rcol[1]), "+v"(rcol[2]), "+v"(rcol[3]), "+v"(rcol[4]), "+v"(rcol[5]),
                        "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(s));
                else
                    asm("s_waitcnt lgkmcnt(0)" : "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(s));
            }
            const float r = D16 ? rcol[jj] : tile_value(ROW ? tile[own + i * tw + (2 * jj - 5)] : tile[own + (2 * jj - 5) * tw + i]);
            const float wt = wcol[jj >> 1][jj & 1];
            const float ws = wt * s;
            sum_src += ws;
            sum_src_src = fma_(ws, s, sum_src_src);
            if (STRICT) sum_ref_src = fma_(wt * r, s, sum_ref_src);   // (w r) s, the oracle's order
            else sum_ref_src = fma_(ws, r, sum_ref_src);              // (w s) r: one multiply fewer per tap
        }
    };
    if (need_clamp) {
#pragma unroll 1
        for (int i = -5; i <= 5; i += 2) line6(i, std::true_type(), std::false_type());
    } else if constexpr (MIX) {
        // the converged launches' hot path: the six lines unrolled (-6 VALU and the loop's scalar bookkeeping per line; +0.65 %
        // Mpix/s, profiles/r04/ab_unrolled_lines)
#pragma unroll
        for (int i = -5; i <= 5; i += 2) line6(i, std::false_type(), std::true_type());
    } else {
#pragma unroll 1
        for (int i = -5; i <= 5; i += 2) line6(i, std::false_type(), std::false_type());
    }
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

// The variant numbers the production kernels are instantiated with (and which profiles name): bits 1, 4, 5, 6 always set.
// + 2097152 (with 250 | 131072): MIX
// + 4194304 / + 8388608 (experiments build): DIAG 1 / 2 of the MIX body
__host__ __device__ constexpr bool r5_diag_variant(int V) { return V == (250 | 131072 | 2097152 | 4194304) || V == (250 | 131072 | 2097152 | 8388608); }
__host__ __device__ constexpr bool r5_production_variant(int V) { return V == 114 || V == 122 || V == 250 || V == (114 | 131072) || V == (122 | 131072) || V == (250 | 131072) || V == (250 | 131072 | 2097152); }
