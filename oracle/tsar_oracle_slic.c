/*
 * tsar_oracle_slic.c — CPU ORACLE of the gSLICr superpixel segmentation as the reference drives it
 * (reference gSLICr_Lib/engines/gSLICr_seg_engine.cpp:30-44, gSLICr_seg_engine_GPU.cu,
 * gSLICr_seg_engine_shared.h; settings main.cpp:608-615).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY: this is the one row whose oracle is PINNED BY THE REFERENCE ITSELF.  gSLICr_seg_engine_shared.h:7-204 is host-compilable
 * as it stands (the reference's own -DCOMPILE_WITHOUT_CUDA switch, ORUtils/MemoryBlock.h:7), so `make -C oracle ref` builds
 * oracle/_ref/libslic_ref.so from the sources where they lie (oracle/ref_harness/slic_ref.cpp only loops the reference's per-pixel
 * functions over arrays) and tests/golden/make_slic_ref_golden.py records its outputs in tests/golden/slic_ref.npz.
 * tests/test_slic_reference_golden.py holds every function below to them: rgb2xyz, the linear branch of rgb2CIELab,
 * init_cluster_centers_shared (both branches), compute_slic_distance, find_center_association_shared,
 * finalize_reduction_result_shared and supress_local_lable BIT FOR BIT / label for label.  NOT reachable by a host compiler and
 * therefore still unpinned: Update_Cluster_Center_device (GPU.cu:260-357, restated in block_partial below) and CUDA's own pow().
 *
 * pow(x, 1.0f / 3.0f) of rgb2CIELab (shared.h:41-46): a libm call whose result depends on the libm (CUDA's libdevice powf on the
 * reference's GPU, glibc's powf in the host build of the reference).  Here it is orc_pow_third: the CORRECTLY ROUNDED fp32 value of
 * x^(0.3333333432674407958984375) — the exponent the reference passes is the float nearest 1/3, not 1/3 — built from IEEE fp64
 * operations only, so the HIP kernel runs the same sequence and agrees bit for bit.  That it IS correctly rounded is enumerated,
 * not argued: orc_pow_third_check compares it with powl (64-bit mantissa) rounded to fp32 on every argument the function can
 * receive from the 2^24 8-bit colours: 50 329 213 evaluations, 0 mismatches.  glibc 2.35's powf — what the reference compiled here
 * calls — differs from the correctly rounded value on 34 835 of them (0.07 %), by one ulp; rounds 1–4 used a Newton CUBE root here,
 * which differs from the reference-compiled value in 15 % of the evaluations (51 % of the Lab components).
 * The block reduction of Update_Cluster_Center_device is restated in its exact tree order so float sums agree bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float cx, cy; float col[4]; int id, n; } spixel;

/* x^(1.0f / 3.0f) correctly rounded to fp32, for x in [2^-10, 4): the cube root in double-double (division-free Newton on the
 * inverse cube root in fp64 — four steps from a 5 % seed; three already pass the enumeration — then one step on the exact residual
 * c^3 - x formed with fma), times x^delta with delta = (double)(1.0f / 3.0f) - 1/3 = 2^-25 / 3 (rounded), for
 * which ln x is needed to ~1e-10 only.  One rounding to fp64, then one to fp32; see the header for the enumeration. */
float orc_pow_third(float xf) {
    const double x = (double)xf;
    union { float f; uint32_t u; } s;
    s.f = xf;
    s.u = s.u / 3u + 0x2a5137a0u;                                               /* cube-root seed within 5 % */
    double y = (double)(1.0f / s.f);                                            /* -> seed of x^(-1/3) */
    for (int i = 0; i < 4; i++) y = y * ((4.0 - x * (y * y * y)) * (1.0 / 3.0));   /* Newton on y^-3 = x: no division */
    const double c = x * (y * y);
    const double c2 = c * c, e2 = fma(c, c, -c2);
    const double c3 = c2 * c, e3 = fma(c2, c, -c3);
    const double r = (c3 - x) + fma(e2, c, e3);                                 /* c^3 - x, exact to ~2^-100 */
    const double lo = -r / (3.0 * c2);
    union { double d; uint64_t u; } v;
    v.d = x;
    const int k = (int)((v.u >> 52) & 0x7ff) - 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;                /* m in [1, 2) */
    const double t = (v.d - 1.0) / (v.d + 1.0), t2 = t * t;                     /* ln m = 2 atanh t */
    double p = 1.0 / 13.0;
    p = fma(p, t2, 1.0 / 11.0);
    p = fma(p, t2, 1.0 / 9.0);
    p = fma(p, t2, 1.0 / 7.0);
    p = fma(p, t2, 1.0 / 5.0);
    p = fma(p, t2, 1.0 / 3.0);
    p = fma(p, t2, 1.0);
    const double lnx = fma((double)k, 0.6931471805599453, 2.0 * t * p);
    const double delta = (double)(1.0f / 3.0f) - 1.0 / 3.0;
    const double u = delta * lnx;
    const double q = fma(0.5 * u, u, u);                                        /* expm1(u), u ~ 5e-8 */
    return (float)(c + fma(c, q, lo));
}
/* the enumeration: every argument rgb2CIELab can hand to pow() from an 8-bit colour, against powl rounded to fp32.
 * out[0] = evaluations, out[1] = mismatches against the correctly rounded value, out[2] = evaluations on which this host's powf
 * (what the reference compiled here calls) differs from it. */
void orc_pow_third_check(int64_t *out) {
    const float e = 1.0f / 3.0f;
    int64_t n = 0, bad = 0, libm = 0;
#pragma omp parallel for reduction(+ : n, bad, libm) schedule(static)
    for (int rg = 0; rg < 65536; rg++)
        for (int b = 0; b < 256; b++) {
            const float _b = (float)b * 0.0039216f, _g = (float)(rg & 255) * 0.0039216f, _r = (float)(rg >> 8) * 0.0039216f;
            const float v[3] = {(_r * 0.412453f + _g * 0.357580f + _b * 0.180423f) / 0.950456f,
                                (_r * 0.212671f + _g * 0.715160f + _b * 0.072169f) / 1.0f,
                                (_r * 0.019334f + _g * 0.119193f + _b * 0.950227f) / 1.088754f};
            for (int k = 0; k < 3; k++) {
                if (!(v[k] > 0.008856f)) continue;
                const float want = (float)powl((long double)v[k], (long double)e);
                n++;
                bad += orc_pow_third(v[k]) != want;
                libm += powf(v[k], e) != want;
            }
        }
    out[0] = n; out[1] = bad; out[2] = libm;
}
/* rgb2CIELab gSLICr_seg_engine_shared.h:19-51 (input order b, g, r, a) */
void orc_rgb2lab(const uint8_t *bgra, float *out) {
    float _b = (float)bgra[0] * 0.0039216f, _g = (float)bgra[1] * 0.0039216f, _r = (float)bgra[2] * 0.0039216f;
    float x = _r * 0.412453f + _g * 0.357580f + _b * 0.180423f;
    float y = _r * 0.212671f + _g * 0.715160f + _b * 0.072169f;
    float z = _r * 0.019334f + _g * 0.119193f + _b * 0.950227f;
    const float epsilon = 0.008856f, kappa = 903.3f;
    float xr = x / 0.950456f, yr = y / 1.0f, zr = z / 1.088754f;
    float fx = xr > epsilon ? orc_pow_third(xr) : (kappa * xr + 16.0f) / 116.0f;
    float fy = yr > epsilon ? orc_pow_third(yr) : (kappa * yr + 16.0f) / 116.0f;
    float fz = zr > epsilon ? orc_pow_third(zr) : (kappa * zr + 16.0f) / 116.0f;
    out[0] = 116.0f * fy - 16.0f;
    out[1] = 500.0f * (fx - fy);
    out[2] = 200.0f * (fy - fz);
    out[3] = 0.0f;
}
static void rgb2xyz(const uint8_t *bgra, float *out) { /* shared.h:7-17 */
    float _b = (float)bgra[0] * 0.0039216f, _g = (float)bgra[1] * 0.0039216f, _r = (float)bgra[2] * 0.0039216f;
    out[0] = _r * 0.412453f + _g * 0.357580f + _b * 0.180423f;
    out[1] = _r * 0.212671f + _g * 0.715160f + _b * 0.072169f;
    out[2] = _r * 0.019334f + _g * 0.119193f + _b * 0.950227f;
    out[3] = 0.0f;
}
/* compute_slic_distance shared.h:92-104 (normalizer_color is unused there too) */
static float slic_distance(const float *pix, int x, int y, const spixel *c, float weight, float norm_xy) {
    float d0 = pix[0] - c->col[0], d1 = pix[1] - c->col[1], d2 = pix[2] - c->col[2];
    float dcolor = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    float ex = (float)x - c->cx, ey = (float)y - c->cy;
    float dxy = sqrtf(ex * ex + ey * ey);
    float t = dxy * norm_xy * weight;
    return sqrtf(dcolor * dcolor + t * t);
}
static void find_association(const float *lab, const spixel *sp, int32_t *idx, int w, int h, int mw, int mh, int S, float weight, float norm_xy) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int cx = x / S, cy = y / S, minidx = -1;
            float dist = 999999.9999f;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++) {
                    int xx = cx + j, yy = cy + i;
                    if (xx >= 0 && yy >= 0 && xx < mw && yy < mh) {
                        const spixel *c = &sp[yy * mw + xx];
                        float cd = slic_distance(lab + 4 * ((size_t)y * w + x), x, y, c, weight, norm_xy);
                        if (cd < dist) { dist = cd; minidx = c->id; }
                    }
                }
            if (minidx >= 0) idx[(size_t)y * w + x] = minidx;
        }
}
/* Update_Cluster_Center_device GPU.cu:260-357: one 16x16 block of the 3S x 3S window, tree-reduced */
static void block_partial(const float *lab, const int32_t *idx, int w, int h, int S, int sx, int sy, int id, int bz, int blocks_per_line,
                          float *col, float *xy, int *cnt) {
    float c[256][4], p[256][2];
    int n[256], any = 0;
    memset(c, 0, sizeof c); memset(p, 0, sizeof p); memset(n, 0, sizeof n);
    int bx = bz % blocks_per_line, by = bz / blocks_per_line;
    for (int ty = 0; ty < 16; ty++)
        for (int tx = 0; tx < 16; tx++) {
            int xo = bx * 16 + tx, yo = by * 16 + ty, l = ty * 16 + tx;
            if (xo < S * 3 && yo < S * 3) {
                int xi = sx * S - S + xo, yi = sy * S - S + yo;
                if (xi >= 0 && xi < w && yi >= 0 && yi < h && idx[(size_t)yi * w + xi] == id) {
                    memcpy(c[l], lab + 4 * ((size_t)yi * w + xi), 16);
                    p[l][0] = (float)xi; p[l][1] = (float)yi; n[l] = 1; any = 1;
                }
            }
        }
    if (any) {
        static const int steps[] = {128, 64, 32, 16, 8, 4, 2, 1};
        for (int s = 0; s < 8; s++) {
            int st = steps[s], lim = st >= 64 ? st : 32;   /* the last six steps run on threads 0..31 in lockstep */
            for (int l = 0; l < lim; l++) {
                /* lockstep: every lane reads the partner's value from before this step; partners l+st never
                 * overlap the updated range within one step except for st < 32, where lane l+st (< 32) is also
                 * updated in the same step — reads happen before writes, so use the old value */
            }
            float oc[256][4], op[256][2]; int on[256];
            memcpy(oc, c, sizeof c); memcpy(op, p, sizeof p); memcpy(on, n, sizeof n);
            for (int l = 0; l < lim; l++) {
                for (int k = 0; k < 4; k++) c[l][k] = oc[l][k] + oc[l + st][k];
                p[l][0] = op[l][0] + op[l + st][0]; p[l][1] = op[l][1] + op[l + st][1];
                n[l] = on[l] + on[l + st];
            }
        }
    }
    memcpy(col, c[0], 16); xy[0] = p[0][0]; xy[1] = p[0][1]; *cnt = n[0];
}
static void update_centers(const float *lab, const int32_t *idx, spixel *sp, int w, int h, int mw, int mh, int S) {
    int nblk = (int)ceilf((float)(S * S * 9) / 256.0f);   /* no_grid_per_center GPU.cu:77-79 */
    int bpl = S * 3 / 16;                                 /* no_blocks_per_line :160 (quirk 10) */
    if (bpl < 1) bpl = 1;
    for (int sy = 0; sy < mh; sy++)
        for (int sx = 0; sx < mw; sx++) {
            spixel *s = &sp[sy * mw + sx];
            float col[4] = {0, 0, 0, 0}, xy[2] = {0, 0};
            int cnt = 0;
            for (int bz = 0; bz < nblk; bz++) {            /* finalize_reduction_result_shared shared.h:151-173 */
                float c[4], p[2]; int n;
                block_partial(lab, idx, w, h, S, sx, sy, sy * mw + sx, bz, bpl, c, p, &n);
                for (int k = 0; k < 4; k++) col[k] += c[k];
                xy[0] += p[0]; xy[1] += p[1]; cnt += n;
            }
            s->cx = xy[0]; s->cy = xy[1]; memcpy(s->col, col, 16); s->n = cnt;
            if (cnt != 0) {
                s->cx /= (float)cnt; s->cy /= (float)cnt;
                for (int k = 0; k < 4; k++) s->col[k] /= (float)cnt;
            }
        }
}
/* supress_local_lable shared.h:175-204 */
static void enforce_connectivity(const int32_t *in, int32_t *out, int w, int h) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int cl = in[(size_t)y * w + x];
            if (x <= 1 || y <= 1 || x >= w - 2 || y >= h - 2) { out[(size_t)y * w + x] = cl; continue; }
            int dc = 0, dl = -1;
            for (int j = -2; j <= 2; j++)
                for (int i = -2; i <= 2; i++) {
                    int nl = in[(size_t)(y + j) * w + (x + i)];
                    if (nl != cl) { dl = nl; dc++; }
                }
            out[(size_t)y * w + x] = dc >= 16 ? dl : cl;
        }
}
static void init_centers(const float *lab, spixel *sp, int w, int h, int mw, int mh, int S) { /* init_cluster_centers_shared shared.h:73-90 */
    for (int y = 0; y < mh; y++)
        for (int x = 0; x < mw; x++) {
            int ix = x * S + S / 2, iy = y * S + S / 2;
            ix = ix >= w ? (x * S + w) / 2 : ix;
            iy = iy >= h ? (y * S + h) / 2 : iy;
            spixel *s = &sp[y * mw + x];
            s->id = y * mw + x; s->cx = (float)ix; s->cy = (float)iy; s->n = 0;
            memcpy(s->col, lab + 4 * ((size_t)iy * w + ix), 16);
        }
}
static void convert(const uint8_t *bgra, float *lab, size_t np, int color_space) { /* cvt_img_space_shared shared.h:53-71 */
    for (size_t p = 0; p < np; p++) {
        if (color_space == 0) orc_rgb2lab(bgra + 4 * p, lab + 4 * p);
        else if (color_space == 1) rgb2xyz(bgra + 4 * p, lab + 4 * p);
        else { lab[4 * p] = bgra[4 * p]; lab[4 * p + 1] = bgra[4 * p + 1]; lab[4 * p + 2] = bgra[4 * p + 2]; lab[4 * p + 3] = 0.f; }
    }
}
static int blocks_per_spixel(int S) { return (int)ceilf((float)(S * S * 9) / 256.0f); }
/* from the converted image on (seg_engine::Perform_Segmentation gSLICr_seg_engine.cpp:33-43) */
static void segment(const float *lab, int w, int h, int S, int iters, float weight, int connectivity, int32_t *labels, spixel *sp) {
    const size_t np = (size_t)w * h;
    const int mw = w / S, mh = h / S;                      /* (int)ceil(int / int) GPU.cu:70-71 */
    init_centers(lab, sp, w, h, mw, mh, S);
    memset(labels, 0, np * 4);
    const float norm_xy = 1.0f / (float)S;
    find_association(lab, sp, labels, w, h, mw, mh, S, weight, norm_xy);
    for (int it = 0; it < iters; it++) {
        update_centers(lab, labels, sp, w, h, mw, mh, S);
        find_association(lab, sp, labels, w, h, mw, mh, S, weight, norm_xy);
    }
    if (connectivity) {
        int32_t *tmp = (int32_t *)malloc(np * 4);
        enforce_connectivity(labels, tmp, w, h);
        enforce_connectivity(tmp, labels, w, h);
        free(tmp);
    }
}
/* seg_engine::Perform_Segmentation gSLICr_seg_engine.cpp:30-44.  bgra [h][w][4] u8 -> labels [h][w];
 * lab_out [h][w][4] (optional) receives the converted image; centers_out (optional) mw*mh*8 floats
 * (cx, cy, col[4], id, n) */
void orc_slic(const uint8_t *bgra, int w, int h, int S, int iters, float weight, int connectivity, int color_space, int32_t *labels,
              float *lab_out, float *centers_out) {
    const size_t np = (size_t)w * h;
    float *lab = (float *)malloc(np * 16);
    convert(bgra, lab, np, color_space);
    const int mw = w / S, mh = h / S;
    spixel *sp = (spixel *)calloc((size_t)mw * mh, sizeof(spixel));
    segment(lab, w, h, S, iters, weight, connectivity, labels, sp);
    if (lab_out) memcpy(lab_out, lab, np * 16);
    if (centers_out)
        for (int i = 0; i < mw * mh; i++) {
            float *o = centers_out + 8 * i;
            o[0] = sp[i].cx; o[1] = sp[i].cy; memcpy(o + 2, sp[i].col, 16); o[6] = (float)sp[i].id; o[7] = (float)sp[i].n;
        }
    free(lab); free(sp);
}

/* ---- the stages one at a time, on caller-supplied inputs: what tests/test_slic_reference_golden.py holds to the outputs of the
 * reference's own functions (tests/golden/slic_ref.npz).  Centres and partial sums travel as raw 32-byte records laid out like the
 * reference's spixel_info (gSLICr_spixel_info.h:11-17: center 2 f32, color_info 4 f32, id i32, no_pixels i32) = `spixel` above. */
void orc_slic_convert(const uint8_t *bgra, float *out, int64_t n, int color_space) { convert(bgra, out, (size_t)n, color_space); }
void orc_slic_init_centers(const float *lab, void *centres, int w, int h, int mw, int mh, int S) { init_centers(lab, (spixel *)centres, w, h, mw, mh, S); }
float orc_slic_distance(const float *pix, int x, int y, const void *centre, float weight, float norm_xy) {
    return slic_distance(pix, x, y, (const spixel *)centre, weight, norm_xy);
}
void orc_slic_find_association(const float *lab, const void *centres, int32_t *idx, int w, int h, int mw, int mh, int S, float weight) {
    find_association(lab, (const spixel *)centres, idx, w, h, mw, mh, S, weight, 1.0f / (float)S);
}
int orc_slic_blocks_per_spixel(int S) { return blocks_per_spixel(S); }
/* Update_Cluster_Center_device alone: the accum_map it leaves (GPU.cu:350-355), blocks_per_spixel(S) records per superpixel */
void orc_slic_partials(const float *lab, const int32_t *idx, void *accum, int w, int h, int mw, int mh, int S) {
    const int nblk = blocks_per_spixel(S);
    int bpl = S * 3 / 16;
    if (bpl < 1) bpl = 1;
    spixel *a = (spixel *)accum;
    for (int i = 0; i < mw * mh; i++)
        for (int bz = 0; bz < nblk; bz++) {
            spixel *o = &a[(size_t)i * nblk + bz];
            block_partial(lab, idx, w, h, S, i % mw, i / mw, i, bz, bpl, o->col, &o->cx, &o->n);
            o->id = 0;
        }
}
/* finalize_reduction_result_shared shared.h:151-173 alone */
void orc_slic_finalize(const void *accum, void *centres, int n_spixels, int nblk) {
    const spixel *a = (const spixel *)accum;
    spixel *sp = (spixel *)centres;
    for (int i = 0; i < n_spixels; i++) {
        spixel *s = &sp[i];
        s->cx = 0.f; s->cy = 0.f; s->n = 0;
        for (int k = 0; k < 4; k++) s->col[k] = 0.f;
        for (int b = 0; b < nblk; b++) {
            const spixel *q = &a[(size_t)i * nblk + b];
            s->cx += q->cx; s->cy += q->cy; s->n += q->n;
            for (int k = 0; k < 4; k++) s->col[k] += q->col[k];
        }
        if (s->n != 0) {
            s->cx /= (float)s->n; s->cy /= (float)s->n;
            for (int k = 0; k < 4; k++) s->col[k] /= (float)s->n;
        }
    }
}
void orc_slic_update_centers(const float *lab, const int32_t *idx, void *centres, int w, int h, int mw, int mh, int S) {
    update_centers(lab, idx, (spixel *)centres, w, h, mw, mh, S);
}
void orc_slic_connectivity(const int32_t *in, int32_t *out, int w, int h) { enforce_connectivity(in, out, w, h); }
/* the whole segmentation from a given converted image (e.g. the reference-compiled rgb2CIELab's) */
void orc_slic_from_lab(const float *lab, int w, int h, int S, int iters, float weight, int connectivity, int32_t *labels, void *centres) {
    spixel *sp = (spixel *)calloc((size_t)(w / S) * (h / S), sizeof(spixel));
    segment(lab, w, h, S, iters, weight, connectivity, labels, sp);
    if (centres) memcpy(centres, sp, (size_t)(w / S) * (h / S) * sizeof(spixel));
    free(sp);
}
