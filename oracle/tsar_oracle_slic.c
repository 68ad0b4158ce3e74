/*
 * tsar_oracle_slic.c — CPU ORACLE of the gSLICr superpixel segmentation as the reference drives it
 * (reference gSLICr_Lib/engines/gSLICr_seg_engine.cpp:30-44, gSLICr_seg_engine_GPU.cu,
 * gSLICr_seg_engine_shared.h; settings main.cpp:608-615).  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED
 * (see tsar_oracle.c).
 *
 * Deterministic choices: pow(x, 1/3) of rgb2CIELab (shared.h:41-46) becomes a Newton cube root built
 * from IEEE operations only (orc_cbrtf), so that both sides of the parity check compute identical
 * CIELAB values; the block reduction of Update_Cluster_Center_device is restated in its exact tree
 * order so float sums agree bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float cx, cy; float col[4]; int id, n; } spixel;

float orc_cbrtf(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    v.u = v.u / 3u + 0x2a5137a0u;
    float y = v.f;
    for (int i = 0; i < 4; i++) y = (y + y + x / (y * y)) * 0.333333343f;
    return y;
}
/* rgb2CIELab gSLICr_seg_engine_shared.h:19-51 (input order b, g, r, a) */
void orc_rgb2lab(const uint8_t *bgra, float *out) {
    float _b = (float)bgra[0] * 0.0039216f, _g = (float)bgra[1] * 0.0039216f, _r = (float)bgra[2] * 0.0039216f;
    float x = _r * 0.412453f + _g * 0.357580f + _b * 0.180423f;
    float y = _r * 0.212671f + _g * 0.715160f + _b * 0.072169f;
    float z = _r * 0.019334f + _g * 0.119193f + _b * 0.950227f;
    const float epsilon = 0.008856f, kappa = 903.3f;
    float xr = x / 0.950456f, yr = y / 1.0f, zr = z / 1.088754f;
    float fx = xr > epsilon ? orc_cbrtf(xr) : (kappa * xr + 16.0f) / 116.0f;
    float fy = yr > epsilon ? orc_cbrtf(yr) : (kappa * yr + 16.0f) / 116.0f;
    float fz = zr > epsilon ? orc_cbrtf(zr) : (kappa * zr + 16.0f) / 116.0f;
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
/* seg_engine::Perform_Segmentation gSLICr_seg_engine.cpp:30-44.  bgra [h][w][4] u8 -> labels [h][w];
 * lab_out [h][w][4] (optional) receives the converted image; centers_out (optional) mw*mh*8 floats
 * (cx, cy, col[4], id, n) */
void orc_slic(const uint8_t *bgra, int w, int h, int S, int iters, float weight, int connectivity, int color_space, int32_t *labels,
              float *lab_out, float *centers_out) {
    const size_t np = (size_t)w * h;
    float *lab = (float *)malloc(np * 16);
    for (size_t p = 0; p < np; p++) {
        if (color_space == 0) orc_rgb2lab(bgra + 4 * p, lab + 4 * p);
        else if (color_space == 1) rgb2xyz(bgra + 4 * p, lab + 4 * p);
        else { lab[4 * p] = bgra[4 * p]; lab[4 * p + 1] = bgra[4 * p + 1]; lab[4 * p + 2] = bgra[4 * p + 2]; lab[4 * p + 3] = 0.f; }
    }
    const int mw = w / S, mh = h / S;                      /* (int)ceil(int / int) GPU.cu:70-71 */
    spixel *sp = (spixel *)calloc((size_t)mw * mh, sizeof(spixel));
    for (int y = 0; y < mh; y++)                           /* init_cluster_centers_shared shared.h:73-90 */
        for (int x = 0; x < mw; x++) {
            int ix = x * S + S / 2, iy = y * S + S / 2;
            ix = ix >= w ? (x * S + w) / 2 : ix;
            iy = iy >= h ? (y * S + h) / 2 : iy;
            spixel *s = &sp[y * mw + x];
            s->id = y * mw + x; s->cx = (float)ix; s->cy = (float)iy; s->n = 0;
            memcpy(s->col, lab + 4 * ((size_t)iy * w + ix), 16);
        }
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
    if (lab_out) memcpy(lab_out, lab, np * 16);
    if (centers_out)
        for (int i = 0; i < mw * mh; i++) {
            float *o = centers_out + 8 * i;
            o[0] = sp[i].cx; o[1] = sp[i].cy; memcpy(o + 2, sp[i].col, 16); o[6] = (float)sp[i].id; o[7] = (float)sp[i].n;
        }
    free(lab); free(sp);
}
