/*
 * tsar_oracle_fusion.c — CPU ORACLE of the depth-map fusion (row N3).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference ships the fuser only as a Windows binary (Fusion.exe; its command line is
 * x/1.sh:20-30: --num_consistent= --reproj_error= --depth_diff= --angle= --used_list=, inputs pair.txt,
 * cams, images, APD/<id>/TSAR_disp.dmb / TSAR_normals.dmb, output APD/APD_TSAR.ply).  The algorithm restated
 * here is the published geometric-consistency fusion of the ACMH/ACMM family that binary derives from:
 * for every pixel of every reference view, back-project, look the point up in each source view, re-project
 * the source's own depth back, and keep the pixel when at least num_consistent sources agree within
 * reproj_error pixels, depth_diff relative depth and `angle` degrees between normals; the fused point is the
 * mean of the agreeing points; source pixels that took part are marked used.
 * Deterministic choices: views are processed in order; marks made while a view is processed become visible
 * when the next view starts (the sequential original lets later pixels of the same view see them); the
 * normal test compares cosines (dot >= cos(angle)) instead of calling acos; points are emitted in raster
 * order per view.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float K[9], R[9], t[3]; } fus_cam;   /* world -> camera, like cams/%08d_cam.txt */

static inline float dot3(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
/* pixel (x, y) at `depth` of camera c -> world point */
static void unproject(const fus_cam *c, float x, float y, float depth, float *X) {
    const float fx = c->K[0], fy = c->K[4], cx = c->K[2], cy = c->K[5];
    float pc[3] = {depth * (x - cx) / fx, depth * (y - cy) / fy, depth};
    float d[3] = {pc[0] - c->t[0], pc[1] - c->t[1], pc[2] - c->t[2]};
    X[0] = fmaf(c->R[6], d[2], fmaf(c->R[3], d[1], c->R[0] * d[0]));      /* R^T d */
    X[1] = fmaf(c->R[7], d[2], fmaf(c->R[4], d[1], c->R[1] * d[0]));
    X[2] = fmaf(c->R[8], d[2], fmaf(c->R[5], d[1], c->R[2] * d[0]));
}
/* world point -> pixel + depth in camera c */
static void project(const fus_cam *c, const float *X, float *px, float *py, float *depth) {
    float pc[3] = {dot3(c->R, X) + c->t[0], dot3(c->R + 3, X) + c->t[1], dot3(c->R + 6, X) + c->t[2]};
    *depth = pc[2];
    *px = c->K[0] * pc[0] / pc[2] + c->K[2];
    *py = c->K[4] * pc[1] / pc[2] + c->K[5];
}
/* depth[v], normal[v] ([h][w][3], world), gray[v]; src lists in CSR form (off[n+1], src[]).  points_out: cap x 9
 * floats (xyz, normal, gray x3 -> stored once as 7th..9th = gray, n_consistent, view).  Returns the count. */
int orc_fuse(int n_views, int w, int h, const fus_cam *cams, const float *const *depth, const float *const *normal, const float *const *gray,
             const int32_t *off, const int32_t *src, int num_consistent, float reproj_error, float depth_diff, float cos_angle, int use_marks,
             float *points_out, int cap) {
    const size_t np = (size_t)w * h;
    uint8_t *mask = (uint8_t *)calloc((size_t)n_views * np, 1), *pending = (uint8_t *)calloc((size_t)n_views * np, 1);
    int n_out = 0;
    for (int i = 0; i < n_views; i++) {
        const fus_cam *ci = &cams[i];
        for (int r = 0; r < h; r++)
            for (int c = 0; c < w; c++) {
                const size_t p = (size_t)r * w + c;
                if (use_marks && mask[(size_t)i * np + p]) continue;
                const float ref_depth = depth[i][p];
                if (!(ref_depth > 0.0f)) continue;
                const float *rn = normal[i] + 3 * p;
                float X[3], acc[3], nacc[3], gacc;
                unproject(ci, (float)c, (float)r, ref_depth, X);
                memcpy(acc, X, 12); memcpy(nacc, rn, 12); gacc = gray[i][p];
                int ncons = 0, used[64][2];
                const int ns = off[i + 1] - off[i];
                for (int k = 0; k < ns && k < 64; k++) {
                    used[k][0] = -1;
                    const int j = src[off[i] + k];
                    const fus_cam *cj = &cams[j];
                    float sx, sy, sd;
                    project(cj, X, &sx, &sy, &sd);
                    if (!(sd > 0.0f)) continue;
                    const int sr = (int)floorf(sy + 0.5f), scn = (int)floorf(sx + 0.5f);
                    if (sr < 0 || sr >= h || scn < 0 || scn >= w) continue;
                    const size_t q = (size_t)sr * w + scn;
                    if (use_marks && mask[(size_t)j * np + q]) continue;
                    const float src_depth = depth[j][q];
                    if (!(src_depth > 0.0f)) continue;
                    float Y[3], bx, by, bd;
                    unproject(cj, (float)scn, (float)sr, src_depth, Y);
                    project(ci, Y, &bx, &by, &bd);
                    const float ex = (float)c - bx, ey = (float)r - by;
                    const float err = sqrtf(fmaf(ex, ex, ey * ey));
                    const float rel = fabsf(bd - ref_depth) / ref_depth;
                    const float cosang = dot3(rn, normal[j] + 3 * q);
                    if (err < reproj_error && rel < depth_diff && cosang >= cos_angle) {
                        acc[0] += Y[0]; acc[1] += Y[1]; acc[2] += Y[2];
                        nacc[0] += normal[j][3 * q]; nacc[1] += normal[j][3 * q + 1]; nacc[2] += normal[j][3 * q + 2];
                        gacc += gray[j][q];
                        used[k][0] = scn; used[k][1] = sr;
                        ncons++;
                    }
                }
                if (ncons >= num_consistent) {
                    const float inv = 1.0f / (float)(ncons + 1);
                    if (n_out < cap) {
                        float *o = points_out + 9 * (size_t)n_out;
                        o[0] = acc[0] * inv; o[1] = acc[1] * inv; o[2] = acc[2] * inv;
                        float nn[3] = {nacc[0] * inv, nacc[1] * inv, nacc[2] * inv};
                        const float nl = 1.0f / sqrtf(dot3(nn, nn));
                        o[3] = nn[0] * nl; o[4] = nn[1] * nl; o[5] = nn[2] * nl;
                        o[6] = gacc * inv; o[7] = (float)ncons; o[8] = (float)i;
                    }
                    n_out++;
                    if (use_marks)
                        for (int k = 0; k < ns && k < 64; k++)
                            if (used[k][0] != -1) pending[(size_t)src[off[i] + k] * np + (size_t)used[k][1] * w + used[k][0]] = 1;
                }
            }
        if (use_marks)
            for (size_t e = 0; e < (size_t)n_views * np; e++) mask[e] |= pending[e];
    }
    free(mask); free(pending);
    return n_out;
}
