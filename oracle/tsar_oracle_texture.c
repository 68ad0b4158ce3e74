/*
 * tsar_oracle_texture.c — CPU ORACLE of the weak-texture region detection (row N2): reference
 * main.cpp texture() :365-596, roberts() :214-240, Connect() :242-362.  TEST INFRASTRUCTURE ONLY,
 * PARITY UNPINNED (see tsar_oracle.c).
 *
 * Third-party arithmetic the reference calls and that is not in its tree (OpenCV 3.4.5):
 *   cv::pyrDown  -> restated from its published definition: 5x5 separable kernel [1 4 6 4 1]/16 per axis,
 *                   BORDER_REFLECT_101, output (w/2, h/2) sampled at even source pixels, 8-bit result
 *                   rounded as (sum + 128) >> 8;
 *   cv::HoughLinesP + cv::line (boundary closing of large regions, main.cpp:391-435) -> a randomised OpenCV-internal
 *                   algorithm that cannot be restated; replaced by a deterministic Hough transform with the same
 *                   parameters (orc_hough_close below; documented deviation, parity unpinned for this step).
 * Everything else is restated literally, including: the uchar wrap of (uchar)sqrt(t1+t2) for gradient
 * magnitudes >= 256 (x86 truncation), `const int sizerat = 2.5` (== 2), and Connect()'s parent
 * overwrite `connection[larger] = smaller`, which can lose an earlier link (orc_connect_literal).  The
 * GPU path labels true 4-connected components; orc_connect_true is that definition on the CPU.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
void orc_pyrdown(const uint8_t *src, int w, int h, uint8_t *dst) { /* dst is (w/2) x (h/2) */
    static const int k[5] = {1, 4, 6, 4, 1};
    const int dw = w / 2, dh = h / 2;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int sum = 0;
            for (int j = -2; j <= 2; j++) {
                const int sy = reflect101(2 * y + j, h);
                int row = 0;
                for (int i = -2; i <= 2; i++) row += k[i + 2] * src[(size_t)sy * w + reflect101(2 * x + i, w)];
                sum += k[j + 2] * row;
            }
            dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
        }
}
/* roberts() + cv::threshold(Robthr = 4, THRESH_BINARY): out = 255 (edge) or 0 */
void orc_roberts_threshold(const uint8_t *src, int w, int h, uint8_t *dst) {
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            int t1, t2;
            if (i > 0 && i < h - 1 && j > 0 && j < w - 1) {
                const int a = src[(size_t)i * w + j] - src[(size_t)(i + 1) * w + j + 1];
                const int b = src[(size_t)(i + 1) * w + j] - src[(size_t)i * w + j + 1];
                t1 = a * a; t2 = b * b;
            } else {
                t1 = 100 * 50; t2 = t1;
            }
            const uint8_t mag = (uint8_t)(int)sqrt((double)(t1 + t2));   /* wraps for values >= 256 */
            dst[(size_t)i * w + j] = mag > 4 ? 255 : 0;
        }
}
/* main.cpp:441-452: a border pixel becomes flat when its inner neighbour is flat */
void orc_border_fix(uint8_t *img, int w, int h) {
    for (int y = 0; y < h; y++) {
        if (img[(size_t)y * w + 1] == 0) img[(size_t)y * w] = 0;
        if (img[(size_t)y * w + w - 2] == 0) img[(size_t)y * w + w - 1] = 0;
    }
    for (int x = 0; x < w; x++) {
        if (img[(size_t)w + x] == 0) img[x] = 0;
        if (img[(size_t)(h - 2) * w + x] == 0) img[(size_t)(h - 1) * w + x] = 0;
    }
}
/* Connect() main.cpp:242-362, literally.  Returns the number of labels (label 0 = edge pixels). */
int orc_connect_literal(const uint8_t *img, int w, int h, int32_t *lab, int32_t *count_out, int cap) {
    const size_t np = (size_t)w * h;
    int *conn = (int *)malloc((np + 2) * sizeof(int));
    int cnt = 1;
    conn[0] = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t p = (size_t)y * w + x;
            if (img[p] == 255) { lab[p] = 0; continue; }
            const int left = x > 0 && img[p] == 0 && img[p - 1] == 0, up = y > 0 && img[p] == 0 && img[p - w] == 0;
            if (left) lab[p] = lab[p - 1];
            if (up) lab[p] = lab[p - w];
            if (!left && !up) { lab[p] = cnt; conn[cnt] = cnt; cnt++; }
            else if (left && up) {
                const int ll = lab[p - 1], ul = lab[p - w];
                if (ll > ul) { conn[ll] = ul; lab[p] = ul; }
                else if (ll < ul) { conn[ul] = ll; lab[p] = ll; }
            }
        }
    for (int i = 1; i < cnt; i++) {
        int cur = conn[i], pre = conn[cur];
        while (pre != cur) { cur = pre; pre = conn[pre]; }
        conn[i] = cur;
    }
    int labelnum = 1;
    int *mapping = (int *)calloc((size_t)cnt + 1, sizeof(int));
    for (int i = 1; i < cnt; i++)
        if (conn[i] == i) mapping[i] = labelnum++;
    for (int i = 1; i < cnt; i++) conn[i] = mapping[conn[i]];
    if (count_out) memset(count_out, 0, (size_t)(labelnum < cap ? labelnum : cap) * sizeof(int32_t));
    for (size_t p = 0; p < np; p++) {
        lab[p] = conn[lab[p]];
        if (count_out && lab[p] < cap) count_out[lab[p]]++;
    }
    free(conn); free(mapping);
    return labelnum;
}
/* true 4-connected components of the zero pixels, numbered 1.. in raster order of each component's first
 * pixel (what Connect() produces whenever its parent overwrite loses nothing) */
int orc_connect_true(const uint8_t *img, int w, int h, int32_t *lab, int32_t *count_out, int cap) {
    const size_t np = (size_t)w * h;
    int32_t *stack = (int32_t *)malloc(np * sizeof(int32_t));
    int labelnum = 1;
    for (size_t p = 0; p < np; p++) lab[p] = img[p] == 255 ? 0 : -1;
    for (size_t s = 0; s < np; s++) {
        if (lab[s] != -1) continue;
        const int l = labelnum++;
        size_t top = 0;
        stack[top++] = (int32_t)s; lab[s] = l;
        while (top) {
            const int32_t p = stack[--top];
            const int x = p % w, y = p / w;
            if (x > 0 && lab[p - 1] == -1) { lab[p - 1] = l; stack[top++] = p - 1; }
            if (x < w - 1 && lab[p + 1] == -1) { lab[p + 1] = l; stack[top++] = p + 1; }
            if (y > 0 && lab[p - w] == -1) { lab[p - w] = l; stack[top++] = p - w; }
            if (y < h - 1 && lab[p + w] == -1) { lab[p + w] = l; stack[top++] = p + w; }
        }
    }
    if (count_out) {
        memset(count_out, 0, (size_t)(labelnum < cap ? labelnum : cap) * sizeof(int32_t));
        for (size_t p = 0; p < np; p++)
            if (lab[p] < cap) count_out[lab[p]]++;
    }
    free(stack);
    return labelnum;
}
/* main.cpp:481-593: per-label statistics -> text (-1 = "true weak"), size (max bbox side), centroid*4.
 * lab is the quarter-resolution label image (w4 x h4). */
void orc_region_stats(const int32_t *lab, int w4, int h4, int labelnum, float *text, float *size, int32_t *cenx, int32_t *ceny, int32_t *count) {
    int32_t *xmin = (int32_t *)malloc((size_t)labelnum * 4), *xmax = (int32_t *)malloc((size_t)labelnum * 4);
    int32_t *ymin = (int32_t *)malloc((size_t)labelnum * 4), *ymax = (int32_t *)malloc((size_t)labelnum * 4);
    int32_t *sx = (int32_t *)calloc((size_t)labelnum, 4), *sy = (int32_t *)calloc((size_t)labelnum, 4);
    memset(count, 0, (size_t)labelnum * 4);
    for (int i = 0; i < labelnum; i++) { xmax[i] = 0; xmin[i] = w4 - 1; ymax[i] = 0; ymin[i] = h4 - 1; }
    for (int y = 0; y < h4; y++)
        for (int x = 0; x < w4; x++) {
            const int l = lab[(size_t)y * w4 + x];
            count[l]++; sx[l] += x; sy[l] += y;      /* int accumulators as in the reference (labelx/labely) */
            if (x > xmax[l]) xmax[l] = x;
            if (x < xmin[l]) xmin[l] = x;
            if (y > ymax[l]) ymax[l] = y;
            if (y < ymin[l]) ymin[l] = y;
        }
    text[0] = 1.0f; size[0] = 0.f; cenx[0] = 0; ceny[0] = 0;
    for (int t = 1; t < labelnum; t++) {
        cenx[t] = (int32_t)(sx[t] * 4 / count[t]);    /* labelx *= 4; labelx /= labelcnt (:577-580) */
        ceny[t] = (int32_t)(sy[t] * 4 / count[t]);
        text[t] = 1.0f; size[t] = 0.f;
        if (count[t] > 5000) {                        /* weaktextnum (:62, :357-361) */
            const int xs = xmax[t] - xmin[t], ys = ymax[t] - ymin[t];
            if (xs * ys < 2 * count[t] || count[t] > 100000) {   /* `const int sizerat = 2.5` (:63, :526) */
                text[t] = -1.0f;
                size[t] = (float)(xs > ys ? xs : ys);
            }
        }
    }
    free(xmin); free(xmax); free(ymin); free(ymax); free(sx); free(sy);
}
/* main.cpp:559-568: full-resolution label per pixel */
void orc_upsample_labels(const int32_t *lab4, int w4, int h4, int w, int h, int32_t *out) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int sxp = x / 4, syp = y / 4;
            if (sxp >= w4) sxp--;
            if (syp >= h4) syp--;
            out[(size_t)y * w + x] = lab4[(size_t)syp * w4 + sxp];
        }
}

/* ---- boundary closing of large regions (main.cpp:385-435) -------------------------------------------------
 * The reference finds long straight pieces of each large region's boundary with cv::HoughLinesP(rho 1, theta 1 deg,
 * threshold Houthr = 110, minLineLength = 160, maxLineGap = 18; :60-62, :425) and draws them into the edge image
 * with cv::line, so that the second labelling cannot leak through gaps in a straight edge.  HoughLinesP is a
 * randomised OpenCV-internal algorithm (not in the reference tree): PARITY UNPINNED for this step.
 * What is built instead is a deterministic Hough transform with the same parameters:
 *   boundary(L)  = pixels not labelled L that have a 4-neighbour labelled L (main.cpp:393-421, literal);
 *   votes        acc[theta][rho], theta = 0..179 deg, rho = rint(x cos + y sin) (fp32 tables, fp32 arithmetic);
 *   lines        cells with acc >= 110 that are local maxima (> left / upper neighbour, >= right / lower one);
 *   segments     walk the line across the image along its major axis, one pixel per step; a run of boundary pixels
 *                ends when more than maxLineGap consecutive steps miss; it is kept when its end points differ by
 *                >= minLineLength in x or in y (OpenCV's good_line test); kept runs are drawn with an 8-connected
 *                Bresenham line.
 * lab0: labels of orc_connect_true on the edge image BEFORE the border fix; edge is modified in place. */
#define ORC_HOUGH_THR 110
#define ORC_HOUGH_MINLEN 160
#define ORC_HOUGH_MAXGAP 18
#define ORC_WEAK_COUNT 5000

void orc_hough_tables(float *cs, float *sn) {          /* 180 entries each */
    for (int t = 0; t < 180; t++) {
        const double a = (double)t * 3.14159265358979323846 / 180.0;
        cs[t] = (float)cos(a);
        sn[t] = (float)sin(a);
    }
}
static void draw_line8(uint8_t *img, int w, int h, int x0, int y0, int x1, int y1) {
    int dx = abs(x1 - x0), sx = x0 < x1 ? 1 : -1;
    int dy = -abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
    int err = dx + dy;
    for (;;) {
        if (x0 >= 0 && x0 < w && y0 >= 0 && y0 < h) img[(size_t)y0 * w + x0] = 255;
        if (x0 == x1 && y0 == y1) break;
        const int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x0 += sx; }
        if (e2 <= dx) { err += dx; y0 += sy; }
    }
}
/* one Hough cell -> segments; returns the number drawn */
static int walk_line(const uint8_t *bmask, uint8_t *edge, int w, int h, float c, float s, float rho) {
    const int xmajor = fabsf(s) >= fabsf(c);            /* the line is closer to horizontal: step in x */
    const int n = xmajor ? w : h;
    int run = 0, sx = 0, sy = 0, lx = 0, ly = 0, gap = 0, drawn = 0;
    for (int k = 0; k <= n; k++) {
        int on = 0, x = 0, y = 0;
        if (k < n) {
            if (xmajor) { x = k; y = (int)lrintf((rho - (float)x * c) / s); }
            else        { y = k; x = (int)lrintf((rho - (float)y * s) / c); }
            on = x >= 0 && x < w && y >= 0 && y < h && bmask[(size_t)y * w + x] != 0;
        }
        if (on) {
            if (!run) { run = 1; sx = x; sy = y; }
            lx = x; ly = y; gap = 0;
        } else if (run && (++gap > ORC_HOUGH_MAXGAP || k == n)) {
            if (abs(lx - sx) >= ORC_HOUGH_MINLEN || abs(ly - sy) >= ORC_HOUGH_MINLEN) { draw_line8(edge, w, h, sx, sy, lx, ly); drawn++; }
            run = 0; gap = 0;
        }
    }
    return drawn;
}
/* returns the number of segments drawn into edge */
int orc_hough_close(uint8_t *edge, const int32_t *lab0, int labelnum0, int w, int h) {
    const size_t np = (size_t)w * h;
    int32_t *count = (int32_t *)calloc((size_t)labelnum0, sizeof(int32_t));
    for (size_t p = 0; p < np; p++) count[lab0[p]]++;
    float cs[180], sn[180];
    orc_hough_tables(cs, sn);
    const int rmax = w + h + 2, nrho = 2 * rmax + 1;
    int32_t *acc = (int32_t *)malloc((size_t)180 * nrho * sizeof(int32_t));
    uint8_t *bmask = (uint8_t *)malloc(np);
    int drawn = 0;
    for (int L = 1; L < labelnum0; L++) {
        if (count[L] <= ORC_WEAK_COUNT) continue;        /* weaklabel0: labelcnt > weaktextnum (main.cpp:356-361) */
        memset(bmask, 0, np);
        memset(acc, 0, (size_t)180 * nrho * sizeof(int32_t));
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const size_t p = (size_t)y * w + x;
                if (lab0[p] == L) continue;
                if ((x > 0 && lab0[p - 1] == L) || (x < w - 1 && lab0[p + 1] == L) || (y > 0 && lab0[p - w] == L) || (y < h - 1 && lab0[p + w] == L)) {
                    bmask[p] = 255;
                    for (int t = 0; t < 180; t++) {
                        const int r = (int)lrintf((float)x * cs[t] + (float)y * sn[t]);
                        acc[(size_t)t * nrho + r + rmax]++;
                    }
                }
            }
        for (int t = 0; t < 180; t++)
            for (int r = 0; r < nrho; r++) {
                const int32_t v = acc[(size_t)t * nrho + r];
                if (v < ORC_HOUGH_THR) continue;
                const int32_t left = r > 0 ? acc[(size_t)t * nrho + r - 1] : 0, right = r < nrho - 1 ? acc[(size_t)t * nrho + r + 1] : 0;
                const int32_t up = t > 0 ? acc[(size_t)(t - 1) * nrho + r] : 0, down = t < 179 ? acc[(size_t)(t + 1) * nrho + r] : 0;
                if (!(v > left && v >= right && v > up && v >= down)) continue;
                drawn += walk_line(bmask, edge, w, h, cs[t], sn[t], (float)(r - rmax));
            }
    }
    free(count); free(acc); free(bmask);
    return drawn;
}
