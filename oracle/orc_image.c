/*
 * orc_image.c -- oracle (TEST INFRASTRUCTURE ONLY, parity unpinned; see
 * slam_oracle.h): filters, extractor and LK pyramid.
 *
 * Follows /root/reference/src/extractor.jl and src/optical_flow/pyramid.jl,
 * src/optical_flow/lucas_kanade.jl:102-138, plus the documented semantics of the
 * un-vendored packages they call.
 */
#include "slam_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#define IDX(y, x, H) ((size_t)(y) + (size_t)(x) * (size_t)(H))

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------ */
/* Kernel.gaussian(sigma) -> KernelFactors.gaussian(sigma, 4*ceil(sigma)+1):
 * g = [exp(-x^2/(2*sigma^2)) for x=-w:w]; g/sum(g).  (extractor.jl:69)      */
int orc_gaussian_taps(double sigma, double *w)
{
    int l = 4 * (int)ceil(sigma) + 1;
    int hw = l >> 1;
    double s = 0.0;
    for (int i = 0; i < l; i++) {
        double x = (double)(i - hw);
        w[i] = exp(-(x * x) / (2.0 * (sigma * sigma)));
        s += w[i];
    }
    for (int i = 0; i < l; i++) w[i] = w[i] / s;
    return l;
}

/* ImageFiltering.imfilter with a 2-factor separable kernel = correlation:
 * tmp = 0; for j in kernel: tmp += img[i+j]*k[j]; first factor (dim 1) first. */
void orc_imfilter_sep(double *out, const double *in, int H, int W,
                      const double *k1, int n1, const double *k2, int n2, int border)
{
    double *tmp = (double *)malloc(sizeof(double) * (size_t)H * W);
    int h1 = n1 >> 1, h2 = n2 >> 1;
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            double acc = 0.0;
            for (int j = 0; j < n1; j++) {
                int yy = y + j - h1;
                double v;
                if (border == 0) v = in[IDX(clampi(yy, 0, H - 1), x, H)];
                else v = (yy < 0 || yy >= H) ? 0.0 : in[IDX(yy, x, H)];
                acc += v * k1[j];
            }
            tmp[IDX(y, x, H)] = acc;
        }
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            double acc = 0.0;
            for (int j = 0; j < n2; j++) {
                int xx = x + j - h2;
                double v;
                if (border == 0) v = tmp[IDX(y, clampi(xx, 0, W - 1), H)];
                else v = (xx < 0 || xx >= W) ? 0.0 : tmp[IDX(y, xx, H)];
                acc += v * k2[j];
            }
            out[IDX(y, x, H)] = acc;
        }
    free(tmp);
}

/* ------------------------------------------------------------------------ */
/* KernelFactors.IIRGaussian(sigma) -> TriggsSdika(a, B)  (SURVEY A.4):
 * Young & van Vliet recursive Gaussian with Triggs & Sdika boundary matrix. */
typedef struct {
    double a[3], scale, M[9], asum, bsum;
} iir_t;

static void iir_init(iir_t *k, double sigma)
{
    const double m0 = 1.16680, m1 = 1.10783, m2 = 1.40586;
    double q = 1.31564 * (sqrt(1 + 0.490811 * sigma * sigma) - 1);
    double ascale = (m0 + q) * (m1 * m1 + m2 * m2 + 2 * m1 * q + q * q);
    double B = (m0 * (m1 * m1 + m2 * m2) / ascale);
    B = B * B;
    double a1 = q * (2 * m0 * m1 + m1 * m1 + m2 * m2 + (2 * m0 + 4 * m1) * q + 3 * q * q) / ascale;
    double a2 = -q * q * (m0 + 2 * m1 + 3 * q) / ascale;
    double a3 = q * q * q / ascale;
    k->a[0] = a1; k->a[1] = a2; k->a[2] = a3;
    k->scale = B;
    double Md = (1 + a1 - a2 + a3) * (1 - a1 - a2 - a3) * (1 + a2 + (a1 - a3) * a3);
    double M[9] = {
        -a3 * a1 + 1 - a3 * a3 - a2, (a3 + a1) * (a2 + a3 * a1), a3 * (a1 + a3 * a2),
        a1 + a3 * a2, -(a2 - 1) * (a2 + a3 * a1), -(a3 * a1 + a3 * a3 + a2 - 1) * a3,
        a3 * a1 + a2 + a1 * a1 - a2 * a2,
        a1 * a2 + a3 * a2 * a2 - a1 * a3 * a3 - a3 * a3 * a3 - a3 * a2 + a3,
        a3 * (a1 + a3 * a2)};
    for (int i = 0; i < 9; i++) k->M[i] = M[i] / Md;
    k->asum = (a1 + a2) + a3;
    k->bsum = k->asum;
}

void orc_iir_coeffs(double sigma, double *a, double *scale, double *M, double *asum)
{
    iir_t k; iir_init(&k, sigma);
    memcpy(a, k.a, sizeof k.a); *scale = k.scale; memcpy(M, k.M, sizeof k.M); *asum = k.asum;
}

/* One line of ImageFiltering._imfilter_dim!(out, img, ::TriggsSdika, ...):
 * left border init, forward recursion, Triggs-Sdika right border, backward
 * recursion, final scaling.  n >= 4.  v = line base, s = stride (in place).
 * iminus/iplus: border values (replicate: first/last sample; Fill: 0). */
static void iir_line(double *v, int n, ptrdiff_t s, const iir_t *k, int fill0)
{
    const double a1 = k->a[0], a2 = k->a[1], a3 = k->a[2];
    double iminus = fill0 ? 0.0 : v[0];
    double iplus = fill0 ? 0.0 : v[(ptrdiff_t)(n - 1) * s];
    double uminus = iminus / (1 - k->asum);
    /* _leftborder!: tmp = img[i]; += a[j]*out[i-j] (j<n); += a[j]*uminus (j>=n) */
    double o0 = ((v[0] + a1 * uminus) + a2 * uminus) + a3 * uminus;
    double o1 = ((v[s] + a1 * o0) + a2 * uminus) + a3 * uminus;
    double o2 = ((v[2 * s] + a1 * o1) + a2 * o0) + a3 * uminus;
    v[0] = o0; v[s] = o1; v[2 * s] = o2;
    /* forward, i = 4 .. n-1 (1-based), plus the last point done in _rightborder! */
    for (int i = 3; i < n; i++) {
        double t = ((v[i * s] + a1 * v[(i - 1) * s]) + a2 * v[(i - 2) * s]) + a3 * v[(i - 3) * s];
        v[i * s] = t;
    }
    /* _rightborder! */
    double uplus = iplus / (1 - k->asum);
    double vplus = uplus / (1 - k->bsum);
    double d0 = v[(n - 1) * s] - uplus, d1 = v[(n - 2) * s] - uplus, d2 = v[(n - 3) * s] - uplus;
    const double *M = k->M;
    double vr0 = ((M[0] * d0 + M[1] * d1) + M[2] * d2) + vplus;
    double vr1 = ((M[3] * d0 + M[4] * d1) + M[5] * d2) + vplus;
    double vr2 = ((M[6] * d0 + M[7] * d1) + M[8] * d2) + vplus;
    v[(n - 1) * s] = vr0;
    v[(n - 2) * s] = ((v[(n - 2) * s] + a1 * v[(n - 1) * s]) + a2 * vr1) + a3 * vr2;
    v[(n - 3) * s] = ((v[(n - 3) * s] + a1 * v[(n - 2) * s]) + a2 * v[(n - 1) * s]) + a3 * vr1;
    for (int i = n - 4; i >= 0; i--) {
        double t = ((v[i * s] + a1 * v[(i + 1) * s]) + a2 * v[(i + 2) * s]) + a3 * v[(i + 3) * s];
        v[i * s] = t;
    }
    for (int i = 0; i < n; i++) v[i * s] *= k->scale;
}

static void iir_2d(double *out, int H, int W, const iir_t *k, int fill0)
{
    for (int x = 0; x < W; x++) iir_line(out + IDX(0, x, H), H, 1, k, fill0); /* dim 1 */
    for (int y = 0; y < H; y++) iir_line(out + y, W, H, k, fill0);             /* dim 2 */
}

void orc_iir_gaussian(double *out, const double *in, int H, int W, double sigma, int border)
{
    iir_t k; iir_init(&k, sigma);
    size_t N = (size_t)H * W;
    if (out != in) memcpy(out, in, N * sizeof(double));
    if (border == 2) {
        /* imfilter(img, kern, NA()): filter with Fill(0), divide by the
         * Fill(0)-filtered indicator of valid pixels (all ones here). */
        double *ones = (double *)malloc(N * sizeof(double));
        for (size_t i = 0; i < N; i++) ones[i] = 1.0;
        iir_2d(out, H, W, &k, 1);
        iir_2d(ones, H, W, &k, 1);
        for (size_t i = 0; i < N; i++) out[i] /= ones[i];
        free(ones);
    } else {
        iir_2d(out, H, W, &k, border == 1);
    }
}

/* Interpolations.BSpline(Linear()) at a 1-based in-bounds position (A.7). */
double orc_bilinear(const double *img, int H, int W, double r, double c)
{
    int iy = (int)floor(r), ix = (int)floor(c);
    if (iy > H - 1) iy = H - 1;
    if (ix > W - 1) ix = W - 1;
    if (iy < 1) iy = 1;
    if (ix < 1) ix = 1;
    double fy = r - iy, fx = c - ix;
    const double *p = img + IDX(iy - 1, ix - 1, H);
    int dy = (H > 1) ? 1 : 0;
    size_t dx = (W > 1) ? (size_t)H : 0;
    /* weighted-index recursion peels the first index first: dim 1 outermost */
    double r0 = (1 - fx) * p[0] + fx * p[dx];            /* row iy   */
    double r1 = (1 - fx) * p[dy] + fx * p[dy + dx];      /* row iy+1 */
    return (1 - fy) * r0 + fy * r1;
}

/* ImageTransformations.imresize!(resized, itp): src = sf*i + (0.5 - 0.5*sf)
 * with sf = n_src/n_dst (>= 1 here, so no clamp branch) -- SURVEY A.6. */
void orc_imresize(double *dst, int Hd, int Wd, const double *src, int Hs, int Ws)
{
    double sy = (double)Hs / (double)Hd, sx = (double)Ws / (double)Wd;
    double oy = 1 - 0.5 - sy * (1 - 0.5), ox = 1 - 0.5 - sx * (1 - 0.5);
    for (int x = 1; x <= Wd; x++)
        for (int y = 1; y <= Hd; y++) {
            double r = sy * y + oy, c = sx * x + ox;
            if (sy < 1) r = r < 1 ? 1 : (r > Hs ? Hs : r);
            if (sx < 1) c = c < 1 ? 1 : (c > Ws ? Ws : c);
            dst[IDX(y - 1, x - 1, Hd)] = orc_bilinear(src, Hs, Ws, r, c);
        }
}

/* ------------------------------------------------------------------------ */
/* get_mask, extractor.jl:116-122; to_cartesian = round-half-even (SLAM.jl:30,41);
 * ImageDraw Ellipse fill: ((i-cy)/r)^2 + ((j-cx)/r)^2 < 1, in-bounds only.  */
void orc_get_mask(double *mask, int H, int W, const double *pts_yx, int n, int radius)
{
    for (size_t i = 0; i < (size_t)H * W; i++) mask[i] = 1.0;
    for (int k = 0; k < n; k++) {
        long cy = (long)rint(pts_yx[2 * k]), cx = (long)rint(pts_yx[2 * k + 1]);
        for (long i = cy - radius; i <= cy + radius; i++)
            for (long j = cx - radius; j <= cx + radius; j++) {
                double a = (double)(i - cy) / (double)radius, b = (double)(j - cx) / (double)radius;
                double val = a * a + b * b;
                if (val < 1 && i >= 1 && i <= H && j >= 1 && j <= W) mask[IDX(i - 1, j - 1, H)] = 0.0;
            }
    }
}

/* Images.shi_tomasi: Sobel (KernelFactors.sobel: (-1,0,1)/2 x
 * (1,2,1)/4), replicate border; products; 3x3 box mean (separable 1/3 x 1/3,
 * replicate); response = ((cxx+cyy) - sqrt((cxx-cyy)^2 + 4*cxy^2))/2.        */
void orc_shi_tomasi(double *resp, const double *cell, int h, int w, int ld)
{
    size_t N = (size_t)h * w;
    double *v = (double *)malloc(N * sizeof(double));
    double *g1 = (double *)malloc(N * sizeof(double));
    double *g2 = (double *)malloc(N * sizeof(double));
    double *c11 = (double *)malloc(N * sizeof(double));
    double *c12 = (double *)malloc(N * sizeof(double));
    double *c22 = (double *)malloc(N * sizeof(double));
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) v[IDX(y, x, h)] = cell[(size_t)y + (size_t)x * ld];
    const double d[3] = {-1.0 / 2, 0.0 / 2, 1.0 / 2}, s[3] = {1.0 / 4, 2.0 / 4, 1.0 / 4};
    const double b[3] = {1.0 / 3, 1.0 / 3, 1.0 / 3};
    orc_imfilter_sep(g1, v, h, w, d, 3, s, 3, 0); /* derivative along dim 1 */
    orc_imfilter_sep(g2, v, h, w, s, 3, d, 3, 0); /* derivative along dim 2 */
    for (size_t i = 0; i < N; i++) {
        c11[i] = g1[i] * g1[i]; c12[i] = g1[i] * g2[i]; c22[i] = g2[i] * g2[i];
    }
    orc_imfilter_sep(g1, c11, h, w, b, 3, b, 3, 0);
    orc_imfilter_sep(g2, c12, h, w, b, 3, b, 3, 0);
    orc_imfilter_sep(v, c22, h, w, b, 3, b, 3, 0);
    for (size_t i = 0; i < N; i++) {
        double xx = g1[i], xy = g2[i], yy = v[i];
        double dd = xx - yy;
        resp[i] = ((xx + yy) - sqrt(dd * dd + 4 * (xy * xy))) / 2;
    }
    free(v); free(g1); free(g2); free(c11); free(c12); free(c22);
}

/* _shi_tomasi, extractor.jl:24-42 (findlocalmaxima: strict 8-neighbour, edges
 * included, column-major order; stable descending sortperm; top n; drop
 * < min_response).                                                          */
int orc_shi_tomasi_cell(uint8_t *corners, double *resp, const double *cell,
                        int h, int w, int ld, int n_keypoints, double min_response)
{
    size_t N = (size_t)h * w;
    memset(corners, 0, N);
    orc_shi_tomasi(resp, cell, h, w, ld);
    int *maxima = (int *)malloc(N * sizeof(int));
    int nm = 0;
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) {
            double c = resp[IDX(y, x, h)];
            int ismax = 1;
            for (int dx = -1; dx <= 1 && ismax; dx++)
                for (int dy = -1; dy <= 1; dy++) {
                    if (!dx && !dy) continue;
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    if (!(resp[IDX(yy, xx, h)] < c)) { ismax = 0; break; }
                }
            if (ismax) maxima[nm++] = (int)IDX(y, x, h);
        }
    /* stable insertion sort, descending by response */
    for (int i = 1; i < nm; i++) {
        int m = maxima[i]; double r = resp[m];
        int j = i - 1;
        while (j >= 0 && r > resp[maxima[j]]) { maxima[j + 1] = maxima[j]; j--; }
        maxima[j + 1] = m;
    }
    if (nm > n_keypoints) nm = n_keypoints;
    int nb = 0;
    for (int i = 0; i < nm; i++) {
        if (resp[maxima[i]] < min_response) continue;
        corners[maxima[i]] = 1; nb++;
    }
    free(maxima);
    return nb;
}

/* detect, extractor.jl:63-95 */
int orc_detect(const double *img, int H, int W, const double *cur_yx, int n_cur,
               int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
               double sigma_mask, double min_response, int64_t *out_rc, int cap)
{
    if (n_cur >= max_points) return 0;
    size_t N = (size_t)H * W;
    double *image = (double *)malloc(N * sizeof(double));
    memcpy(image, img, N * sizeof(double));
    if (n_cur > 0) {
        double *mask = (double *)malloc(N * sizeof(double));
        orc_get_mask(mask, H, W, cur_yx, n_cur, radius);
        if (sigma_mask != 0) { /* `σ_mask ≉ 0`: isapprox against 0 with atol=0 is exact equality */
            double taps[128];
            int l = orc_gaussian_taps(sigma_mask, taps);
            orc_imfilter_sep(mask, mask, H, W, taps, l, taps, l, 0);
        }
        for (size_t i = 0; i < N; i++) image[i] = image[i] * mask[i];
        free(mask);
    }
    int n_cells = grid_rows * grid_cols;
    int n_detect = max_points - n_cur;
    int n_cell_detect = (int)ceil((double)n_detect / (double)n_cells);

    uint8_t *corners = (uint8_t *)malloc((size_t)cell_size * cell_size);
    double *resp = (double *)malloc(sizeof(double) * cell_size * cell_size);
    int n_out = 0;
    for (int y = 0; y < grid_rows; y++)
        for (int x = 0; x < grid_cols; x++) {
            int y_shift = y * cell_size, x_shift = x * cell_size;
            int y_end = (y + 1) * cell_size < H ? (y + 1) * cell_size : H;
            int x_end = (x + 1) * cell_size < W ? (x + 1) * cell_size : W;
            int h = y_end - y_shift, w = x_end - x_shift;
            if (h <= 0 || w <= 0) continue; /* empty range: shi_tomasi of an empty view finds nothing */
            int nd = orc_shi_tomasi_cell(corners, resp, image + IDX(y_shift, x_shift, H), h, w, H,
                                         n_cell_detect, min_response);
            if (nd == 0) continue;
            for (int cx = 0; cx < w; cx++)
                for (int cy = 0; cy < h; cy++)
                    if (corners[IDX(cy, cx, h)]) {
                        if (n_out >= cap) { n_out = -1; goto done; }
                        out_rc[2 * n_out] = cy + 1 + y_shift;
                        out_rc[2 * n_out + 1] = cx + 1 + x_shift;
                        n_out++;
                    }
        }
done:
    free(corners); free(resp); free(image);
    return n_out;
}

/* ImageFeatures.create_descriptor(img, keypoints, BRIEF):
 * blur with gaussian(sigma) taps of length `window`, keep keypoints whose
 * +-ceil(window/2) box is in bounds, bit k = I[kp+s1_k] < I[kp+s2_k].      */
int orc_describe(const double *img, int H, int W, const int64_t *rc, int n,
                 const int32_t *pattern, int n_bits, double sigma, int window,
                 uint64_t *out_bits, int64_t *out_rc)
{
    size_t N = (size_t)H * W;
    double *sm = (double *)malloc(N * sizeof(double));
    double taps[64];
    int hw = window >> 1; double s = 0;
    for (int i = 0; i < window; i++) { double x = i - hw; taps[i] = exp(-(x * x) / (2.0 * (sigma * sigma))); s += taps[i]; }
    for (int i = 0; i < window; i++) taps[i] = taps[i] / s;
    orc_imfilter_sep(sm, img, H, W, taps, window, taps, window, 0);
    int lim = (int)ceil(window / 2.0);
    int words = n_bits / 64, n_out = 0;
    for (int k = 0; k < n; k++) {
        long y = rc[2 * k], x = rc[2 * k + 1];
        if (y - lim < 1 || y + lim > H || x - lim < 1 || x + lim > W) continue;
        uint64_t *dst = out_bits + (size_t)n_out * words;
        memset(dst, 0, sizeof(uint64_t) * words);
        for (int b = 0; b < n_bits; b++) {
            const int32_t *p = pattern + 4 * b;
            double v1 = sm[IDX(y - 1 + p[0], x - 1 + p[1], H)];
            double v2 = sm[IDX(y - 1 + p[2], x - 1 + p[3], H)];
            if (v1 < v2) dst[b >> 6] |= (uint64_t)1 << (b & 63);
        }
        out_rc[2 * n_out] = y; out_rc[2 * n_out + 1] = x;
        n_out++;
    }
    free(sm);
    return n_out;
}

/* ------------------------------------------------------------------------ */
/* LK pyramid */
int64_t orc_pyr_layout(int H, int W, int total_levels, int *Hs, int *Ws, int64_t *off)
{
    int64_t o = 0;
    for (int l = 0; l < total_levels; l++) {
        Hs[l] = H; Ws[l] = W; off[l] = o;
        o += (int64_t)H * W;
        H = (H + 1) / 2; W = (W + 1) / 2; /* ceil(size/2), Images.pyramid_scale / pyramid.jl sizes */
    }
    off[total_levels] = o;
    return o;
}

/* integral_image!, lucas_kanade.jl:131-138: cumsum dim 1 then dim 2 */
static void integral_image(double *out, const double *in, int H, int W)
{
    for (int x = 0; x < W; x++) {
        double acc = in[IDX(0, x, H)];
        out[IDX(0, x, H)] = acc;
        for (int y = 1; y < H; y++) { acc = acc + in[IDX(y, x, H)]; out[IDX(y, x, H)] = acc; }
    }
    for (int x = 1; x < W; x++)
        for (int y = 0; y < H; y++) out[IDX(y, x, H)] = out[IDX(y, x - 1, H)] + out[IDX(y, x, H)];
}

/* compute_partial_derivatives!, lucas_kanade.jl:109-129 (sigma = 4, replicate) */
static void partial_derivatives(double *Iyy, double *Ixx, double *Iyx, const double *Iy, const double *Ix, int H, int W)
{
    size_t N = (size_t)H * W;
    double *sq = (double *)malloc(N * sizeof(double));
    for (size_t i = 0; i < N; i++) sq[i] = Iy[i] * Iy[i];
    orc_iir_gaussian(sq, sq, H, W, 4.0, 0); integral_image(Iyy, sq, H, W);
    for (size_t i = 0; i < N; i++) sq[i] = Ix[i] * Ix[i];
    orc_iir_gaussian(sq, sq, H, W, 4.0, 0); integral_image(Ixx, sq, H, W);
    for (size_t i = 0; i < N; i++) sq[i] = Iy[i] * Ix[i];
    orc_iir_gaussian(sq, sq, H, W, 4.0, 0); integral_image(Iyx, sq, H, W);
    free(sq);
}

void orc_pyr_build(orc_pyr *p, const double *img, double sigma, int mode)
{
    const double d[3] = {-1.0 / 2, 0.0 / 2, 1.0 / 2}, s[3] = {3.0 / 16, 10.0 / 16, 3.0 / 16}; /* KernelFactors.scharr */
    memcpy(p->layers, img, sizeof(double) * (size_t)p->H[0] * p->W[0]); /* copy!(pyramid[1], img) */
    for (int l = 0; l + 1 < p->levels; l++) {
        int H = p->H[l], W = p->W[l];
        double *tmp = (double *)malloc(sizeof(double) * (size_t)H * W);
        /* pyramid.jl:119/131 (replicate) or Images.gaussian_pyramid (NA()) */
        orc_iir_gaussian(tmp, p->layers + p->off[l], H, W, sigma, mode == 0 ? 2 : 0);
        orc_imresize(p->layers + p->off[l + 1], p->H[l + 1], p->W[l + 1], tmp, H, W);
        free(tmp);
    }
    for (int l = 0; l < p->levels; l++) {
        int H = p->H[l], W = p->W[l];
        const double *L = p->layers + p->off[l];
        int border = mode == 0 ? 1 : 0; /* pyramid.jl:51,59 Fill(0) vs :100-101 replicate */
        orc_imfilter_sep(p->Iy + p->off[l], L, H, W, d, 3, s, 3, border);
        orc_imfilter_sep(p->Ix + p->off[l], L, H, W, s, 3, d, 3, border);
        partial_derivatives(p->Iyy + p->off[l], p->Ixx + p->off[l], p->Iyx + p->off[l],
                            p->Iy + p->off[l], p->Ix + p->off[l], H, W);
    }
}

void orc_pyr_build_flat(const double *img, int H, int W, int total_levels, double sigma, int mode,
                        double *layers, double *Iy, double *Ix, double *Iyy, double *Ixx, double *Iyx)
{
    orc_pyr p; p.levels = total_levels;
    orc_pyr_layout(H, W, total_levels, p.H, p.W, p.off);
    p.layers = layers; p.Iy = Iy; p.Ix = Ix; p.Iyy = Iyy; p.Ixx = Ixx; p.Iyx = Iyx;
    orc_pyr_build(&p, img, sigma, mode);
}
