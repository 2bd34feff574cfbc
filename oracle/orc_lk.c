/*
 * orc_lk.c -- oracle (TEST INFRASTRUCTURE ONLY, parity unpinned; see
 * slam_oracle.h): pyramidal Lucas-Kanade and forward-backward tracking.
 *
 * Follows /root/reference/src/optical_flow/lucas_kanade.jl:1-100,140-212,
 * src/optical_flow/utils.jl:5-45 and src/tracker.jl:17-82.
 */
#include "slam_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(y, x, H) ((size_t)(y) + (size_t)(x) * (size_t)(H))

/* svd2x2, utils.jl:5-31.  Matrices column-major: M[0]=M11, M[1]=M21, M[2]=M12, M[3]=M22 */
void orc_svd2x2(const double M[4], double U[4], double S[2], double V[4])
{
    double E = (M[0] + M[3]) / 2, F = (M[0] - M[3]) / 2;
    double G = (M[1] + M[2]) / 2, H = (M[1] - M[2]) / 2;
    double Q = sqrt(E * E + H * H), R = sqrt(F * F + G * G);
    double sx = Q + R, sy = Q - R;
    double a1 = atan2(G, F), a2 = atan2(H, E);
    double th = (a2 - a1) / 2, ph = (a2 + a1) / 2;
    double s = (sy > 0) - (sy < 0); /* sign(sy), sign(0)=0 */
    double sp = sin(ph), cp = cos(ph), st = sin(th), ct = cos(th);
    U[0] = cp; U[1] = sp; U[2] = -s * sp; U[3] = s * cp;
    S[0] = sx; S[1] = fabs(sy);
    V[0] = ct; V[1] = -st; V[2] = st; V[3] = ct;
}

/* pinv2x2, utils.jl:35-45: U * D * V' with tol = sqrt(eps) */
void orc_pinv2x2(const double M[4], double Gi[4], double S[2])
{
    double U[4], V[4];
    orc_svd2x2(M, U, S, V);
    const double tol = 1.4901161193847656e-08; /* sqrt(eps(Float64)) */
    double d1 = S[0] > tol ? 1.0 / S[0] : 0.0, d2 = S[1] > tol ? 1.0 / S[1] : 0.0;
    /* UD = U*D (StaticArrays: row . column, left to right) */
    double ud11 = U[0] * d1 + U[2] * 0.0, ud21 = U[1] * d1 + U[3] * 0.0;
    double ud12 = U[0] * 0.0 + U[2] * d2, ud22 = U[1] * 0.0 + U[3] * d2;
    /* (UD)*V' : [i,j] = UD[i,1]*V[j,1] + UD[i,2]*V[j,2] */
    Gi[0] = ud11 * V[0] + ud12 * V[2];
    Gi[1] = ud21 * V[0] + ud22 * V[2];
    Gi[2] = ud11 * V[1] + ud12 * V[3];
    Gi[3] = ud21 * V[1] + ud22 * V[3];
}

/* Images.boxdiff, 1-based inclusive ranges */
static double boxdiff(const double *I, int H, int y1, int y2, int x1, int x2)
{
    double sum = I[IDX(y2 - 1, x2 - 1, H)];
    sum -= x1 > 1 ? I[IDX(y2 - 1, x1 - 2, H)] : 0.0;
    sum -= y1 > 1 ? I[IDX(y1 - 2, x2 - 1, H)] : 0.0;
    sum += (y1 > 1 && x1 > 1) ? I[IDX(y1 - 2, x1 - 2, H)] : 0.0;
    return sum;
}

typedef struct { int up, down, left, right; } offs_t;

/* get_offsets, lucas_kanade.jl:199-208 (image axes 1:H, 1:W) */
static offs_t get_offsets(const long point[2], const double np[2], int window, int H, int W)
{
    offs_t o;
    double p0 = (double)point[0], p1 = (double)point[1];
    o.up = (int)floor(fmin((double)window, fmin(p0, np[0]) - 1));
    o.down = (int)floor(fmin((double)window, (double)H - fmax(p0, np[0])));
    o.left = (int)floor(fmin((double)window, fmin(p1, np[1]) - 1));
    o.right = (int)floor(fmin((double)window, (double)W - fmax(p1, np[1])));
    return o;
}

/* compute_spatial_gradient, lucas_kanade.jl:140-157; grid = point + offsets */
static void spatial_gradient(const orc_pyr *p, int lv, const long point[2], offs_t o,
                             double Ginv[4], double *min_eig)
{
    int H = p->H[lv];
    int y1 = (int)point[0] - o.up, y2 = (int)point[0] + o.down;
    int x1 = (int)point[1] - o.left, x2 = (int)point[1] + o.right;
    double syy = boxdiff(p->Iyy + p->off[lv], H, y1, y2, x1, x2);
    double sxx = boxdiff(p->Ixx + p->off[lv], H, y1, y2, x1, x2);
    double syx = boxdiff(p->Iyx + p->off[lv], H, y1, y2, x1, x2);
    double G[4] = {syy, syx, syx, sxx}, S[2];
    orc_pinv2x2(G, Ginv, S);
    double cnt = (double)((long)(y2 - y1 + 1) * (long)(x2 - x1 + 1));
    *min_eig = fmin(S[0], S[1]) / cnt;
}

/* prepare_linear_system + compute_flow_vector, lucas_kanade.jl:159-187 */
static void flow_vector(const orc_pyr *first, const orc_pyr *second, int lv, const long point[2],
                        const double corr[2], offs_t o, const double Ginv[4], int sum_order, double flow[2])
{
    int H = first->H[lv], W = first->W[lv];
    const double *A = first->layers + first->off[lv];
    const double *Iy = first->Iy + first->off[lv], *Ix = first->Ix + first->off[lv];
    const double *B = second->layers + second->off[lv];
    int P = o.up + o.down + 1, Q = o.left + o.right + 1;
    double by = 0.0, bx = 0.0;
    if (sum_order == 0) {
        for (int q = 0; q < Q; q++)
            for (int p = 0; p < P; p++) {
                double r = corr[0] + (double)(p - o.up), c = corr[1] + (double)(q - o.left);
                size_t a = IDX(point[0] - o.up + p - 1, point[1] - o.left + q - 1, H);
                double dI = A[a] - orc_bilinear(B, H, W, r, c);
                by += dI * Iy[a];
                bx += dI * Ix[a];
            }
    } else {
        /* sum_order 1: the element e of the (q outer, p inner) enumeration goes to lane e % 64, a lane adds its elements in order, the
         * 64 partial sums are folded by the xor butterfly m = 32, 16, 1, 2, 4, 8 (rounds 1-3 of the device kernel: one wave per keypoint);
         * sum_order 2: lane e % 32 and the butterfly m = 16, 1, 2, 4, 8 -- the order of the PARKED half-wave experiment
         * (scripts/ubench/lk_halfwave.hip.txt, measured 1.6x slower, never shipped): no product kernel sums in this order; the bit-exact GPU
         * tests stay on sum_order 1 (csrc/lk.hip: wave_sum2) */
        const int NL = sum_order == 2 ? 32 : 64;
        double ay[64], ax[64];
        for (int l = 0; l < 64; l++) ay[l] = ax[l] = 0.0;
        int e = 0;
        for (int q = 0; q < Q; q++)
            for (int p = 0; p < P; p++, e++) {
                double r = corr[0] + (double)(p - o.up), c = corr[1] + (double)(q - o.left);
                size_t a = IDX(point[0] - o.up + p - 1, point[1] - o.left + q - 1, H);
                double dI = A[a] - orc_bilinear(B, H, W, r, c);
                ay[e % NL] += dI * Iy[a];
                ax[e % NL] += dI * Ix[a];
            }
        static const int order1[6] = {32, 16, 1, 2, 4, 8}, order2[5] = {16, 1, 2, 4, 8};
        const int *order = sum_order == 2 ? order2 : order1, nsteps = sum_order == 2 ? 5 : 6;
        for (int mi = 0; mi < nsteps; mi++) {
            const int m = order[mi];
            double ty[64], tx[64];
            for (int l = 0; l < NL; l++) { ty[l] = ay[l] + ay[l ^ m]; tx[l] = ax[l] + ax[l ^ m]; }
            memcpy(ay, ty, sizeof(double) * NL); memcpy(ax, tx, sizeof(double) * NL);
        }
        by = ay[0]; bx = ax[0];
    }
    flow[0] = Ginv[0] * by + Ginv[2] * bx;
    flow[1] = Ginv[1] * by + Ginv[3] * bx;
}

static int lies_in(int H, int W, const double p[2])
{
    return 1.0 <= p[0] && p[0] <= (double)H && 1.0 <= p[1] && p[1] <= (double)W;
}

static int offs_eq(offs_t a, offs_t b) { return a.up == b.up && a.down == b.down && a.left == b.left && a.right == b.right; }

/* optflow!, lucas_kanade.jl:9-100 */
int orc_optflow(double *disp, const orc_pyr *first, const orc_pyr *second,
                const double *pts, int n, int iterations, int window, int pyramid_levels,
                double eig_thr, double eps, uint8_t *status, int sum_order, int threads)
{
    if (!(first->levels > pyramid_levels && second->levels > pyramid_levels)) return -1;
    for (int i = 0; i < n; i++) status[i] = 1;
    (void)threads;
    for (int level = pyramid_levels + 1; level >= 1; level--) {
        int lv = level - 1;
        int H = first->H[lv], W = first->W[lv];
        double scale = ldexp(1.0, level - 1); /* 2^(level-1) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads > 0 ? threads : 1)
#endif
        for (int i = 0; i < n; i++) {
            if (!status[i]) continue;
            long point[2] = {(long)floor(pts[2 * i] / scale), (long)floor(pts[2 * i + 1] / scale)};
            double pf[2] = {(double)point[0], (double)point[1]};
            offs_t offsets = get_offsets(point, pf, window, H, W);
            double Ginv[4], min_eig;
            spatial_gradient(first, lv, point, offsets, Ginv, &min_eig);
            if (min_eig < eig_thr) { status[i] = 0; continue; }
            double contrib[2] = {0.0, 0.0};
            for (int it = 0; it < iterations; it++) {
                double pflow[2] = {disp[2 * i] + contrib[0], disp[2 * i + 1] + contrib[1]};
                double corr[2] = {pf[0] + pflow[0], pf[1] + pflow[1]};
                if (!lies_in(H, W, corr)) { status[i] = 0; break; }
                offs_t no = get_offsets(point, corr, window, H, W);
                if (!offs_eq(no, offsets)) {
                    offsets = no;
                    spatial_gradient(first, lv, point, offsets, Ginv, &min_eig);
                    if (min_eig < eig_thr) { status[i] = 0; break; }
                }
                double flow[2];
                flow_vector(first, second, lv, point, corr, offsets, Ginv, sum_order, flow);
                if (fabs(flow[0]) < eps && fabs(flow[1]) < eps) break;
                contrib[0] += flow[0]; contrib[1] += flow[1];
                double nx[2] = {corr[0] + flow[0], corr[1] + flow[1]};
                if (!lies_in(H, W, nx)) { status[i] = 0; break; }
            }
            if (!status[i]) continue;
            disp[2 * i] += contrib[0]; disp[2 * i + 1] += contrib[1];
            if (level > 1) { disp[2 * i] *= 2.0; disp[2 * i + 1] *= 2.0; }
        }
    }
    int n_good = 0;
    for (int i = 0; i < n; i++) n_good += status[i];
    return n_good;
}

/* fb_tracking!, tracker.jl:17-66 */
int orc_fb_tracking(const orc_pyr *prev, const orc_pyr *cur, const double *pts,
                    const double *disp0, int n, int iterations, int window, int pyramid_levels,
                    double eig_thr, double eps, double max_distance,
                    double *out, uint8_t *status, int sum_order, int threads)
{
    if (n == 0) return 0;
    double *disp = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    if (disp0) memcpy(disp, disp0, sizeof(double) * 2 * (size_t)n);
    else memset(disp, 0, sizeof(double) * 2 * (size_t)n);
    int n_good = orc_optflow(disp, prev, cur, pts, n, iterations, window, pyramid_levels, eig_thr, eps,
                             status, sum_order, threads);
    if (n_good < 0) { free(disp); return -1; }
    int *valid_ids = (int *)malloc(sizeof(int) * (size_t)(n_good + 1));
    double *vc = (double *)malloc(sizeof(double) * 2 * (size_t)(n_good + 1));
    double *bd = (double *)malloc(sizeof(double) * 2 * (size_t)(n_good + 1));
    uint8_t *bs = (uint8_t *)malloc((size_t)(n_good + 1));
    const int back_levels = 0;                   /* tracker.jl:34 */
    const double scale = 1.0 / ldexp(1.0, back_levels);
    int c = 0;
    for (int i = 0; i < n; i++) {
        if (!status[i]) continue;
        double np0 = pts[2 * i] + disp[2 * i], np1 = pts[2 * i + 1] + disp[2 * i + 1];
        out[2 * i] = np0; out[2 * i + 1] = np1;
        vc[2 * c] = np0; vc[2 * c + 1] = np1;
        bd[2 * c] = -disp[2 * i] * scale; bd[2 * c + 1] = -disp[2 * i + 1] * scale;
        valid_ids[c] = i; c++;
    }
    /* back_algorithm keeps iterations/window/eig threshold; eps falls back to
     * the LucasKanade default 1e-2 (tracker.jl:51-54, lucas_kanade.jl:6) */
    orc_optflow(bd, cur, prev, vc, c, iterations, window, back_levels, eig_thr, 1e-2, bs, sum_order, threads);
    for (int k = 0; k < c; k++) {
        int idx = valid_ids[k];
        if (!bs[k]) { status[idx] = 0; continue; }
        double b0 = vc[2 * k] + bd[2 * k], b1 = vc[2 * k + 1] + bd[2 * k + 1];
        double d0 = pts[2 * idx] - b0, d1 = pts[2 * idx + 1] - b1;
        /* StaticArrays norm of a 2-vector: sqrt(abs2 + abs2), no rescaling */
        if (sqrt(d0 * d0 + d1 * d1) >= max_distance) status[idx] = 0;
    }
    free(disp); free(valid_ids); free(vc); free(bd); free(bs);
    return 0;
}

int orc_fb_tracking_flat(int H, int W, int total_levels,
                         const double *p_layers, const double *p_Iy, const double *p_Ix,
                         const double *p_Iyy, const double *p_Ixx, const double *p_Iyx,
                         const double *c_layers, const double *c_Iy, const double *c_Ix,
                         const double *c_Iyy, const double *c_Ixx, const double *c_Iyx,
                         const double *pts_yx, const double *disp0_yx, int n,
                         int iterations, int window, int pyramid_levels,
                         double eig_thr, double eps, double max_distance,
                         double *out_yx, uint8_t *status, int sum_order, int threads)
{
    orc_pyr p, c;
    p.levels = c.levels = total_levels;
    orc_pyr_layout(H, W, total_levels, p.H, p.W, p.off);
    orc_pyr_layout(H, W, total_levels, c.H, c.W, c.off);
    p.layers = (double *)p_layers; p.Iy = (double *)p_Iy; p.Ix = (double *)p_Ix;
    p.Iyy = (double *)p_Iyy; p.Ixx = (double *)p_Ixx; p.Iyx = (double *)p_Iyx;
    c.layers = (double *)c_layers; c.Iy = (double *)c_Iy; c.Ix = (double *)c_Ix;
    c.Iyy = (double *)c_Iyy; c.Ixx = (double *)c_Ixx; c.Iyx = (double *)c_Iyx;
    return orc_fb_tracking(&p, &c, pts_yx, disp0_yx, n, iterations, window, pyramid_levels,
                           eig_thr, eps, max_distance, out_yx, status, sum_order, threads);
}
