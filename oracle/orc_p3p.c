/* orc_p3p.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned): P3P RANSAC of compute_pose!
 * (reference: src/front_end.jl:132-219, the call `p3p_ransac(points, pixels, pdn, K; threshold)` at :164-167 and
 * what it consumes of the result at :174-186).
 *
 * `p3p_ransac` lives in the un-vendored dependency RecoverPose 0.1 (Project.toml:22,41; not under /root/reference):
 * which minimal solver it uses, how it draws samples (Julia's RNG) and how many iterations it runs are not visible
 * here.  What is restated is the published structure -- minimal 3-point absolute pose, hypotheses scored by
 * reprojection error against `threshold`, the hypothesis with most inliers wins -- with these explicit choices:
 *   - minimal solver: Grunert's distance formulation (Haralick et al. 1994, "Review and analysis of solutions of
 *     the three point perspective pose estimation problem"), the quartic built by polynomial arithmetic from
 *     u = N(v)/D(v), solved by Ferrari's factorisation; the resolvent cubic's root comes from a safeguarded
 *     Newton/bisection, so the whole solver uses only + - * / sqrt (IEEE-exact on CPU and GPU alike);
 *   - sample triples are SUPPLIED BY THE CALLER (like the BRIEF pattern): no RNG inside, all of them are scored;
 *   - winner = most inliers, ties to the lower iteration, then the lower solution index;
 *   - inlier: depth > 0 and reprojection error < threshold.
 * The reference has no test or golden vector for this: parity unpinned; tests pin this file against ground-truth
 * scenes (the true pose must be recovered) and numpy.roots. */
#include "slam_oracle.h"
#include <math.h>

/* a real root >= 0 of z^3 + B z^2 + C z + D with D <= 0: safeguarded Newton inside [0, 1 + max|coef|] */
static double cubic_root_nonneg(double B, double C, double D)
{
    double hi = fabs(B);
    if (fabs(C) > hi) hi = fabs(C);
    if (fabs(D) > hi) hi = fabs(D);
    hi = hi + 1.0;
    double lo = 0.0;                       /* f(lo) <= 0 < f(hi) */
    double x = hi;
    for (int it = 0; it < 200; it++) {
        const double f = ((x + B) * x + C) * x + D;
        if (f == 0.0) return x;
        if (f > 0.0) hi = x; else lo = x;
        const double df = (3.0 * x + 2.0 * B) * x + C;
        double xn = x - f / df;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (xn == x || xn == lo || xn == hi) return xn;
        x = xn;
    }
    return x;
}

/* real roots of A[4] x^4 + A[3] x^3 + A[2] x^2 + A[1] x + A[0]; returns their number (0..4) */
int orc_quartic_real_roots(const double A[5], double roots[4])
{
    if (!(fabs(A[4]) > 0.0)) return 0;
    const double a = A[3] / A[4], b = A[2] / A[4], c = A[1] / A[4], d = A[0] / A[4];
    if (!isfinite(a) || !isfinite(b) || !isfinite(c) || !isfinite(d)) return 0;
    const double a2 = a * a;
    const double p = b - 0.375 * a2;
    const double q = (c - 0.5 * a * b) + 0.125 * a2 * a;
    const double r = ((d - 0.25 * a * c) + 0.0625 * a2 * b) - 0.01171875 * a2 * a2;
    const double sh = 0.25 * a;
    double y[4];
    int n = 0;
    const double z = cubic_root_nonneg(2.0 * p, p * p - 4.0 * r, -(q * q));
    if (z > 0.0) {
        const double s = sqrt(z), h = 0.5 * (p + z), g = q / (2.0 * s);
        const double d1 = z - 4.0 * (h - g), d2 = z - 4.0 * (h + g);
        if (d1 >= 0.0) { const double w = sqrt(d1); y[n++] = 0.5 * (-s + w); y[n++] = 0.5 * (-s - w); }
        if (d2 >= 0.0) { const double w = sqrt(d2); y[n++] = 0.5 * (s + w); y[n++] = 0.5 * (s - w); }
    } else {                               /* q == 0: biquadratic */
        const double disc = p * p - 4.0 * r;
        if (disc >= 0.0) {
            const double w = sqrt(disc), t1 = 0.5 * (-p + w), t2 = 0.5 * (-p - w);
            if (t1 >= 0.0) { const double e = sqrt(t1); y[n++] = e; y[n++] = -e; }
            if (t2 >= 0.0) { const double e = sqrt(t2); y[n++] = e; y[n++] = -e; }
        }
    }
    int m = 0;
    for (int i = 0; i < n; i++) {
        double x = y[i] - sh;
        for (int k = 0; k < 4; k++) {      /* four Newton steps on the original polynomial */
            const double f = (((A[4] * x + A[3]) * x + A[2]) * x + A[1]) * x + A[0];
            const double df = ((4.0 * A[4] * x + 3.0 * A[3]) * x + 2.0 * A[2]) * x + A[1];
            const double xn = x - f / df;
            if (isfinite(xn)) x = xn;
        }
        if (isfinite(x)) roots[m++] = x;
    }
    return m;
}

static void v3_sub(const double *a, const double *b, double *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static double v3_dot(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static void v3_cross(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static int v3_unit(const double *a, double *o)
{
    const double n = sqrt(v3_dot(a, a));
    if (!(n > 0.0)) return 0;
    const double i = 1.0 / n;
    o[0] = a[0] * i; o[1] = a[1] * i; o[2] = a[2] * i;
    return 1;
}
/* right-handed orthonormal frame (columns e1,e2,e3) of the triangle p1,p2,p3; 0 if degenerate */
static int tri_frame(const double *p1, const double *p2, const double *p3, double E[9])
{
    double d12[3], d13[3], w[3];
    v3_sub(p2, p1, d12); v3_sub(p3, p1, d13);
    if (!v3_unit(d12, E)) return 0;
    v3_cross(E, d13, w);
    if (!v3_unit(w, E + 6)) return 0;
    v3_cross(E + 6, E, E + 3);
    return 1;
}

/* Minimal solver.  X: the three world points (3x3, one per row), F: their bearing vectors in the camera frame
 * (normalised here).  Rt: up to 4 poses, each 12 doubles column-major 3x4 [R | t] with  x_cam = R X + t.
 * Returns the number of poses. */
int orc_p3p_solve(const double X[9], const double F[9], double Rt[48])
{
    double f1[3], f2[3], f3[3];
    if (!v3_unit(F, f1) || !v3_unit(F + 3, f2) || !v3_unit(F + 6, f3)) return 0;
    double t[3];
    v3_sub(X + 3, X + 6, t); const double a2 = v3_dot(t, t);      /* |X2 - X3|^2 */
    v3_sub(X, X + 6, t);     const double b2 = v3_dot(t, t);      /* |X1 - X3|^2 */
    v3_sub(X, X + 3, t);     const double c2 = v3_dot(t, t);      /* |X1 - X2|^2 */
    if (!(a2 > 0.0 && b2 > 0.0 && c2 > 0.0)) return 0;
    const double ca = v3_dot(f2, f3), cb = v3_dot(f1, f3), cg = v3_dot(f1, f2);
    /* s2 = u s1, s3 = v s1;  u = N(v)/D(v),  N = n2 v^2 + n1 v + n0,  D = d1 v + d0 */
    const double k = (a2 - c2) / b2, m = c2 / b2;
    const double N[3] = {1.0 + k, -2.0 * k * cb, k - 1.0};
    const double D[2] = {2.0 * cg, -2.0 * ca};
    /* Q(v) = D^2 + N^2 - 2 cg N D - m (1 - 2 cb v + v^2) D^2 */
    double DD[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
    double NN[5] = {N[0] * N[0], 2.0 * N[0] * N[1], 2.0 * N[0] * N[2] + N[1] * N[1], 2.0 * N[1] * N[2], N[2] * N[2]};
    double ND[4] = {N[0] * D[0], N[0] * D[1] + N[1] * D[0], N[1] * D[1] + N[2] * D[0], N[2] * D[1]};
    const double Wp[3] = {1.0, -2.0 * cb, 1.0};
    double WD[5] = {Wp[0] * DD[0], Wp[0] * DD[1] + Wp[1] * DD[0], (Wp[0] * DD[2] + Wp[1] * DD[1]) + Wp[2] * DD[0],
                    Wp[1] * DD[2] + Wp[2] * DD[1], Wp[2] * DD[2]};
    double Q[5];
    for (int i = 0; i < 5; i++) {
        const double dd = i < 3 ? DD[i] : 0.0, nd = i < 4 ? ND[i] : 0.0;
        Q[i] = ((dd + NN[i]) - 2.0 * cg * nd) - m * WD[i];
    }
    double vr[4];
    const int nr = orc_quartic_real_roots(Q, vr);
    double Ew[9];
    if (!tri_frame(X, X + 3, X + 6, Ew)) return 0;
    int ns = 0;
    for (int i = 0; i < nr; i++) {
        const double v = vr[i];
        if (!(v > 0.0)) continue;
        const double den = D[1] * v + D[0];
        const double u = ((N[2] * v + N[1]) * v + N[0]) / den;
        if (!(u > 0.0) || !isfinite(u)) continue;
        const double w = (1.0 + v * v) - 2.0 * v * cb;
        if (!(w > 0.0)) continue;
        const double s1 = sqrt(b2 / w), s2 = u * s1, s3 = v * s1;
        if (!isfinite(s1) || !(s1 > 0.0)) continue;
        const double Y1[3] = {s1 * f1[0], s1 * f1[1], s1 * f1[2]};
        const double Y2[3] = {s2 * f2[0], s2 * f2[1], s2 * f2[2]};
        const double Y3[3] = {s3 * f3[0], s3 * f3[1], s3 * f3[2]};
        double Ec[9];
        if (!tri_frame(Y1, Y2, Y3, Ec)) continue;
        double *P = Rt + 12 * ns;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                P[r + 3 * c] = (Ec[r] * Ew[c] + Ec[3 + r] * Ew[3 + c]) + Ec[6 + r] * Ew[6 + c];
        for (int r = 0; r < 3; r++)
            P[9 + r] = Y1[r] - ((P[r] * X[0] + P[r + 3] * X[1]) + P[r + 6] * X[2]);
        int fin = 1;
        for (int j = 0; j < 12; j++) fin &= isfinite(P[j]) != 0;
        if (fin) ns++;
    }
    return ns;
}

/* reprojection error of X under pose P (3x4 column-major) and K (3x3 column-major; fx=K[0], fy=K[4], cx=K[6],
 * cy=K[7]); px = (x, y).  Returns a negative number when the point is not in front of the camera. */
static double p3p_reproj(const double *P, const double *K, const double *X, const double *px)
{
    const double xc = ((P[0] * X[0] + P[3] * X[1]) + P[6] * X[2]) + P[9];
    const double yc = ((P[1] * X[0] + P[4] * X[1]) + P[7] * X[2]) + P[10];
    const double zc = ((P[2] * X[0] + P[5] * X[1]) + P[8] * X[2]) + P[11];
    if (!(zc > 0.0)) return -1.0;
    const double iz = 1.0 / zc;
    const double dx = px[0] - (K[0] * xc * iz + K[6]), dy = px[1] - (K[4] * yc * iz + K[7]);
    return sqrt(dx * dx + dy * dy);
}

/* p3p_ransac (front_end.jl:164-167).  pts3d n x 3, px_xy n x 2 (x, y), pdn n x 3 (bearing vectors), K 3x3
 * column-major, samples iters x 3 (0-based point indices).  Outputs: KP = K [R | t] (3x4 column-major), Rt,
 * inliers (n bytes), *error = sum of the inliers' reprojection errors (index order), *best_iter.
 * Returns the inlier count of the winner (0: no hypothesis). */
int orc_p3p_ransac(const double *pts3d, const double *px_xy, const double *pdn, int n, const double *K, double threshold,
                   const int32_t *samples, int iters, double *KP, double *Rt_out, unsigned char *inliers, double *error,
                   int *best_iter)
{
    int best = 0, bi = -1;
    double bestP[12] = {0};
    for (int it = 0; it < iters; it++) {
        const int i0 = samples[3 * it], i1 = samples[3 * it + 1], i2 = samples[3 * it + 2];
        if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= n || i1 >= n || i2 >= n || i0 == i1 || i0 == i2 || i1 == i2) continue;
        double X[9], F[9], Rt[48];
        for (int j = 0; j < 3; j++) {
            X[j] = pts3d[3 * i0 + j]; X[3 + j] = pts3d[3 * i1 + j]; X[6 + j] = pts3d[3 * i2 + j];
            F[j] = pdn[3 * i0 + j]; F[3 + j] = pdn[3 * i1 + j]; F[6 + j] = pdn[3 * i2 + j];
        }
        const int ns = orc_p3p_solve(X, F, Rt);
        for (int s = 0; s < ns; s++) {
            int cnt = 0;
            for (int i = 0; i < n; i++) {
                const double e = p3p_reproj(Rt + 12 * s, K, pts3d + 3 * i, px_xy + 2 * i);
                cnt += (e >= 0.0 && e < threshold);
            }
            if (cnt > best) { best = cnt; bi = it; for (int j = 0; j < 12; j++) bestP[j] = Rt[12 * s + j]; }
        }
    }
    if (best_iter) *best_iter = bi;
    double esum = 0.0;
    for (int i = 0; i < n; i++) {
        double e = -1.0;
        if (best > 0) e = p3p_reproj(bestP, K, pts3d + 3 * i, px_xy + 2 * i);
        const int in = best > 0 && e >= 0.0 && e < threshold;
        inliers[i] = (unsigned char)in;
        if (in) esum += e;
    }
    if (error) *error = esum;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 3; r++) {
            KP[r + 3 * c] = (K[r] * bestP[3 * c] + K[r + 3] * bestP[3 * c + 1]) + K[r + 6] * bestP[3 * c + 2];
            if (Rt_out) Rt_out[r + 3 * c] = bestP[r + 3 * c];
        }
    return best;
}
