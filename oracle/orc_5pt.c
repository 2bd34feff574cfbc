/* orc_5pt.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned): five-point essential-matrix RANSAC of
 * compute_pose_5pt! (reference: src/front_end.jl:243-332; the call `five_point_ransac(previous_points,
 * current_points, previous_pd, current_pd, K, K, cache; max_repr_error)` at :305-308, result consumed at :305-331:
 * `n_inliers, (_, P, inliers, _)`).
 *
 * `five_point_ransac` lives in the un-vendored dependency RecoverPose 0.1 (Project.toml:22,41): its solver variant,
 * sampler, iteration count and inlier rule are not visible here.  Restated from the published algorithm with these
 * explicit choices:
 *   - minimal solver: Nister, "An efficient solution to the five-point relative pose problem" (PAMI 2004): null
 *     space of the 5x9 epipolar system, the ten cubic constraints (det E = 0, 2 E E'E - tr(E E')E = 0) expanded by
 *     polynomial arithmetic on the basis, Gauss-Jordan elimination in Nister's monomial order, the 3x3 polynomial
 *     matrix B(z), its determinant (degree 10), real roots by bracketing between the roots of the derivative
 *     (safeguarded Newton/bisection), back-substitution -- only + - * / sqrt throughout;
 *   - pose from E in closed form (Horn 1990: b b' = tr(E E')/2 I - E E', (b.b) R = cof(E) -/+ [b]x E), the four
 *     (R, +-t) candidates disambiguated by cheirality (closed-form ray depths) on the five sample points;
 *   - scoring: DLT triangulation of every correspondence (the mapper's `triangulate`, orc_tri.c, its eigenvector by
 *     inverse iteration instead of Jacobi sweeps: orc_sym4_min_eigvec_invit), inlier iff both
 *     depths > 0 and both reprojection errors < max_repr_error;
 *   - sample 5-tuples are SUPPLIED BY THE CALLER, all are evaluated; winner = most inliers, ties to the lower
 *     sample, then the lower root.
 * The reference has no test or golden vector for this: parity unpinned; tests pin this file against ground-truth
 * two-view scenes, the defining constraints and numpy.roots. */
#include "slam_oracle.h"
#include <math.h>
#include <string.h>

/* monomials of (x, y, z):  degree 1: x y z 1;  degree <= 2: x2 xy xz y2 yz z2 x y z 1;  degree <= 3 in Nister's
 * order: x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1.  T2/T3: index of the product monomial. */
static const int T2[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
static const int T3[10][4] = {{0, 2, 4, 5}, {2, 3, 8, 9}, {4, 8, 10, 11}, {3, 1, 6, 7}, {8, 6, 13, 14},
                              {10, 13, 16, 17}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};

static void mul11(const double *a, const double *b, double *o)       /* o(10) += a(4) b(4) */
{
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) o[T2[i][j]] += a[i] * b[j];
}
static void mul21(const double *a, const double *b, double *o)       /* o(20) += a(10) b(4) */
{
    for (int i = 0; i < 10; i++) for (int j = 0; j < 4; j++) o[T3[i][j]] += a[i] * b[j];
}
static void minor2(const double *a, const double *b, const double *c, const double *d, double *o)   /* o(10) = a b - c d */
{
    double t1[10] = {0}, t2[10] = {0};
    mul11(a, b, t1); mul11(c, d, t2);
    for (int i = 0; i < 10; i++) o[i] = t1[i] - t2[i];
}

/* one-variable polynomials, coefficients low -> high */
static void pmul(const double *a, int na, const double *b, int nb, double *o)   /* o(na+nb-1) = a b */
{
    for (int i = 0; i < na + nb - 1; i++) o[i] = 0.0;
    for (int i = 0; i < na; i++) for (int j = 0; j < nb; j++) o[i + j] += a[i] * b[j];
}
static double peval(const double *p, int deg, double x)
{
    double f = p[deg];
    for (int i = deg - 1; i >= 0; i--) f = f * x + p[i];
    return f;
}

/* root of p (degree deg) inside (lo, hi) where p(lo), p(hi) have opposite signs: safeguarded Newton */
static double bracket_root(const double *p, const double *dp, int deg, double lo, double hi, double flo)
{
    double x = 0.5 * (lo + hi);
    for (int it = 0; it < 200; it++) {
        const double f = peval(p, deg, x);
        if (f == 0.0) return x;
        if ((f < 0.0) == (flo < 0.0)) lo = x; else hi = x;
        const double df = peval(dp, deg - 1, x);
        double xn = x - f / df;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (xn == x || xn == lo || xn == hi) return xn;
        x = xn;
    }
    return x;
}

/* real roots (ascending) of p[0] + p[1] x + ... + p[deg] x^deg, deg <= 10: the roots of each derivative bracket
 * the roots of the one before it */
int orc_poly_real_roots(const double *p, int deg, double *roots)
{
    while (deg > 0 && p[deg] == 0.0) deg--;
    if (deg <= 0 || deg > 10) return 0;
    double D[11][11];                                   /* D[k] = k-th derivative, degree deg - k */
    for (int i = 0; i <= deg; i++) D[0][i] = p[i];
    for (int k = 1; k < deg; k++)
        for (int i = 0; i <= deg - k; i++) D[k][i] = (double)(i + 1) * D[k - 1][i + 1];
    double crit[11], next[11];
    int nc = 0;
    crit[nc++] = -D[deg - 1][0] / D[deg - 1][1];       /* the linear derivative */
    if (!isfinite(crit[0])) return 0;
    for (int k = deg - 2; k >= 0; k--) {
        const int d = deg - k;
        const double *q = D[k], *dq = D[k + 1];
        double bound = 0.0;
        for (int i = 0; i < d; i++) { const double c = fabs(q[i] / q[d]); if (c > bound) bound = c; }
        bound = bound + 1.0;
        if (!isfinite(bound)) return 0;
        int nn = 0;
        double lo = -bound, flo = peval(q, d, lo);
        for (int j = 0; j <= nc; j++) {
            const double hi = j < nc ? crit[j] : bound;
            const double fhi = peval(q, d, hi);
            if (hi > lo) {
                if (flo != 0.0 && fhi != 0.0 && (flo < 0.0) != (fhi < 0.0)) {
                    /* an end that is only the Cauchy bound can be astronomically far from the root: walk towards
                     * it from the finite end with doubling steps until the sign change is enclosed */
                    double a = lo, fa = flo, b = hi, root = 0.0;
                    int a_far = j == 0, b_far = j == nc, hit = 0;
                    if (a_far && b_far) {
                        const double f0 = peval(q, d, 0.0);
                        if (f0 == 0.0) hit = 1;
                        else if ((f0 < 0.0) == (fa < 0.0)) { a = 0.0; fa = f0; a_far = 0; }
                        else { b = 0.0; b_far = 0; }
                    }
                    if (!hit && a_far) {
                        double h = fabs(b) > 1.0 ? fabs(b) : 1.0;
                        for (int it = 0; it < 1100; it++) {
                            const double x = b - h;
                            if (!(x > a)) break;
                            const double fx = peval(q, d, x);
                            if (fx == 0.0) { hit = 1; root = x; break; }
                            if ((fx < 0.0) == (fa < 0.0)) { a = x; fa = fx; break; }
                            b = x; h = 2.0 * h;
                        }
                    } else if (!hit && b_far) {
                        double h = fabs(a) > 1.0 ? fabs(a) : 1.0;
                        for (int it = 0; it < 1100; it++) {
                            const double x = a + h;
                            if (!(x < b)) break;
                            const double fx = peval(q, d, x);
                            if (fx == 0.0) { hit = 1; root = x; break; }
                            if ((fx < 0.0) != (fa < 0.0)) { b = x; break; }
                            a = x; fa = fx; h = 2.0 * h;
                        }
                    }
                    next[nn++] = hit ? root : bracket_root(q, dq, d, a, b, fa);
                }
                else if (fhi == 0.0 && j < nc) next[nn++] = hi;                 /* a root that is also critical */
                lo = hi; flo = fhi;
            }
        }
        nc = nn;
        for (int j = 0; j < nn; j++) crit[j] = next[j];
    }
    for (int j = 0; j < nc; j++) roots[j] = crit[j];
    return nc;
}

/* null space of the 5x9 system (rows destroyed): Gauss-Jordan with complete pivoting; 4 basis vectors */
static int nullspace_5x9(double A[5][9], double N[4][9])
{
    int piv[5], used[9] = {0};
    for (int s = 0; s < 5; s++) {
        int pr = -1, pc = -1; double best = 0.0;
        for (int r = s; r < 5; r++)
            for (int c = 0; c < 9; c++)
                if (!used[c] && fabs(A[r][c]) > best) { best = fabs(A[r][c]); pr = r; pc = c; }
        if (pr < 0) return 0;
        if (pr != s) for (int c = 0; c < 9; c++) { const double t = A[s][c]; A[s][c] = A[pr][c]; A[pr][c] = t; }
        piv[s] = pc; used[pc] = 1;
        const double inv = 1.0 / A[s][pc];
        for (int c = 0; c < 9; c++) A[s][c] *= inv;
        A[s][pc] = 1.0;
        for (int r = 0; r < 5; r++) {
            if (r == s) continue;
            const double f = A[r][pc];
            if (f == 0.0) continue;
            for (int c = 0; c < 9; c++) A[r][c] -= f * A[s][c];
            A[r][pc] = 0.0;
        }
    }
    int k = 0;
    for (int f = 0; f < 9; f++) {
        if (used[f]) continue;
        for (int c = 0; c < 9; c++) N[k][c] = 0.0;
        N[k][f] = 1.0;
        for (int s = 0; s < 5; s++) N[k][piv[s]] = -A[s][f];
        k++;
    }
    for (int a = 0; a < 4; a++) {                       /* modified Gram-Schmidt: an orthonormal basis conditions the cubics */
        for (int b = 0; b < a; b++) {
            double d = 0.0;
            for (int c = 0; c < 9; c++) d += N[a][c] * N[b][c];
            for (int c = 0; c < 9; c++) N[a][c] -= d * N[b][c];
        }
        double nn = 0.0;
        for (int c = 0; c < 9; c++) nn += N[a][c] * N[a][c];
        if (!(nn > 0.0)) return 0;
        const double inv = 1.0 / sqrt(nn);
        for (int c = 0; c < 9; c++) N[a][c] *= inv;
    }
    return 1;
}

/* Minimal solver.  q1, q2: five normalised points each, (x, y) pairs, with  q2' E q1 = 0.  Es: up to 10 essential
 * matrices, row-major 3x3 each.  Returns their number. */
int orc_five_point_solve(const double q1[10], const double q2[10], double Es[90])
{
    double A[5][9], N[4][9];
    for (int i = 0; i < 5; i++) {
        const double x = q1[2 * i], y = q1[2 * i + 1], u = q2[2 * i], v = q2[2 * i + 1];
        A[i][0] = u * x; A[i][1] = u * y; A[i][2] = u; A[i][3] = v * x; A[i][4] = v * y; A[i][5] = v;
        A[i][6] = x; A[i][7] = y; A[i][8] = 1.0;
    }
    if (!nullspace_5x9(A, N)) return 0;
    double Ep[9][4];                                    /* E entries as polynomials in (x, y, z, 1) */
    for (int e = 0; e < 9; e++) for (int m = 0; m < 4; m++) Ep[e][m] = N[m][e];
    double M[10][20];
    {   /* E E' - tr/2 I, times E: nine cubics */
        double EEt[9][10], tr[10];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                double *o = EEt[3 * r + c];
                for (int i = 0; i < 10; i++) o[i] = 0.0;
                for (int k = 0; k < 3; k++) mul11(Ep[3 * r + k], Ep[3 * c + k], o);
            }
        for (int i = 0; i < 10; i++) tr[i] = 0.5 * ((EEt[0][i] + EEt[4][i]) + EEt[8][i]);
        for (int d = 0; d < 3; d++) for (int i = 0; i < 10; i++) EEt[4 * d][i] -= tr[i];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                double *o = M[3 * r + c];
                for (int i = 0; i < 20; i++) o[i] = 0.0;
                for (int k = 0; k < 3; k++) mul21(EEt[3 * r + k], Ep[3 * k + c], o);
            }
        /* det E */
        double m0[10], m1[10], m2[10], t[20];
        minor2(Ep[4], Ep[8], Ep[5], Ep[7], m0);
        minor2(Ep[3], Ep[8], Ep[5], Ep[6], m1);
        minor2(Ep[3], Ep[7], Ep[4], Ep[6], m2);
        double *o = M[9];
        for (int i = 0; i < 20; i++) o[i] = 0.0;
        mul21(m0, Ep[0], o);
        for (int i = 0; i < 20; i++) t[i] = 0.0;
        mul21(m1, Ep[1], t);
        for (int i = 0; i < 20; i++) o[i] -= t[i];
        for (int i = 0; i < 20; i++) t[i] = 0.0;
        mul21(m2, Ep[2], t);
        for (int i = 0; i < 20; i++) o[i] += t[i];
    }
    for (int s = 0; s < 10; s++) {                      /* Gauss-Jordan on the first ten columns, row pivoting */
        int pr = s; double best = fabs(M[s][s]);
        for (int r = s + 1; r < 10; r++) if (fabs(M[r][s]) > best) { best = fabs(M[r][s]); pr = r; }
        if (!(best > 0.0)) return 0;
        if (pr != s) for (int c = 0; c < 20; c++) { const double t = M[s][c]; M[s][c] = M[pr][c]; M[pr][c] = t; }
        const double inv = 1.0 / M[s][s];
        for (int c = 0; c < 20; c++) M[s][c] *= inv;
        M[s][s] = 1.0;
        for (int r = 0; r < 10; r++) {
            if (r == s) continue;
            const double f = M[r][s];
            if (f == 0.0) continue;
            for (int c = 0; c < 20; c++) M[r][c] -= f * M[s][c];
            M[r][s] = 0.0;
        }
    }
    double Ba[3][4], Bb[3][4], Bc[3][5];                /* B(z) rows: x-, y- and constant column, low -> high in z */
    for (int r = 0; r < 3; r++) {
        const double *e = M[4 + 2 * r], *f = M[5 + 2 * r];
        Ba[r][0] = e[12]; Ba[r][1] = e[11] - f[12]; Ba[r][2] = e[10] - f[11]; Ba[r][3] = -f[10];
        Bb[r][0] = e[15]; Bb[r][1] = e[14] - f[15]; Bb[r][2] = e[13] - f[14]; Bb[r][3] = -f[13];
        Bc[r][0] = e[19]; Bc[r][1] = e[18] - f[19]; Bc[r][2] = e[17] - f[18]; Bc[r][3] = e[16] - f[17]; Bc[r][4] = -f[16];
    }
    double P[11];
    {
        double t1[8], t2[8], u[8], w[11];
        pmul(Bb[1], 4, Bc[2], 5, t1); pmul(Bc[1], 5, Bb[2], 4, t2);
        for (int i = 0; i < 8; i++) u[i] = t1[i] - t2[i];
        pmul(Ba[0], 4, u, 8, P);
        pmul(Ba[1], 4, Bc[2], 5, t1); pmul(Bc[1], 5, Ba[2], 4, t2);
        for (int i = 0; i < 8; i++) u[i] = t1[i] - t2[i];
        pmul(Bb[0], 4, u, 8, w);
        for (int i = 0; i < 11; i++) P[i] -= w[i];
        pmul(Ba[1], 4, Bb[2], 4, t1); pmul(Bb[1], 4, Ba[2], 4, t2);
        for (int i = 0; i < 7; i++) u[i] = t1[i] - t2[i];
        pmul(Bc[0], 5, u, 7, w);
        for (int i = 0; i < 11; i++) P[i] += w[i];
    }
    for (int i = 0; i < 11; i++) if (!isfinite(P[i])) return 0;
    double zr[10];
    const int nr = orc_poly_real_roots(P, 10, zr);
    int ne = 0;
    for (int i = 0; i < nr; i++) {
        const double z = zr[i];
        double R[3][3];
        for (int r = 0; r < 3; r++) { R[r][0] = peval(Ba[r], 3, z); R[r][1] = peval(Bb[r], 3, z); R[r][2] = peval(Bc[r], 4, z); }
        double best = 0.0, nx = 0.0, ny = 0.0, nz = 0.0;
        for (int a = 0; a < 2; a++)
            for (int b = a + 1; b < 3; b++) {
                const double c0 = R[a][1] * R[b][2] - R[a][2] * R[b][1];
                const double c1 = R[a][2] * R[b][0] - R[a][0] * R[b][2];
                const double c2 = R[a][0] * R[b][1] - R[a][1] * R[b][0];
                if (fabs(c2) > best) { best = fabs(c2); nx = c0; ny = c1; nz = c2; }
            }
        if (!(best > 0.0)) continue;
        const double x = nx / nz, y = ny / nz;
        double *E = Es + 9 * ne;
        int fin = 1;
        for (int e = 0; e < 9; e++) {
            E[e] = ((x * N[0][e] + y * N[1][e]) + z * N[2][e]) + N[3][e];
            fin &= isfinite(E[e]) != 0;
        }
        if (fin) ne++;
    }
    return ne;
}

/* The four (R, t) of an essential matrix E (row-major), |t| = 1, each 12 doubles column-major 3x4.
 * Order: (Ra, +t), (Ra, -t), (Rb, +t), (Rb, -t).  Returns 4, or 0 if E is degenerate. */
int orc_essential_poses(const double E[9], double Rt[48])
{
    double G[9];                                        /* tr(E E')/2 I - E E' = b b' */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) G[3 * r + c] = (E[3 * r] * E[3 * c] + E[3 * r + 1] * E[3 * c + 1]) + E[3 * r + 2] * E[3 * c + 2];
    const double h = 0.5 * ((G[0] + G[4]) + G[8]);
    for (int i = 0; i < 9; i++) G[i] = -G[i];
    G[0] += h; G[4] += h; G[8] += h;
    int m = 0;
    if (G[4] > G[0]) m = 1;
    if (G[8] > G[4 * m]) m = 2;
    if (!(G[4 * m] > 0.0)) return 0;
    const double s = 1.0 / sqrt(G[4 * m]);
    const double b[3] = {G[3 * m] * s, G[3 * m + 1] * s, G[3 * m + 2] * s};
    const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    if (!(bb > 0.0) || !isfinite(bb)) return 0;
    double C[9], BE[9];
    for (int r = 0; r < 3; r++) {                       /* cofactor rows: E_{r+1} x E_{r+2} */
        const double *p = E + 3 * ((r + 1) % 3), *q = E + 3 * ((r + 2) % 3);
        C[3 * r] = p[1] * q[2] - p[2] * q[1]; C[3 * r + 1] = p[2] * q[0] - p[0] * q[2]; C[3 * r + 2] = p[0] * q[1] - p[1] * q[0];
    }
    for (int c = 0; c < 3; c++) {                       /* [b]x E */
        BE[c] = b[1] * E[6 + c] - b[2] * E[3 + c];
        BE[3 + c] = b[2] * E[c] - b[0] * E[6 + c];
        BE[6 + c] = b[0] * E[3 + c] - b[1] * E[c];
    }
    const double ib = 1.0 / bb, in = 1.0 / sqrt(bb);
    const double t[3] = {b[0] * in, b[1] * in, b[2] * in};
    for (int k = 0; k < 4; k++) {
        double *P = Rt + 12 * k;
        const double sg = (k & 1) ? -1.0 : 1.0;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++)
                P[r + 3 * c] = (k < 2 ? C[3 * r + c] - BE[3 * r + c] : C[3 * r + c] + BE[3 * r + c]) * ib;
            P[9 + r] = sg * t[r];
        }
    }
    return 4;
}

/* DLT triangulation of one correspondence under P1 = K1 [I | 0], P2 = K2 [R | t] (K: fx, fy, cx, cy; pixels
 * (x, y)); same A'A / smallest-eigenvector construction as orc_triangulate_point (inverse iteration).  X: point in camera-1
 * coordinates, Y: in camera-2 coordinates.  Returns 0 when the homogeneous scale vanishes. */
static int tri_two_view(const double *k1, const double *k2, const double *Rt, const double *a, const double *b, double *X, double *Y)
{
    double P1[12], P2[12];                              /* row-major 3x4 */
    for (int i = 0; i < 12; i++) P1[i] = 0.0;
    P1[0] = k1[0]; P1[2] = k1[2]; P1[5] = k1[1]; P1[6] = k1[3]; P1[10] = 1.0;
    for (int c = 0; c < 4; c++) {
        const double r0 = Rt[3 * c], r1 = Rt[3 * c + 1], r2 = Rt[3 * c + 2];
        P2[c] = k2[0] * r0 + k2[2] * r2; P2[4 + c] = k2[1] * r1 + k2[3] * r2; P2[8 + c] = r2;
    }
    double A[16], S[16], v[4];
    for (int j = 0; j < 4; j++) {
        A[j] = a[0] * P1[8 + j] - P1[j]; A[4 + j] = a[1] * P1[8 + j] - P1[4 + j];
        A[8 + j] = b[0] * P2[8 + j] - P2[j]; A[12 + j] = b[1] * P2[8 + j] - P2[4 + j];
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += A[4 * k + i] * A[4 * k + j];
            S[4 * i + j] = acc;
        }
    orc_sym4_min_eigvec_invit(S, v);
    const double iw = 1.0 / v[3];
    X[0] = v[0] * iw; X[1] = v[1] * iw; X[2] = v[2] * iw;
    for (int r = 0; r < 3; r++) Y[r] = ((Rt[r] * X[0] + Rt[3 + r] * X[1]) + Rt[6 + r] * X[2]) + Rt[9 + r];
    return isfinite(X[0]) && isfinite(X[1]) && isfinite(X[2]);
}

/* both reprojection errors of a correspondence; returns 0 (not an inlier candidate) when a depth is not positive */
static int two_view_errors(const double *k1, const double *k2, const double *Rt, const double *a, const double *b, double *e1, double *e2)
{
    double X[3], Y[3];
    if (!tri_two_view(k1, k2, Rt, a, b, X, Y)) return 0;
    if (!(X[2] > 0.0) || !(Y[2] > 0.0)) return 0;
    const double i1 = 1.0 / X[2], i2 = 1.0 / Y[2];
    const double dx1 = a[0] - (k1[0] * X[0] * i1 + k1[2]), dy1 = a[1] - (k1[1] * X[1] * i1 + k1[3]);
    const double dx2 = b[0] - (k2[0] * Y[0] * i2 + k2[2]), dy2 = b[1] - (k2[1] * Y[1] * i2 + k2[3]);
    *e1 = sqrt(dx1 * dx1 + dy1 * dy1); *e2 = sqrt(dx2 * dx2 + dy2 * dy2);
    return 1;
}

/* depths of a correspondence along its two rays under x2 = R x1 + t, normalised coordinates q = (x, y, 1):
 * lambda1 (q2 x R q1) = -(q2 x t)  (least squares over the three components), lambda2 = (lambda1 R q1 + t)_z */
static void ray_depths(const double *Rt, const double *q1, const double *q2, double *l1, double *l2)
{
    const double r0 = (Rt[0] * q1[0] + Rt[3] * q1[1]) + Rt[6], r1 = (Rt[1] * q1[0] + Rt[4] * q1[1]) + Rt[7],
                 r2 = (Rt[2] * q1[0] + Rt[5] * q1[1]) + Rt[8];
    const double a0 = q2[1] * r2 - r1, a1 = r0 - q2[0] * r2, a2 = q2[0] * r1 - q2[1] * r0;          /* q2 x R q1 */
    const double b0 = q2[1] * Rt[11] - Rt[10], b1 = Rt[9] - q2[0] * Rt[11], b2 = q2[0] * Rt[10] - q2[1] * Rt[9];   /* q2 x t */
    const double num = (a0 * b0 + a1 * b1) + a2 * b2, den = (a0 * a0 + a1 * a1) + a2 * a2;
    *l1 = -num / den;
    *l2 = *l1 * r2 + Rt[11];
}

/* pose of an essential matrix: the candidate with most of the five sample points in front of both cameras
 * (first maximum in the order of orc_essential_poses).  Returns 0 if E is degenerate. */
int orc_essential_pose_cheirality(const double E[9], const double q1[10], const double q2[10], double Rt[12])
{
    double C[48];
    if (!orc_essential_poses(E, C)) return 0;
    int best = -1, bk = 0;
    for (int k = 0; k < 4; k++) {
        int cnt = 0;
        for (int i = 0; i < 5; i++) {
            double l1, l2;
            ray_depths(C + 12 * k, q1 + 2 * i, q2 + 2 * i, &l1, &l2);
            cnt += (l1 > 0.0 && l2 > 0.0);
        }
        if (cnt > best) { best = cnt; bk = k; }
    }
    for (int j = 0; j < 12; j++) Rt[j] = C[12 * bk + j];
    return 1;
}

/* five_point_ransac (front_end.jl:305-308).  px1/px2: n x 2 pixels (x, y) of the previous key-frame / current
 * frame, pd1/pd2: n x 2 normalised coordinates, K1/K2 3x3 column-major, samples iters x 5 (0-based).
 * Outputs: E (3x3 column-major), P = [R | t] (3x4 column-major, previous -> current, |t| = 1), inliers (n bytes),
 * *error = sum over inliers of both reprojection errors, *best_iter.  Returns the winner's inlier count. */
int orc_five_point_ransac(const double *px1, const double *px2, const double *pd1, const double *pd2, int n,
                          const double *K1, const double *K2, double max_repr_error, const int32_t *samples, int iters,
                          double *E_out, double *P_out, unsigned char *inliers, double *error, int *best_iter)
{
    const double k1[4] = {K1[0], K1[4], K1[6], K1[7]}, k2[4] = {K2[0], K2[4], K2[6], K2[7]};
    int best = 0, bi = -1;
    double bestP[12] = {0}, bestE[9] = {0};
    for (int it = 0; it < iters; it++) {
        const int32_t *sm = samples + 5 * it;
        int ok = 1;
        for (int a = 0; a < 5; a++) {
            if (sm[a] < 0 || sm[a] >= n) ok = 0;
            for (int b = 0; b < a; b++) if (sm[a] == sm[b]) ok = 0;
        }
        if (!ok) continue;
        double q1[10], q2[10], Es[90];
        for (int a = 0; a < 5; a++) {
            q1[2 * a] = pd1[2 * sm[a]]; q1[2 * a + 1] = pd1[2 * sm[a] + 1];
            q2[2 * a] = pd2[2 * sm[a]]; q2[2 * a + 1] = pd2[2 * sm[a] + 1];
        }
        const int ne = orc_five_point_solve(q1, q2, Es);
        for (int e = 0; e < ne; e++) {
            double Rt[12];
            if (!orc_essential_pose_cheirality(Es + 9 * e, q1, q2, Rt)) continue;
            int cnt = 0;
            for (int i = 0; i < n; i++) {
                double e1, e2;
                if (two_view_errors(k1, k2, Rt, px1 + 2 * i, px2 + 2 * i, &e1, &e2)) cnt += (e1 < max_repr_error && e2 < max_repr_error);
            }
            if (cnt > best) {
                best = cnt; bi = it;
                for (int j = 0; j < 12; j++) bestP[j] = Rt[j];
                for (int j = 0; j < 9; j++) bestE[j] = Es[9 * e + j];
            }
        }
    }
    if (best_iter) *best_iter = bi;
    double esum = 0.0;
    for (int i = 0; i < n; i++) {
        double e1 = 0.0, e2 = 0.0;
        const int in = best > 0 && two_view_errors(k1, k2, bestP, px1 + 2 * i, px2 + 2 * i, &e1, &e2) && e1 < max_repr_error && e2 < max_repr_error;
        inliers[i] = (unsigned char)in;
        if (in) esum += e1 + e2;
    }
    if (error) *error = esum;
    for (int j = 0; j < 12; j++) P_out[j] = bestP[j];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) E_out[r + 3 * c] = bestE[3 * r + c];
    return best;
}
