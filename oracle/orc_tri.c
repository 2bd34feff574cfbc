/* orc_tri.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned): two-view triangulation and the gating of
 * triangulate_stereo! / triangulate_temporal! (reference: src/mapper.jl:142-183, 185-262).
 *
 * The arithmetic of `triangulate(p1, p2, P1, P2, cache)` lives in the un-vendored dependency RecoverPose 0.1
 * (Project.toml:22,41; not under /root/reference).  Restated from its published algorithm: linear (DLT)
 * triangulation -- A = [x1 P1[3,:] - P1[1,:]; y1 P1[3,:] - P1[2,:]; x2 P2[3,:] - P2[1,:]; y2 P2[3,:] - P2[2,:]],
 * the homogeneous point is the eigenvector of A'A for its smallest eigenvalue (RecoverPose calls LAPACK geev on the
 * 4x4 through GEEV4x4Cache; here a cyclic Jacobi iteration, the same vector up to scale/sign, which the division
 * by its 4th component removes).  The reference has no test or golden vector for it: parity unpinned; the CPU
 * tests pin this file against numpy.linalg.eigh / svd and against ground-truth scenes. */
#include "slam_oracle.h"
#include <math.h>

/* eigenvector of the symmetric 4x4 `S` (row-major, destroyed) for its smallest eigenvalue: cyclic Jacobi */
void orc_sym4_min_eigvec(double S[16], double v[4])
{
    double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 32; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < 4; p++) { dg += S[5 * p] * S[5 * p]; for (int q = p + 1; q < 4; q++) off += S[4 * p + q] * S[4 * p + q]; }
        if (off <= 1e-60 * dg || off == 0.0) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                const double apq = S[4 * p + q];
                if (apq == 0.0) continue;
                const double theta = (S[5 * q] - S[5 * p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) {           /* S <- S J   (columns p, q) */
                    const double skp = S[4 * k + p], skq = S[4 * k + q];
                    S[4 * k + p] = c * skp - s * skq; S[4 * k + q] = s * skp + c * skq;
                }
                for (int k = 0; k < 4; k++) {           /* S <- J' S  (rows p, q) */
                    const double spk = S[4 * p + k], sqk = S[4 * q + k];
                    S[4 * p + k] = c * spk - s * sqk; S[4 * q + k] = s * spk + c * sqk;
                }
                for (int k = 0; k < 4; k++) {           /* V <- V J */
                    const double vkp = V[4 * k + p], vkq = V[4 * k + q];
                    V[4 * k + p] = c * vkp - s * vkq; V[4 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int p = 1; p < 4; p++) if (S[5 * p] < S[5 * m]) m = p;
    for (int k = 0; k < 4; k++) v[k] = V[4 * k + m];
}

/* The same eigenvector by inverse iteration on S + mu I (LDL' without pivoting, mu = 1e-13 tr S, six steps from
 * (1,1,1,1), max-norm scaling): ~20x fewer operations than the Jacobi sweeps and equal to them to ~(lambda_min /
 * lambda_2)^6 -- machine precision for a well-posed triangulation, looser only for rays that are nearly parallel,
 * where the point's depth is undetermined but its reprojection errors are not.  Used where a hypothesis has to be
 * scored against every correspondence (orc_5pt.c); S (row-major) is not modified. */
void orc_sym4_min_eigvec_invit(const double S[16], double v[4])
{
    const double mu = 1e-13 * (((S[0] + S[5]) + S[10]) + S[15]);
    const double s00 = S[0] + mu, s11 = S[5] + mu, s22 = S[10] + mu, s33 = S[15] + mu;
    double d0 = s00;
    if (d0 == 0.0) d0 = 1e-300;
    const double i0 = 1.0 / d0;
    const double l10 = S[4] * i0, l20 = S[8] * i0, l30 = S[12] * i0;
    double d1 = s11 - l10 * l10 * d0;
    if (d1 == 0.0) d1 = 1e-300;
    const double i1 = 1.0 / d1;
    const double l21 = (S[9] - l20 * l10 * d0) * i1, l31 = (S[13] - l30 * l10 * d0) * i1;
    double d2 = (s22 - l20 * l20 * d0) - l21 * l21 * d1;
    if (d2 == 0.0) d2 = 1e-300;
    const double i2 = 1.0 / d2;
    const double l32 = ((S[14] - l30 * l20 * d0) - l31 * l21 * d1) * i2;
    double d3 = ((s33 - l30 * l30 * d0) - l31 * l31 * d1) - l32 * l32 * d2;
    if (d3 == 0.0) d3 = 1e-300;
    const double i3 = 1.0 / d3;
    double x0 = 1.0, x1 = 1.0, x2 = 1.0, x3 = 1.0;
    for (int it = 0; it < 6; it++) {
        /* L y = x */
        const double y0 = x0, y1 = x1 - l10 * y0, y2 = (x2 - l20 * y0) - l21 * y1, y3 = ((x3 - l30 * y0) - l31 * y1) - l32 * y2;
        /* D z = y ; L' w = z */
        const double z0 = y0 * i0, z1 = y1 * i1, z2 = y2 * i2, z3 = y3 * i3;
        const double w3 = z3, w2 = z2 - l32 * w3, w1 = (z1 - l21 * w2) - l31 * w3, w0 = ((z0 - l10 * w1) - l20 * w2) - l30 * w3;
        double m = fabs(w0);
        if (fabs(w1) > m) m = fabs(w1);
        if (fabs(w2) > m) m = fabs(w2);
        if (fabs(w3) > m) m = fabs(w3);
        const double im = 1.0 / m;
        x0 = w0 * im; x1 = w1 * im; x2 = w2 * im; x3 = w3 * im;
    }
    v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3;
}

/* One point of triangulate_stereo! / triangulate_temporal!.  P1, P2, T21: 4x4 column-major (Julia SMatrix);
 * cam = (fx, fy, cx, cy); pixels (y, x).  Returns 1 if the map point is updated, 0 if the observation is removed. */
int orc_triangulate_point(const double *P1, const double *P2, const double *T21, const double *cam1, const double *cam2,
                          const double *px1_yx, const double *px2_yx, double max_error, double min_depth,
                          int gate_always, double parallax, double min_parallax, double *xyz)
{
    const double x1 = px1_yx[1], y1 = px1_yx[0], x2 = px2_yx[1], y2 = px2_yx[0];   /* [[2, 1]]: (x, y) */
    double A[16];                                                                   /* row-major 4x4 */
    for (int j = 0; j < 4; j++) {                                                   /* P[i,j] = P[(i-1) + 4 (j-1)] */
        A[0 + j] = x1 * P1[2 + 4 * j] - P1[0 + 4 * j];
        A[4 + j] = y1 * P1[2 + 4 * j] - P1[1 + 4 * j];
        A[8 + j] = x2 * P2[2 + 4 * j] - P2[0 + 4 * j];
        A[12 + j] = y2 * P2[2 + 4 * j] - P2[1 + 4 * j];
    }
    double S[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += A[4 * k + i] * A[4 * k + j];
            S[4 * i + j] = acc;
        }
    double v[4];
    orc_sym4_min_eigvec(S, v);
    const double iw = 1.0 / v[3];                                                   /* left_point *= 1.0 / left_point[4] */
    const double L[4] = {v[0] * iw, v[1] * iw, v[2] * iw, v[3] * iw};
    xyz[0] = L[0]; xyz[1] = L[1]; xyz[2] = L[2];
    const int gated = gate_always || parallax > min_parallax;                       /* temporal: `&& parallax > 20.0` */
    if (L[2] < min_depth && gated) return 0;
    double R[3];
    for (int i = 0; i < 3; i++) R[i] = ((T21[i] * L[0] + T21[i + 4] * L[1]) + T21[i + 8] * L[2]) + T21[i + 12] * L[3];
    if (R[2] < min_depth && gated) return 0;
    {   /* project(camera, left_point): (fy y / z + cy, fx x / z + cx), camera.jl:62-67 */
        const double iz = 1.0 / L[2];
        const double py = cam1[1] * L[1] * iz + cam1[3], px = cam1[0] * L[0] * iz + cam1[2];
        const double dy = px1_yx[0] - py, dx = px1_yx[1] - px;
        if (sqrt(dy * dy + dx * dx) > max_error && gated) return 0;
    }
    {
        const double iz = 1.0 / R[2];
        const double py = cam2[1] * R[1] * iz + cam2[3], px = cam2[0] * R[0] * iz + cam2[2];
        const double dy = px2_yx[0] - py, dx = px2_yx[1] - px;
        if (sqrt(dy * dy + dx * dx) > max_error && gated) return 0;
    }
    return 1;
}

/* parallax == NULL: stereo semantics (every gate applies); otherwise temporal semantics (a gate only removes the
 * observation when parallax[i] > min_parallax, mapper.jl:243-258). */
int orc_triangulate(const double *P1, const double *P2, const double *T21, const double *cam1, const double *cam2,
                    const double *px1_yx, const double *px2_yx, int n, double max_error, double min_depth,
                    const double *parallax, double min_parallax, double *out_xyz, unsigned char *status)
{
    int good = 0;
    for (int i = 0; i < n; i++) {
        status[i] = (unsigned char)orc_triangulate_point(P1, P2, T21, cam1, cam2, px1_yx + 2 * i, px2_yx + 2 * i, max_error, min_depth,
                                                         parallax == 0, parallax ? parallax[i] : 0.0, min_parallax, out_xyz + 3 * i);
        good += status[i];
    }
    return good;
}
