// tri_device.hpp -- device routines shared by the triangulation and the five-point scoring kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

// eigenvector of the symmetric 4x4 S (row-major) for its smallest eigenvalue: cyclic Jacobi, same operation
// order as the oracle (orc_sym4_min_eigvec)
__device__ static inline void sym4_min_eigvec(double *S, double *v)
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
                for (int k = 0; k < 4; k++) { const double a = S[4 * k + p], b = S[4 * k + q]; S[4 * k + p] = c * a - s * b; S[4 * k + q] = s * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = S[4 * p + k], b = S[4 * q + k]; S[4 * p + k] = c * a - s * b; S[4 * q + k] = s * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = V[4 * k + p], b = V[4 * k + q]; V[4 * k + p] = c * a - s * b; V[4 * k + q] = s * a + c * b; }
            }
    }
    int m = 0; double dm = S[0];                  // selects instead of V[4 k + m]: keeps V in registers
    for (int p = 1; p < 4; p++) if (S[5 * p] < dm) { dm = S[5 * p]; m = p; }
    for (int k = 0; k < 4; k++) v[k] = m == 0 ? V[4 * k] : (m == 1 ? V[4 * k + 1] : (m == 2 ? V[4 * k + 2] : V[4 * k + 3]));
}


// The same eigenvector by inverse iteration on S + mu I (see orc_sym4_min_eigvec_invit): used by the hypothesis
// scoring kernels, ~20x cheaper than the Jacobi sweeps.  S (row-major) is not modified.
__device__ static inline void sym4_min_eigvec_invit(const double *S, double *v)
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
        const double y0 = x0, y1 = x1 - l10 * y0, y2 = (x2 - l20 * y0) - l21 * y1, y3 = ((x3 - l30 * y0) - l31 * y1) - l32 * y2;
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
