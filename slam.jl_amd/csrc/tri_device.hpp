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

