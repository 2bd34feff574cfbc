// triangulate.hip -- two-view linear triangulation with the depth / reprojection gates of the mapper.
//
// Replaces the per-keypoint arithmetic of triangulate_stereo! and triangulate_temporal!
// (src/mapper.jl:142-183, 185-262): RecoverPose.triangulate (DLT; homogeneous point = eigenvector of A'A for
// its smallest eigenvalue), normalisation, both depth gates, both reprojection gates.  The map surgery around it
// (update_mappoint!, remove_stereo_keypoint!, remove_mappoint_obs!) stays on the host and is driven by `status`.
// One thread per keypoint: a few hundred flops each, the call is bound by its launch + the PCIe round trip of the
// keypoint lists (zero-copy mapped host block, like the tracking kernels).
#include "common.hpp"
#include "tri_device.hpp"
#include <cmath>

struct TriArgs {
    double P1[16], P2[16], T21[16];   // column-major 4x4 (Julia SMatrix)
    double cam1[4], cam2[4];          // fx, fy, cx, cy
    const double *px1, *px2;          // (y, x) pairs
    const double *parallax;           // nullptr: stereo semantics (every gate applies)
    double max_error, min_depth, min_parallax;
    int n;
    double *out;                      // n x 3
    uint8_t *status;
};

__global__ __launch_bounds__(64) void k_triangulate(TriArgs T)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= T.n) return;
    const double x1 = T.px1[2 * i + 1], y1 = T.px1[2 * i], x2 = T.px2[2 * i + 1], y2 = T.px2[2 * i];
    double A[16], S[16], v[4];
    for (int j = 0; j < 4; j++) {
        A[0 + j] = x1 * T.P1[2 + 4 * j] - T.P1[0 + 4 * j];
        A[4 + j] = y1 * T.P1[2 + 4 * j] - T.P1[1 + 4 * j];
        A[8 + j] = x2 * T.P2[2 + 4 * j] - T.P2[0 + 4 * j];
        A[12 + j] = y2 * T.P2[2 + 4 * j] - T.P2[1 + 4 * j];
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += A[4 * k + r] * A[4 * k + c];
            S[4 * r + c] = acc;
        }
    sym4_min_eigvec(S, v);
    const double iw = 1.0 / v[3];
    const double L0 = v[0] * iw, L1 = v[1] * iw, L2 = v[2] * iw, L3 = v[3] * iw;
    T.out[3 * i] = L0; T.out[3 * i + 1] = L1; T.out[3 * i + 2] = L2;
    const bool gated = T.parallax == nullptr || T.parallax[i] > T.min_parallax;
    bool ok = !(L2 < T.min_depth && gated);
    double R[3];
    for (int r = 0; r < 3; r++) R[r] = ((T.T21[r] * L0 + T.T21[r + 4] * L1) + T.T21[r + 8] * L2) + T.T21[r + 12] * L3;
    if (ok && R[2] < T.min_depth && gated) ok = false;
    if (ok) {
        const double iz = 1.0 / L2;
        const double py = T.cam1[1] * L1 * iz + T.cam1[3], px = T.cam1[0] * L0 * iz + T.cam1[2];
        const double dy = y1 - py, dx = x1 - px;
        if (sqrt(dy * dy + dx * dx) > T.max_error && gated) ok = false;
    }
    if (ok) {
        const double iz = 1.0 / R[2];
        const double py = T.cam2[1] * R[1] * iz + T.cam2[3], px = T.cam2[0] * R[0] * iz + T.cam2[2];
        const double dy = y2 - py, dx = x2 - px;
        if (sqrt(dy * dy + dx * dx) > T.max_error && gated) ok = false;
    }
    T.status[i] = ok ? 1 : 0;
}

extern "C" int slam_triangulate(slam_ctx *ctx, const double *P1, const double *P2, const double *T21,
                                const double *cam1, const double *cam2, const double *px1_yx, const double *px2_yx, int n,
                                double max_error, double min_depth, const double *parallax, double min_parallax,
                                double *out_xyz, uint8_t *status)
{
    ARG_TRY(ctx, ctx != nullptr && n >= 0);
    if (n == 0) return SLAM_OK;
    ARG_TRY(ctx, P1 && P2 && T21 && cam1 && cam2 && px1_yx && px2_yx && out_xyz && status);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pb = ((size_t)n * 16 + 255) & ~(size_t)255, qb = ((size_t)n * 8 + 255) & ~(size_t)255;
    const size_t ob = ((size_t)n * 24 + 255) & ~(size_t)255, sb = ((size_t)n + 255) & ~(size_t)255;
    char *h, *d;
    int rc = slam_pinned(ctx, 2 * pb + qb + ob + sb, (void **)&h);
    if (rc) return rc;
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&d, h, 0));
    memcpy(h, px1_yx, (size_t)n * 16); memcpy(h + pb, px2_yx, (size_t)n * 16);
    if (parallax) memcpy(h + 2 * pb, parallax, (size_t)n * 8);
    TriArgs T;
    memcpy(T.P1, P1, sizeof T.P1); memcpy(T.P2, P2, sizeof T.P2); memcpy(T.T21, T21, sizeof T.T21);
    memcpy(T.cam1, cam1, sizeof T.cam1); memcpy(T.cam2, cam2, sizeof T.cam2);
    T.px1 = (const double *)d; T.px2 = (const double *)(d + pb); T.parallax = parallax ? (const double *)(d + 2 * pb) : nullptr;
    T.max_error = max_error; T.min_depth = min_depth; T.min_parallax = min_parallax; T.n = n;
    T.out = (double *)(d + 2 * pb + qb); T.status = (uint8_t *)(d + 2 * pb + qb + ob);
    { ProfScope span(ctx, "triangulate");
      hipLaunchKernelGGL(k_triangulate, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, T); }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    memcpy(out_xyz, h + 2 * pb + qb, (size_t)n * 24);
    memcpy(status, h + 2 * pb + qb + ob, (size_t)n);
    return SLAM_OK;
}
