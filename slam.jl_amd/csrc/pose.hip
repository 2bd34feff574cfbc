// pose.hip -- P3P RANSAC of compute_pose! (src/front_end.jl:132-219; the call p3p_ransac(...) at :164-167).
//
// One 256-thread workgroup per caller-supplied sample triple: every wave runs the (wave-uniform) minimal solver --
// Grunert's quartic by polynomial arithmetic, Ferrari's factorisation with a safeguarded-Newton resolvent root,
// i.e. only + - * / sqrt, so the hypotheses are bit-identical to the CPU statement -- then wave s strides over
// the map points and counts the inliers of solution s.  A second single-workgroup kernel picks the winner (most inliers,
// ties to the lower iteration then solution), writes its inlier mask, the summed inlier error (index order) and
// K [R | t].  Inputs/outputs live in the context's mapped pinned block (a few tens of KB), scores in device scratch.
// The work is ~iters x 4 x n reprojections (256 x 4 x 1000 = 1 M): launch- and latency-bound, not HBM-bound.
#include "common.hpp"
#include <cmath>

#define P3P_ERR_LDS 4096                // map points whose errors the select kernel stages in LDS

struct P3PArgs {                    // S independent problems (S = 1: slam_p3p_ransac); problem z owns points [off[z], off[z+1])
    const double *pts, *px, *pdn;   // concatenated: n x 3, n x 2 (x, y), n x 3
    const int32_t *samples;         // S x iters x 3, 0-based, local to the problem
    const int *off;                 // S + 1
    const double *Ks;               // S x 9, column-major 3x3
    int iters;
    double thr;
    int *counts;                    // S x iters x 4
    double *poses;                  // S x iters x 4 x 12
    double *errs;                   // off[S] (only read back when a problem has more than P3P_ERR_LDS points)
    double *out;                    // S x 32 (mapped host): KP 12 | Rt 12 | error | {n_inliers, best_iter} as two ints
    uint8_t *inliers;               // off[S] (mapped host)
};

__device__ static double cubic_root_nonneg(double B, double C, double D)
{
    double hi = fabs(B);
    if (fabs(C) > hi) hi = fabs(C);
    if (fabs(D) > hi) hi = fabs(D);
    hi = hi + 1.0;
    double lo = 0.0, x = hi;
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

__device__ static int quartic_real_roots(const double *A, double *roots)
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
    } else {
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
        for (int k = 0; k < 4; k++) {
            const double f = (((A[4] * x + A[3]) * x + A[2]) * x + A[1]) * x + A[0];
            const double df = ((4.0 * A[4] * x + 3.0 * A[3]) * x + 2.0 * A[2]) * x + A[1];
            const double xn = x - f / df;
            if (isfinite(xn)) x = xn;
        }
        if (isfinite(x)) roots[m++] = x;
    }
    return m;
}

__device__ static inline void v3_sub(const double *a, const double *b, double *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
__device__ static inline double v3_dot(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ static inline void v3_cross(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ static inline bool v3_unit(const double *a, double *o)
{
    const double n = sqrt(v3_dot(a, a));
    if (!(n > 0.0)) return false;
    const double i = 1.0 / n;
    o[0] = a[0] * i; o[1] = a[1] * i; o[2] = a[2] * i;
    return true;
}
__device__ static bool tri_frame(const double *p1, const double *p2, const double *p3, double *E)
{
    double d12[3], d13[3], w[3];
    v3_sub(p2, p1, d12); v3_sub(p3, p1, d13);
    if (!v3_unit(d12, E)) return false;
    v3_cross(E, d13, w);
    if (!v3_unit(w, E + 6)) return false;
    v3_cross(E + 6, E, E + 3);
    return true;
}

// X, F: 3 rows of 3; Rt: up to 4 poses x 12 (column-major 3x4) in LDS
__device__ static int p3p_solve(const double *X, const double *F, double *Rt)
{
    double f1[3], f2[3], f3[3];
    if (!v3_unit(F, f1) || !v3_unit(F + 3, f2) || !v3_unit(F + 6, f3)) return 0;
    double t[3];
    v3_sub(X + 3, X + 6, t); const double a2 = v3_dot(t, t);
    v3_sub(X, X + 6, t);     const double b2 = v3_dot(t, t);
    v3_sub(X, X + 3, t);     const double c2 = v3_dot(t, t);
    if (!(a2 > 0.0 && b2 > 0.0 && c2 > 0.0)) return 0;
    const double ca = v3_dot(f2, f3), cb = v3_dot(f1, f3), cg = v3_dot(f1, f2);
    const double k = (a2 - c2) / b2, m = c2 / b2;
    const double N0 = 1.0 + k, N1 = -2.0 * k * cb, N2 = k - 1.0;
    const double D0 = 2.0 * cg, D1 = -2.0 * ca;
    const double DD0 = D0 * D0, DD1 = 2.0 * D0 * D1, DD2 = D1 * D1;
    const double NN0 = N0 * N0, NN1 = 2.0 * N0 * N1, NN2 = 2.0 * N0 * N2 + N1 * N1, NN3 = 2.0 * N1 * N2, NN4 = N2 * N2;
    const double ND0 = N0 * D0, ND1 = N0 * D1 + N1 * D0, ND2 = N1 * D1 + N2 * D0, ND3 = N2 * D1;
    const double W0 = 1.0, W1 = -2.0 * cb, W2 = 1.0;
    const double WD0 = W0 * DD0, WD1 = W0 * DD1 + W1 * DD0, WD2 = (W0 * DD2 + W1 * DD1) + W2 * DD0,
                 WD3 = W1 * DD2 + W2 * DD1, WD4 = W2 * DD2;
    double Q[5];
    Q[0] = ((DD0 + NN0) - 2.0 * cg * ND0) - m * WD0;
    Q[1] = ((DD1 + NN1) - 2.0 * cg * ND1) - m * WD1;
    Q[2] = ((DD2 + NN2) - 2.0 * cg * ND2) - m * WD2;
    Q[3] = ((0.0 + NN3) - 2.0 * cg * ND3) - m * WD3;
    Q[4] = ((0.0 + NN4) - 2.0 * cg * 0.0) - m * WD4;
    double vr[4];
    const int nr = quartic_real_roots(Q, vr);
    double Ew[9];
    if (!tri_frame(X, X + 3, X + 6, Ew)) return 0;
    int ns = 0;
    for (int i = 0; i < nr; i++) {
        const double v = vr[i];
        if (!(v > 0.0)) continue;
        const double den = D1 * v + D0;
        const double u = ((N2 * v + N1) * v + N0) / den;
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
        double P[12];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                P[r + 3 * c] = (Ec[r] * Ew[c] + Ec[3 + r] * Ew[3 + c]) + Ec[6 + r] * Ew[6 + c];
        for (int r = 0; r < 3; r++)
            P[9 + r] = Y1[r] - ((P[r] * X[0] + P[r + 3] * X[1]) + P[r + 6] * X[2]);
        bool fin = true;
        for (int j = 0; j < 12; j++) fin = fin && isfinite(P[j]);
        if (!fin) continue;
        for (int j = 0; j < 12; j++) Rt[12 * ns + j] = P[j];
        ns++;
    }
    return ns;
}

__device__ static inline double p3p_reproj(const double *P, const double *K, const double *X, const double *px)
{
    const double xc = ((P[0] * X[0] + P[3] * X[1]) + P[6] * X[2]) + P[9];
    const double yc = ((P[1] * X[0] + P[4] * X[1]) + P[7] * X[2]) + P[10];
    const double zc = ((P[2] * X[0] + P[5] * X[1]) + P[8] * X[2]) + P[11];
    if (!(zc > 0.0)) return -1.0;
    const double iz = 1.0 / zc;
    const double dx = px[0] - (K[0] * xc * iz + K[6]), dy = px[1] - (K[4] * yc * iz + K[7]);
    return sqrt(dx * dx + dy * dy);
}

__global__ __launch_bounds__(256) void k_p3p_score(P3PArgs T)
{
    // four waves per triple: each runs the (wave-uniform) solver, wave s then scores pose s with its 64 lanes
    const int it = blockIdx.x, z = blockIdx.y, lane = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int base = T.off[z], n = T.off[z + 1] - base;
    const double *pts = T.pts + 3 * (size_t)base, *px = T.px + 2 * (size_t)base, *pdn = T.pdn + 3 * (size_t)base;
    const int32_t *sm = T.samples + 3 * ((size_t)z * T.iters + it);
    const int i0 = sm[0], i1 = sm[1], i2 = sm[2];
    double K[9];
    for (int j = 0; j < 9; j++) K[j] = T.Ks[9 * z + j];
    int ns = 0;
    double P[12];
    const bool valid = !(i0 < 0 || i1 < 0 || i2 < 0 || i0 >= n || i1 >= n || i2 >= n || i0 == i1 || i0 == i2 || i1 == i2);
    if (valid) {
        double X[9], F[9];
        for (int j = 0; j < 3; j++) {
            X[j] = pts[3 * i0 + j]; X[3 + j] = pts[3 * i1 + j]; X[6 + j] = pts[3 * i2 + j];
            F[j] = pdn[3 * i0 + j]; F[3 + j] = pdn[3 * i1 + j]; F[6 + j] = pdn[3 * i2 + j];
        }
        double Rt[48];
        ns = p3p_solve(X, F, Rt);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k == s) for (int j = 0; j < 12; j++) P[j] = Rt[12 * k + j];
    }
    int cnt = 0;
    if (s < ns) {
        for (int i = lane; i < n; i += 64) {
            const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
            const double q[2] = {px[2 * i], px[2 * i + 1]};
            const double e = p3p_reproj(P, K, X, q);
            cnt += (e >= 0.0 && e < T.thr) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    }
    if (lane == 0) {
        const size_t e = ((size_t)z * T.iters + it) * 4 + s;
        T.counts[e] = cnt;
        if (s < ns)
            for (int j = 0; j < 12; j++) T.poses[e * 12 + j] = P[j];
    }
}

__global__ __launch_bounds__(256) void k_p3p_select(P3PArgs T)
{
    __shared__ int s_cnt[256], s_idx[256];
    __shared__ double s_P[12], s_K[9];
    __shared__ double s_err[P3P_ERR_LDS];
    const int tid = threadIdx.x, z = blockIdx.x, ne = 4 * T.iters;
    const int base = T.off[z], n = T.off[z + 1] - base;
    const double *pts = T.pts + 3 * (size_t)base, *px = T.px + 2 * (size_t)base;
    const int *counts = T.counts + (size_t)z * ne;
    const double *poses = T.poses + (size_t)z * ne * 12;
    double *errs = T.errs + base, *out = T.out + 32 * (size_t)z;
    uint8_t *inliers = T.inliers + base;
    const bool in_lds = n <= P3P_ERR_LDS;
    if (tid < 9) s_K[tid] = T.Ks[9 * z + tid];
    int bc = 0, bi = -1;
    for (int e = tid; e < ne; e += 256) {
        const int c = counts[e];
        if (c > bc) { bc = c; bi = e; }        // ascending e: the first maximum is kept
    }
    s_cnt[tid] = bc; s_idx[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const int c2 = s_cnt[tid + o], i2 = s_idx[tid + o];
            if (c2 > s_cnt[tid] || (c2 == s_cnt[tid] && c2 > 0 && i2 < s_idx[tid])) { s_cnt[tid] = c2; s_idx[tid] = i2; }
        }
        __syncthreads();
    }
    const int best = s_cnt[0], be = s_idx[0];
    if (tid < 12) s_P[tid] = best > 0 ? poses[(size_t)be * 12 + tid] : 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        double e = -1.0;
        if (best > 0) {
            const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
            const double q[2] = {px[2 * i], px[2 * i + 1]};
            e = p3p_reproj(s_P, s_K, X, q);
        }
        const bool in = best > 0 && e >= 0.0 && e < T.thr;
        inliers[i] = in ? 1 : 0;
        if (in_lds) s_err[i] = in ? e : 0.0; else errs[i] = in ? e : 0.0;   // + 0.0 leaves the sum unchanged
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        double esum = 0.0;
        if (in_lds) {
#pragma unroll 16
            for (int i = 0; i < n; i++) esum += s_err[i];             // index order; the reads pipeline, the adds are the chain
        } else {
#pragma unroll 16
            for (int i = 0; i < n; i++) esum += errs[i];
        }
        out[24] = esum;
        int *oi = (int *)(out + 25);
        oi[0] = best; oi[1] = best > 0 ? be / 4 : -1;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) {
                out[r + 3 * c] = (s_K[r] * s_P[3 * c] + s_K[r + 3] * s_P[3 * c + 1]) + s_K[r + 6] * s_P[3 * c + 2];
                out[12 + r + 3 * c] = s_P[r + 3 * c];
            }
    }
}

// S problems in one pair of launches (grid.y / grid.x = problem); offsets, intrinsics, inputs and outputs go through
// the context's mapped pinned block
static int p3p_run(slam_ctx *ctx, int S, const int32_t *off, const double *pts3d, const double *px_xy, const double *pdn,
                   const double *K, double threshold, const int32_t *samples, int iters,
                   double *KP, double *Rt, uint8_t *inliers, int *n_inliers, double *error, int *best_iter)
{
    const int ntot = off[S];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_off = 0, o_K = o_off + up((size_t)(S + 1) * 4), o_pts = o_K + up((size_t)S * 72);
    const size_t o_px = o_pts + up((size_t)ntot * 24), o_pdn = o_px + up((size_t)ntot * 16), o_smp = o_pdn + up((size_t)ntot * 24);
    const size_t o_out = o_smp + up((size_t)S * iters * 12), o_inl = o_out + (size_t)S * 256, total = o_inl + up((size_t)ntot);
    char *h, *d;
    int rc = slam_pinned(ctx, total, (void **)&h);
    if (rc) return rc;
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&d, h, 0));
    memcpy(h + o_off, off, (size_t)(S + 1) * 4); memcpy(h + o_K, K, (size_t)S * 72);
    memcpy(h + o_pts, pts3d, (size_t)ntot * 24); memcpy(h + o_px, px_xy, (size_t)ntot * 16); memcpy(h + o_pdn, pdn, (size_t)ntot * 24);
    memcpy(h + o_smp, samples, (size_t)S * iters * 12);
    const size_t s_cnt = up((size_t)S * iters * 16), s_pose = up((size_t)S * iters * 4 * 96), s_err = up((size_t)ntot * 8);
    char *scr;
    rc = slam_scratch(ctx, s_cnt + s_pose + s_err, (void **)&scr);
    if (rc) return rc;
    P3PArgs T;
    T.pts = (const double *)(d + o_pts); T.px = (const double *)(d + o_px); T.pdn = (const double *)(d + o_pdn);
    T.samples = (const int32_t *)(d + o_smp); T.off = (const int *)(d + o_off); T.Ks = (const double *)(d + o_K);
    T.iters = iters; T.thr = threshold;
    T.counts = (int *)scr; T.poses = (double *)(scr + s_cnt); T.errs = (double *)(scr + s_cnt + s_pose);
    T.out = (double *)(d + o_out); T.inliers = (uint8_t *)(d + o_inl);
    { ProfScope span(ctx, "p3p_ransac");
      hipLaunchKernelGGL(k_p3p_score, dim3(iters, S), dim3(256), 0, ctx->stream, T);
      hipLaunchKernelGGL(k_p3p_select, dim3(S), dim3(256), 0, ctx->stream, T); }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    for (int z = 0; z < S; z++) {
        const char *o = h + o_out + (size_t)z * 256;
        memcpy(KP + 12 * z, o, 96);
        if (Rt) memcpy(Rt + 12 * z, o + 96, 96);
        if (error) memcpy(error + z, o + 192, 8);
        memcpy(n_inliers + z, o + 200, 4);
        if (best_iter) memcpy(best_iter + z, o + 204, 4);
    }
    memcpy(inliers, h + o_inl, (size_t)ntot);
    return SLAM_OK;
}

extern "C" int slam_p3p_ransac(slam_ctx *ctx, const double *pts3d, const double *px_xy, const double *pdn, int n,
                               const double *K, double threshold, const int32_t *samples, int iters,
                               double *KP, double *Rt, uint8_t *inliers, int *n_inliers, double *error, int *best_iter)
{
    ARG_TRY(ctx, ctx != nullptr && n >= 0 && iters >= 0);
    ARG_TRY(ctx, K && KP && n_inliers);
    ARG_TRY(ctx, n == 0 || (pts3d && px_xy && pdn && inliers));
    ARG_TRY(ctx, iters == 0 || samples);
    if (n < 3 || iters == 0) {                 // nothing to sample from: "p3p_ransac returned nothing"
        *n_inliers = 0;
        for (int j = 0; j < 12; j++) { KP[j] = 0.0; if (Rt) Rt[j] = 0.0; }
        for (int i = 0; i < n; i++) inliers[i] = 0;
        if (error) *error = 0.0;
        if (best_iter) *best_iter = -1;
        return SLAM_OK;
    }
    const int32_t off[2] = {0, n};
    return p3p_run(ctx, 1, off, pts3d, px_xy, pdn, K, threshold, samples, iters, KP, Rt, inliers, n_inliers, error, best_iter);
}

extern "C" int slam_p3p_ransac_batch(slam_ctx *ctx, int S, const int32_t *offsets, const double *pts3d, const double *px_xy,
                                     const double *pdn, const double *K, double threshold, const int32_t *samples, int iters,
                                     double *KP, double *Rt, uint8_t *inliers, int *n_inliers, double *error, int *best_iter)
{
    ARG_TRY(ctx, ctx != nullptr && S >= 0 && iters >= 0);
    if (S == 0) return SLAM_OK;
    ARG_TRY(ctx, offsets && K && KP && n_inliers && offsets[0] == 0);
    for (int z = 0; z < S; z++) ARG_TRY(ctx, offsets[z + 1] >= offsets[z]);
    const int ntot = offsets[S];
    ARG_TRY(ctx, ntot == 0 || (pts3d && px_xy && pdn && inliers));
    ARG_TRY(ctx, iters == 0 || samples);
    if (ntot == 0 || iters == 0) {
        for (int z = 0; z < S; z++) {
            n_inliers[z] = 0;
            for (int j = 0; j < 12; j++) { KP[12 * z + j] = 0.0; if (Rt) Rt[12 * z + j] = 0.0; }
            if (error) error[z] = 0.0;
            if (best_iter) best_iter[z] = -1;
        }
        for (int i = 0; i < ntot; i++) inliers[i] = 0;
        return SLAM_OK;
    }
    return p3p_run(ctx, S, offsets, pts3d, px_xy, pdn, K, threshold, samples, iters, KP, Rt, inliers, n_inliers, error, best_iter);
}
