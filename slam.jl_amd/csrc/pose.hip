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
    const int *cnt; int stride;     // keypoint-set layout instead (cnt != nullptr): problem z owns [z * stride, z * stride + cnt[z])
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

__device__ __forceinline__ int quartic_real_roots(const double *A, double *roots)
{
    if (!(fabs(A[4]) > 0.0)) return 0;
    const double a = A[3] / A[4], b = A[2] / A[4], c = A[1] / A[4], d = A[0] / A[4];
    if (!isfinite(a) || !isfinite(b) || !isfinite(c) || !isfinite(d)) return 0;
    const double a2 = a * a;
    const double p = b - 0.375 * a2;
    const double q = (c - 0.5 * a * b) + 0.125 * a2 * a;
    const double r = ((d - 0.25 * a * c) + 0.0625 * a2 * b) - 0.01171875 * a2 * a2;
    const double sh = 0.25 * a;
    // candidate roots in fixed slots with validity flags (a run-time counter would index the arrays dynamically and move them,
    // and everything derived from them, to scratch: 400 bytes per lane in k_p3p_score); the order of the valid ones is the
    // reference order y[n++]
    double y[4] = {0.0, 0.0, 0.0, 0.0};
    bool yv[4] = {false, false, false, false};
    const double z = cubic_root_nonneg(2.0 * p, p * p - 4.0 * r, -(q * q));
    if (z > 0.0) {
        const double s = sqrt(z), h = 0.5 * (p + z), g = q / (2.0 * s);
        const double d1 = z - 4.0 * (h - g), d2 = z - 4.0 * (h + g);
        if (d1 >= 0.0) { const double w = sqrt(d1); y[0] = 0.5 * (-s + w); y[1] = 0.5 * (-s - w); yv[0] = yv[1] = true; }
        if (d2 >= 0.0) { const double w = sqrt(d2); y[2] = 0.5 * (s + w); y[3] = 0.5 * (s - w); yv[2] = yv[3] = true; }
    } else {
        const double disc = p * p - 4.0 * r;
        if (disc >= 0.0) {
            const double w = sqrt(disc), t1 = 0.5 * (-p + w), t2 = 0.5 * (-p - w);
            if (t1 >= 0.0) { const double e = sqrt(t1); y[0] = e; y[1] = -e; yv[0] = yv[1] = true; }
            if (t2 >= 0.0) { const double e = sqrt(t2); y[2] = e; y[3] = -e; yv[2] = yv[3] = true; }
        }
    }
    int mask = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double x = y[i] - sh;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double f = (((A[4] * x + A[3]) * x + A[2]) * x + A[1]) * x + A[0];
            const double df = ((4.0 * A[4] * x + 3.0 * A[3]) * x + 2.0 * A[2]) * x + A[1];
            const double xn = x - f / df;
            if (isfinite(xn)) x = xn;
        }
        roots[i] = x;
        if (yv[i] && isfinite(x)) mask |= 1 << i;
    }
    return mask;                                                  // bit i: roots[i] is a real root (slots in reference order)
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

// X, F: 3 rows of 3.  Returns the number of poses; Pout receives pose number `want` (column-major 3 x 4) of the reference
// enumeration when it exists.  The candidate slots are visited with static indices (see quartic_real_roots).
typedef __attribute__((address_space(3))) double p3p_lds;
template <bool ALL>
__device__ __forceinline__ int p3p_solve(const double *X, const double *F, int want, double *Pout, p3p_lds *all)
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
    const int rmask = quartic_real_roots(Q, vr);
    double Ew[9];
    if (!tri_frame(X, X + 3, X + 6, Ew)) return 0;
    int ns = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (!((rmask >> i) & 1)) continue;
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
        if (ns == want) {
#pragma unroll
            for (int j = 0; j < 12; j++) Pout[j] = P[j];
        }
        if (ALL) {                                                // every pose, in enumeration order (an LDS array may be indexed at run time; LDS offset 0 is
                                                                  // a valid address, so the choice is a template flag, not a null test)
#pragma unroll
            for (int j = 0; j < 12; j++) all[12 * ns + j] = P[j];
        }
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

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_p3p_score(P3PArgs T)      // (128 instead of 134 registers, 8 spilled: four waves per SIMD, 0.94 -> 0.88 ms per 128-stream call)
{
    // one wave per triple: the (wave-uniform, latency-bound: ~20 k cycles of dependent f64 divisions and square roots) minimal
    // solver runs ONCE, its up to four poses go to LDS, then the 64 lanes stride over the map points and score every pose on
    // each point they load.  (Four waves per triple, each repeating the solver for "its" pose, took 190 us per 32-stream call.)
    __shared__ double s_P[48];
    const int it = blockIdx.x, z = blockIdx.y, lane = threadIdx.x;
    const int base = T.cnt ? z * T.stride : T.off[z], n = T.cnt ? T.cnt[z] : T.off[z + 1] - base;
    const double *pts = T.pts + 3 * (size_t)base, *px = T.px + 2 * (size_t)base, *pdn = T.pdn + 3 * (size_t)base;
    const int32_t *sm = T.samples + 3 * ((size_t)z * T.iters + it);
    const int i0 = sm[0], i1 = sm[1], i2 = sm[2];
    double K[9];
    for (int j = 0; j < 9; j++) K[j] = T.Ks[9 * z + j];
    int ns = 0;
    const bool valid = !(i0 < 0 || i1 < 0 || i2 < 0 || i0 >= n || i1 >= n || i2 >= n || i0 == i1 || i0 == i2 || i1 == i2);
    if (valid) {
        double X[9], F[9], Pd[12];
        for (int j = 0; j < 3; j++) {
            X[j] = pts[3 * i0 + j]; X[3 + j] = pts[3 * i1 + j]; X[6 + j] = pts[3 * i2 + j];
            F[j] = pdn[3 * i0 + j]; F[3 + j] = pdn[3 * i1 + j]; F[6 + j] = pdn[3 * i2 + j];
        }
        ns = p3p_solve<true>(X, F, -1, Pd, (p3p_lds *)s_P);             // all lanes hold the same values: the LDS stores coincide
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    // (Round 6 tried the incumbent bound of k_5pt_score here -- counts by ballots, a pose dropped once it cannot reach the best completely scored
    //  count of its stream: 0.9 -> 1.3 ms per 128-stream call.  Most triples of a rigid scene give a pose near the best one, so little is dropped,
    //  and four ballots + scalar adds per round cost more than the per-lane counters.  Not kept.)
    int cnt[4] = {0, 0, 0, 0};
    if (ns > 0) {
        double P[4][12];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int j = 0; j < 12; j++) P[k][j] = k < ns ? s_P[12 * k + j] : 0.0;
        for (int i = lane; i < n; i += 64) {
            const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
            const double q[2] = {px[2 * i], px[2 * i + 1]};
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < ns) {
                    const double e = p3p_reproj(P[k], K, X, q);
                    cnt[k] += (e >= 0.0 && e < T.thr) ? 1 : 0;
                }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            for (int o = 32; o > 0; o >>= 1) cnt[k] += __shfl_xor(cnt[k], o, 64);
    }
    if (lane < 4) {
        const int k = lane;
        const size_t e = ((size_t)z * T.iters + it) * 4 + k;
        T.counts[e] = k == 0 ? cnt[0] : k == 1 ? cnt[1] : k == 2 ? cnt[2] : cnt[3];
        if (k < ns)
            for (int j = 0; j < 12; j++) T.poses[e * 12 + j] = s_P[12 * k + j];
    }
}

__global__ __launch_bounds__(256) void k_p3p_select(P3PArgs T)
{
    __shared__ int s_cnt[256], s_idx[256];
    __shared__ double s_P[12], s_K[9];
    __shared__ double s_err[P3P_ERR_LDS];
    const int tid = threadIdx.x, z = blockIdx.x, ne = 4 * T.iters;
    const int base = T.cnt ? z * T.stride : T.off[z], n = T.cnt ? T.cnt[z] : T.off[z + 1] - base;
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
    T.iters = iters; T.thr = threshold; T.cnt = nullptr; T.stride = 0;
    T.counts = (int *)scr; T.poses = (double *)(scr + s_cnt); T.errs = (double *)(scr + s_cnt + s_pose);
    T.out = (double *)(d + o_out); T.inliers = (uint8_t *)(d + o_inl);
    { ProfScope span(ctx, "p3p_ransac");
      hipLaunchKernelGGL(k_p3p_score, dim3(iters, S), dim3(64), 0, ctx->stream, T);
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

// =====================================================================================================================
// compute_pose! on the device-resident keypoint set (src/front_end.jl:132-219): for every stream, the 3-D keypoints of the
// set -> P3P RANSAC (k_p3p_score / k_p3p_select above) -> removal of its outliers -> PnP refinement of the inliers
// (k_pnp_batch, ba_single.hip) -> removal of its outliers, acceptance tests of :136, :179-183, :207-211.  Nothing but the S poses,
// the S status words and the S list lengths travels to the host; the gather of :139-160 (ordered, is_3d keypoints only),
// the inlier compaction of :190-201 and the observation removals are kernels on the set's arrays.
// The reference draws its triples from Julia's global RNG inside RecoverPose; here they come from a counter-based generator
// (splitmix64 of seed, stream, iteration, attempt; three distinct indices) that keypoint_set.py restates, so that the same
// triples can be handed to the host seam (slam_p3p_ransac_batch) in the parity test.
// =====================================================================================================================
struct KPoseArgs {
    const double *yx, *xyz; const uint8_t *is3d; const int *count; int cap;   // the set
    const double *par;                 // S x 32: [16..19] fx fy cx cy, [20..23] k1 k2 p1 p2
    double *pts, *px, *pdn; int *slot; int *n3;                                // gathered 3-D keypoints, stride cap per stream
    int32_t *samples; int iters; unsigned long long seed;
    double *p3p_out; uint8_t *inl;     // k_p3p_select's outputs (S x 32 doubles; stride cap)
    double *bpx, *bpts; int *bslot; uint8_t *outl; PnPArgs *pnp; double *res;   // refinement inputs / outputs
    int iters_fast, iterations; double depth_eps, repr_eps;
    uint8_t *flags;                    // S x cap: observations to remove
    double *poses; int *status, *ninl; // S x 16 (column-major Tcw), S, S
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ordered compaction of a stream's flagged elements: returns this thread's output position (or -1), advances *base
__device__ __forceinline__ int ordered_slot(bool take, int *s_w, int *s_base)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long m = __ballot(take);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wv] = __popcll(m);
    __syncthreads();
    int off = *s_base;
    for (int w = 0; w < wv; w++) off += s_w[w];
    const int pos = take ? off + before : -1;
    __syncthreads();
    if (tid == 0) *s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return pos;
}

__global__ __launch_bounds__(256) void k_kpose_gather(KPoseArgs A)
{
    __shared__ int s_w[4], s_base;
    const int z = blockIdx.x, tid = threadIdx.x, n = A.count[z];
    const size_t b = (size_t)z * A.cap;
    const double fx = A.par[32 * z + 16], fy = A.par[32 * z + 17], cx = A.par[32 * z + 18], cy = A.par[32 * z + 19];
    const double k1 = A.par[32 * z + 20], k2 = A.par[32 * z + 21], p1 = A.par[32 * z + 22], p2 = A.par[32 * z + 23];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int j = c0 + tid;
        const bool take = j < n && A.is3d[b + j] != 0;
        const int pos = ordered_slot(take, s_w, &s_base);
        if (take) {
            const size_t q = b + j, o = b + pos;
            // undistort_point (camera.jl:98-125) -> undistorted_pixel (y, x); backproject (:138-140) -> position; normalize
            const double ny = (A.yx[2 * q] - cy) / fy, nx = (A.yx[2 * q + 1] - cx) / fx;
            const double s0 = ny * ny, s1 = nx * nx, r2 = s0 + s1;
            const double rd = (1.0 + k1 * r2) + k2 * (r2 * r2);
            const double pp = ny * nx;
            const double dtx = 2 * p1 * pp + p2 * (r2 + 2 * s0), dty = p1 * (r2 + 2 * s1) + 2 * p2 * pp;
            const double uy = (rd * ny + dty) * fy + cy, ux = (rd * nx + dtx) * fx + cx;
            const double bx = (ux - cx) / fx, by = (uy - cy) / fy;
            const double inv = 1.0 / sqrt((bx * bx + by * by) + 1.0);
            A.px[2 * o] = ux; A.px[2 * o + 1] = uy;                                   // (x, y), front_end.jl:151
            A.pdn[3 * o] = inv * bx; A.pdn[3 * o + 1] = inv * by; A.pdn[3 * o + 2] = inv * 1.0;
            A.pts[3 * o] = A.xyz[3 * q]; A.pts[3 * o + 1] = A.xyz[3 * q + 1]; A.pts[3 * o + 2] = A.xyz[3 * q + 2];
            A.slot[o] = j;
        }
    }
    if (tid == 0) A.n3[z] = s_base;
}

__global__ __launch_bounds__(256) void k_kpose_samples(KPoseArgs A)
{
    const int z = blockIdx.y, it = blockIdx.x * 256 + threadIdx.x;
    if (it >= A.iters) return;
    const int n = A.n3[z];
    int32_t *sm = A.samples + 3 * ((size_t)z * A.iters + it);
    if (n < 5) { sm[0] = sm[1] = sm[2] = -1; return; }              // front_end.jl:133-136: fewer than 5 3-D keypoints -> no P3P
    int idx[3]; unsigned att = 0;
    for (int k = 0; k < 3; k++) {
        for (;;) {
            const unsigned long long h = splitmix64(A.seed ^ ((unsigned long long)z << 48) ^ ((unsigned long long)it << 16) ^ (unsigned long long)att);
            att++;
            const int c = (int)(h % (unsigned long long)n);
            bool dup = false;
            for (int m = 0; m < k; m++) dup = dup || idx[m] == c;
            if (!dup) { idx[k] = c; break; }
        }
    }
    sm[0] = idx[0]; sm[1] = idx[1]; sm[2] = idx[2];
}

// RotZYX(R).theta1..3 (Rotations.jl; get_cw_ba, frame.jl:432-437) of a column-major 3 x 4 [R | t]
__device__ __forceinline__ void rt_to_x(const double *Rt, double *X)
{
    const double R11 = Rt[0], R21 = Rt[1], R31 = Rt[2], R12 = Rt[3], R22 = Rt[4], R13 = Rt[6], R23 = Rt[7];
    const double t1 = atan2(R21, R11), s1 = sin(t1), c1 = cos(t1);
    X[0] = t1; X[1] = atan2(-R31, sqrt(R11 * R11 + R21 * R21)); X[2] = atan2(R13 * s1 - R23 * c1, R22 * c1 - R12 * s1);
    X[3] = Rt[9]; X[4] = Rt[10]; X[5] = Rt[11];
}

__global__ __launch_bounds__(256) void k_kpose_prep(KPoseArgs A)
{
    __shared__ int s_w[4], s_base;
    const int z = blockIdx.x, tid = threadIdx.x, n = A.n3[z];
    const size_t b = (size_t)z * A.cap;
    const double *out = A.p3p_out + 32 * (size_t)z;
    const int best = ((const int *)(out + 25))[0];
    const bool ok = n >= 5 && best >= 5;                                     // :133, :179
    if (tid == 0) s_base = 0;
    __syncthreads();
    if (ok) {
        for (int c0 = 0; c0 < n; c0 += 256) {
            const int i = c0 + tid;
            const bool in = i < n && A.inl[b + i] != 0;
            if (i < n && !in) A.flags[b + A.slot[b + i]] = 1;                // :187-189 remove_obs_from_current_frame!
            const int pos = ordered_slot(in, s_w, &s_base);
            if (in) {
                const size_t o = b + pos, q = b + i;
                A.bpx[2 * o] = A.px[2 * q + 1]; A.bpx[2 * o + 1] = A.px[2 * q];      // back to (y, x), :195-199
                A.bpts[3 * o] = A.pts[3 * q]; A.bpts[3 * o + 1] = A.pts[3 * q + 1]; A.bpts[3 * o + 2] = A.pts[3 * q + 2];
                A.bslot[o] = A.slot[q];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        PnPArgs P;
        P.cam = {A.par[32 * z + 16], A.par[32 * z + 17], A.par[32 * z + 18], A.par[32 * z + 19]};
        P.px = A.bpx + 2 * b; P.pts = A.bpts + 3 * b; P.n = ok ? s_base : 0;
        P.iters_fast = A.iters_fast; P.iterations = A.iterations; P.depth_eps = A.depth_eps; P.repr_eps = A.repr_eps;
        P.outl = A.outl + b; P.result = A.res + 16 * (size_t)z;
        if (ok) rt_to_x(out + 12, P.X0);                                     // set_cw!(frame, iK * KP) = [R | t] (:184)
        else for (int a = 0; a < 6; a++) P.X0[a] = 0.0;
        A.pnp[z] = P;
        A.status[z] = ok ? 1 : 0; A.ninl[z] = ok ? best : 0;
    }
}

__global__ __launch_bounds__(256) void k_kpose_finish(KPoseArgs A)
{
    const int z = blockIdx.x, tid = threadIdx.x;
    const size_t b = (size_t)z * A.cap;
    const double *r = A.res + 16 * (size_t)z;
    const int n = A.pnp[z].n, no = (int)r[8];
    const bool p3p_ok = A.status[z] != 0;
    // :207-211: too few inliers after the refinement, or the refinement made the error worse -> reset (pose not applied, its
    // outliers not removed); r[9] = pnp_bundle_adjustment's own < 5 inliers exit
    const bool accept = p3p_ok && !(n - no < 5 || r[7] > r[6]) && r[9] == 0.0;
    if (accept)
        for (int i = tid; i < n; i += 256)
            if (A.outl[b + i]) A.flags[b + A.bslot[b + i]] = 1;              // :213-215
    if (tid == 0) {
        double *T = A.poses + 16 * (size_t)z;
        for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
        if (accept) {
            const double s1 = sin(r[0]), c1 = cos(r[0]), s2 = sin(r[1]), c2 = cos(r[1]), s3 = sin(r[2]), c3 = cos(r[2]);
            const double R[9] = {c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3, s1 * c2, s1 * s2 * s3 + c1 * c3, s1 * s2 * c3 - c1 * s3, -s2, c2 * s3, c2 * c3};
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i + 4 * j] = R[3 * i + j];
            T[12] = r[3]; T[13] = r[4]; T[14] = r[5];
        }
        A.status[z] = accept ? 1 : 0;
    }
}

extern "C" int slam_kpset_compute_pose(slam_ctx *ctx, slam_kpset *ks, const double *params, double threshold, int iters, uint64_t seed,
                                       int pnp_iters_fast, int pnp_iterations, double depth_eps, double repr_eps,
                                       double *poses_cw, int32_t *status, int32_t *n_inliers, int32_t *counts)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && params != nullptr && iters > 0 && poses_cw && status);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int S = ks->S, cap = ks->cap;
    const size_t nc = (size_t)S * cap;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // scratch layout
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += up(bytes); return at; };
    const size_t o_pts = take(nc * 24), o_px = take(nc * 16), o_pdn = take(nc * 24), o_slot = take(nc * 4), o_n3 = take((size_t)S * 4);
    const size_t o_smp = take((size_t)S * iters * 12), o_cnt = take((size_t)S * iters * 16), o_pose = take((size_t)S * iters * 4 * 96);
    const size_t o_err = take(nc * 8), o_out = take((size_t)S * 256), o_inl = take(nc);
    const size_t o_bpx = take(nc * 16), o_bpts = take(nc * 24), o_bslot = take(nc * 4), o_outl = take(nc), o_pnp = take((size_t)S * sizeof(PnPArgs));
    const size_t o_res = take((size_t)S * 128), o_flags = take(nc), o_T = take((size_t)S * 128), o_st = take((size_t)S * 4), o_ni = take((size_t)S * 4);
    char *scr;
    int rc = slam_scratch2(ctx, o, (void **)&scr);
    if (rc) return rc;
    const double *par_dev;
    rc = kpset_stage_params(ctx, ks, params, (size_t)S * 32, &par_dev);
    if (rc) return rc;
    KPoseArgs A;
    A.yx = ks->yx; A.xyz = ks->xyz; A.is3d = ks->is3d; A.count = ks->count; A.cap = cap; A.par = par_dev;
    A.pts = (double *)(scr + o_pts); A.px = (double *)(scr + o_px); A.pdn = (double *)(scr + o_pdn); A.slot = (int *)(scr + o_slot); A.n3 = (int *)(scr + o_n3);
    A.samples = (int32_t *)(scr + o_smp); A.iters = iters; A.seed = seed;
    A.p3p_out = (double *)(scr + o_out); A.inl = (uint8_t *)(scr + o_inl);
    A.bpx = (double *)(scr + o_bpx); A.bpts = (double *)(scr + o_bpts); A.bslot = (int *)(scr + o_bslot); A.outl = (uint8_t *)(scr + o_outl);
    A.pnp = (PnPArgs *)(scr + o_pnp); A.res = (double *)(scr + o_res);
    A.iters_fast = pnp_iters_fast; A.iterations = pnp_iterations; A.depth_eps = depth_eps; A.repr_eps = repr_eps;
    A.flags = (uint8_t *)(scr + o_flags); A.poses = (double *)(scr + o_T); A.status = (int *)(scr + o_st); A.ninl = (int *)(scr + o_ni);
    P3PArgs T;
    T.pts = A.pts; T.px = A.px; T.pdn = A.pdn; T.samples = A.samples; T.off = nullptr; T.cnt = A.n3; T.stride = cap;
    T.Ks = nullptr; T.iters = iters; T.thr = threshold;
    T.counts = (int *)(scr + o_cnt); T.poses = (double *)(scr + o_pose); T.errs = (double *)(scr + o_err);
    T.out = A.p3p_out; T.inliers = A.inl;
    // K per stream (column-major 3 x 3) goes with the parameters: built on the host, staged like them
    {
        double *Kh; void *hv;
        rc = slam_pinned(ctx, up((size_t)S * 72) + up((size_t)S * 128) + 3 * up((size_t)S * 4), &hv);
        if (rc) return rc;
        Kh = (double *)hv;
        for (int z = 0; z < S; z++) {
            double *K = Kh + 9 * z;
            for (int j = 0; j < 9; j++) K[j] = 0.0;
            K[0] = params[32 * z + 16]; K[4] = params[32 * z + 17]; K[6] = params[32 * z + 18]; K[7] = params[32 * z + 19]; K[8] = 1.0;
        }
        double *Kd;
        HIP_TRY(ctx, hipHostGetDevicePointer((void **)&Kd, Kh, 0));
        T.Ks = Kd;
    }
    HIP_TRY(ctx, hipMemsetAsync(A.flags, 0, nc, ctx->stream));
    { ProfScope span(ctx, "kpset_compute_pose");
      hipLaunchKernelGGL(k_kpose_gather, dim3(S), dim3(256), 0, ctx->stream, A);
      hipLaunchKernelGGL(k_kpose_samples, dim3((iters + 255) / 256, S), dim3(256), 0, ctx->stream, A);
      hipLaunchKernelGGL(k_p3p_score, dim3(iters, S), dim3(64), 0, ctx->stream, T);
      hipLaunchKernelGGL(k_p3p_select, dim3(S), dim3(256), 0, ctx->stream, T);
      hipLaunchKernelGGL(k_kpose_prep, dim3(S), dim3(256), 0, ctx->stream, A);
      rc = pnp_launch_device(ctx, S, A.pnp);
      if (rc) return rc;
      hipLaunchKernelGGL(k_kpose_finish, dim3(S), dim3(256), 0, ctx->stream, A); }
    HIP_TRY(ctx, hipGetLastError());
    rc = kpset_compact(ctx, ks, 1, A.flags);
    if (rc) return rc;
    // results: poses, status, inlier counts, list lengths -- one wait
    void *hv;
    rc = slam_pinned(ctx, up((size_t)S * 72) + up((size_t)S * 128) + 3 * up((size_t)S * 4), &hv);     // same block as above (K first)
    if (rc) return rc;
    char *h = (char *)hv + up((size_t)S * 72);
    HIP_TRY(ctx, hipMemcpyAsync(h, A.poses, (size_t)S * 128, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + up((size_t)S * 128), A.status, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + up((size_t)S * 128) + up((size_t)S * 4), A.ninl, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + up((size_t)S * 128) + 2 * up((size_t)S * 4), ks->count, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    memcpy(poses_cw, h, (size_t)S * 128);
    memcpy(status, h + up((size_t)S * 128), (size_t)S * 4);
    if (n_inliers) memcpy(n_inliers, h + up((size_t)S * 128) + up((size_t)S * 4), (size_t)S * 4);
    if (counts) memcpy(counts, h + up((size_t)S * 128) + 2 * up((size_t)S * 4), (size_t)S * 4);
    return SLAM_OK;
}
