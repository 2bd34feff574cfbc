// lk.hip -- pyramidal Lucas-Kanade with forward-backward check, one wavefront
// per keypoint.
//
// Replaces fb_tracking! / optflow! of the reference (src/tracker.jl:17-82,
// src/optical_flow/lucas_kanade.jl:9-100,140-212, src/optical_flow/utils.jl).
// The whole fb_tracking! call (all forward levels, survivor "compaction", the
// level-1 backward pass and the consistency test) is ONE launch: points are
// independent, so the reference's valid_ids bookkeeping (tracker.jl:30-48)
// reduces to per-point control flow.
//
// Mapping: a 64-lane wave owns a point; the (2w+1)^2 window is dealt to lanes
// in the reference's iteration order (q outer, p inner; element e -> lane
// e % 64, sequential per lane) and the two sums of prepare_linear_system are
// folded with a 6-step xor butterfly.  That summation order is restated in the
// CPU oracle (sum_order = 1) and is bit-reproducible; it differs from the
// reference's single-accumulator order only in rounding (<= 1e-12 px observed).
#include "common.hpp"
#include <cmath>

struct LKArgs {
    PyrView prev, cur;
    const double *pts, *disp0;
    int n, pyramid_levels, window, iterations;
    double eig_thr, eps, max_distance;
    double *out;
    uint8_t *status;
};

struct Offs { int up, down, left, right; };

__device__ __forceinline__ Offs get_offsets(int p0, int p1, double n0, double n1, int window, int H, int W)
{
    Offs o;
    const double q0 = (double)p0, q1 = (double)p1, w = (double)window;
    o.up = (int)floor(fmin(w, fmin(q0, n0) - 1));
    o.down = (int)floor(fmin(w, (double)H - fmax(q0, n0)));
    o.left = (int)floor(fmin(w, fmin(q1, n1) - 1));
    o.right = (int)floor(fmin(w, (double)W - fmax(q1, n1)));
    return o;
}

__device__ __forceinline__ double boxdiff(const double *I, int H, int y1, int y2, int x1, int x2)
{
    double sum = I[(size_t)(y2 - 1) + (size_t)(x2 - 1) * H];
    sum -= x1 > 1 ? I[(size_t)(y2 - 1) + (size_t)(x1 - 2) * H] : 0.0;
    sum -= y1 > 1 ? I[(size_t)(y1 - 2) + (size_t)(x2 - 1) * H] : 0.0;
    sum += (y1 > 1 && x1 > 1) ? I[(size_t)(y1 - 2) + (size_t)(x1 - 2) * H] : 0.0;
    return sum;
}

// compute_spatial_gradient + svd2x2 + pinv2x2 (utils.jl:5-45)
__device__ __forceinline__ double spatial_gradient(const LevelView &v, int p0, int p1, Offs o, double Gi[4])
{
    const int y1 = p0 - o.up, y2 = p0 + o.down, x1 = p1 - o.left, x2 = p1 + o.right;
    const double syy = boxdiff(v.Iyy, v.H, y1, y2, x1, x2);
    const double sxx = boxdiff(v.Ixx, v.H, y1, y2, x1, x2);
    const double syx = boxdiff(v.Iyx, v.H, y1, y2, x1, x2);
    // M col-major: M11 = syy, M21 = syx, M12 = syx, M22 = sxx
    const double E = (syy + sxx) / 2, F = (syy - sxx) / 2, G = (syx + syx) / 2, Hh = (syx - syx) / 2;
    const double Q = sqrt(E * E + Hh * Hh), R = sqrt(F * F + G * G);
    const double sx = Q + R, sy = Q - R;
    const double a1 = atan2(G, F), a2 = atan2(Hh, E);
    const double th = (a2 - a1) / 2, ph = (a2 + a1) / 2;
    const double s = (double)((sy > 0) - (sy < 0));
    const double sp = sin(ph), cp = cos(ph), st = sin(th), ct = cos(th);
    const double U0 = cp, U1 = sp, U2 = -s * sp, U3 = s * cp;
    const double S0 = sx, S1 = fabs(sy);
    const double V0 = ct, V1 = -st, V2 = st, V3 = ct;
    const double tol = 1.4901161193847656e-08;
    const double d1 = S0 > tol ? 1.0 / S0 : 0.0, d2 = S1 > tol ? 1.0 / S1 : 0.0;
    const double ud11 = U0 * d1 + U2 * 0.0, ud21 = U1 * d1 + U3 * 0.0;
    const double ud12 = U0 * 0.0 + U2 * d2, ud22 = U1 * 0.0 + U3 * d2;
    Gi[0] = ud11 * V0 + ud12 * V2;
    Gi[1] = ud21 * V0 + ud22 * V2;
    Gi[2] = ud11 * V1 + ud12 * V3;
    Gi[3] = ud21 * V1 + ud22 * V3;
    const double cnt = (double)((long)(y2 - y1 + 1) * (long)(x2 - x1 + 1));
    return fmin(S0, S1) / cnt;
}

__device__ __forceinline__ double bilinear(const double *img, int H, int W, double r, double c)
{
    int iy = (int)floor(r), ix = (int)floor(c);
    if (iy > H - 1) iy = H - 1;
    if (ix > W - 1) ix = W - 1;
    if (iy < 1) iy = 1;
    if (ix < 1) ix = 1;
    const double fy = r - iy, fx = c - ix;
    const double *p = img + (size_t)(iy - 1) + (size_t)(ix - 1) * H;
    const int dy = H > 1 ? 1 : 0; const size_t dx = W > 1 ? (size_t)H : 0;
    const double r0 = (1 - fx) * p[0] + fx * p[dx];
    const double r1 = (1 - fx) * p[dy] + fx * p[dy + dx];
    return (1 - fy) * r0 + fy * r1;
}

__device__ __forceinline__ bool lies_in(int H, int W, double a, double b)
{
    return 1.0 <= a && a <= (double)H && 1.0 <= b && b <= (double)W;
}

// One pyramid level of optflow! for one point (lucas_kanade.jl:33-96).  All
// control flow is wave-uniform.  Returns the point's status.
__device__ bool lk_level(const LevelView &first, const LevelView &second, int level,
                         double pty, double ptx, double &dy, double &dx,
                         int window, int iterations, double eig_thr, double eps)
{
    const int lane = threadIdx.x & 63;
    const int H = first.H, W = first.W;
    const double scale = (double)(1 << (level - 1));
    const int p0 = (int)floor(pty / scale), p1 = (int)floor(ptx / scale);
    const double pf0 = (double)p0, pf1 = (double)p1;
    Offs o = get_offsets(p0, p1, pf0, pf1, window, H, W);
    double Gi[4];
    double min_eig = spatial_gradient(first, p0, p1, o, Gi);
    if (min_eig < eig_thr) return false;
    double c0 = 0.0, c1 = 0.0;
    for (int it = 0; it < iterations; it++) {
        const double f0 = dy + c0, f1 = dx + c1;
        const double r0 = pf0 + f0, r1 = pf1 + f1;
        if (!lies_in(H, W, r0, r1)) return false;
        Offs no = get_offsets(p0, p1, r0, r1, window, H, W);
        if (no.up != o.up || no.down != o.down || no.left != o.left || no.right != o.right) {
            o = no;
            min_eig = spatial_gradient(first, p0, p1, o, Gi);
            if (min_eig < eig_thr) return false;
        }
        // prepare_linear_system (lucas_kanade.jl:159-173), wave order
        const int P = o.up + o.down + 1, Q = o.left + o.right + 1, NE = P * Q;
        double ay = 0.0, ax = 0.0;
        for (int e = lane; e < NE; e += 64) {
            const int p = e % P, q = e / P;
            const double r = r0 + (double)(p - o.up), c = r1 + (double)(q - o.left);
            const size_t a = (size_t)(p0 - o.up + p - 1) + (size_t)(p1 - o.left + q - 1) * H;
            const double dI = first.L[a] - bilinear(second.L, H, W, r, c);
            ay += dI * first.Iy[a];
            ax += dI * first.Ix[a];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double ty = __shfl_xor(ay, m), tx = __shfl_xor(ax, m);
            ay = ay + ty; ax = ax + tx;
        }
        const double fl0 = Gi[0] * ay + Gi[2] * ax, fl1 = Gi[1] * ay + Gi[3] * ax;
        if (fabs(fl0) < eps && fabs(fl1) < eps) break;
        c0 += fl0; c1 += fl1;
        if (!lies_in(H, W, r0 + fl0, r1 + fl1)) return false;
    }
    dy += c0; dx += c1;
    if (level > 1) { dy *= 2.0; dx *= 2.0; }
    return true;
}

__global__ __launch_bounds__(64) void k_fb_track(LKArgs A)
{
    const int i = blockIdx.x;
    const double py = A.pts[2 * i], px = A.pts[2 * i + 1];
    double dy = A.disp0 ? A.disp0[2 * i] : 0.0, dx = A.disp0 ? A.disp0[2 * i + 1] : 0.0;
    bool ok = true;
    for (int level = A.pyramid_levels + 1; level >= 1 && ok; level--)
        ok = lk_level(A.prev.lv[level - 1], A.cur.lv[level - 1], level, py, px, dy, dx,
                      A.window, A.iterations, A.eig_thr, A.eps);
    double ny = nan(""), nx = nan("");
    if (ok) {
        ny = py + dy; nx = px + dx;                       // tracker.jl:41-42
        double by = -dy * 1.0, bx = -dx * 1.0;            // back_displacement, scale = 1/2^0
        // backward: pyramid_levels = 0, default eps 1e-2 (tracker.jl:34,51-57)
        ok = lk_level(A.cur.lv[0], A.prev.lv[0], 1, ny, nx, by, bx, A.window, A.iterations, A.eig_thr, 1e-2);
        if (ok) {
            const double b0 = ny + by, b1 = nx + bx;
            const double d0 = py - b0, d1 = px - b1;
            if (sqrt(d0 * d0 + d1 * d1) >= A.max_distance) ok = false;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        A.out[2 * i] = ny; A.out[2 * i + 1] = nx;
        A.status[i] = ok ? 1 : 0;
    }
}

extern "C" int slam_fb_track(slam_ctx *ctx, const slam_pyr *prev, const slam_pyr *cur,
                             const double *pts_yx, const double *disp0_yx, int n,
                             int pyramid_levels, int window, int iterations,
                             double eig_thr, double eps, double max_distance,
                             double *out_yx, uint8_t *status)
{
    ARG_TRY(ctx, ctx != nullptr && prev != nullptr && cur != nullptr);
    ARG_TRY(ctx, n >= 0 && pyramid_levels >= 0 && window >= 0 && iterations >= 0);
    if (n == 0) return SLAM_OK;                                       // tracker.jl:24
    ARG_TRY(ctx, pts_yx != nullptr && out_yx != nullptr && status != nullptr);
    if (!(prev->levels > pyramid_levels && cur->levels > pyramid_levels))
        return slam_fail(ctx, SLAM_ERR_LAYERS, "Not enough layers in pyramids.");   // lucas_kanade.jl:12-15
    ARG_TRY(ctx, prev->H[0] == cur->H[0] && prev->W[0] == cur->W[0]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pb = (size_t)n * 16, sb = ((size_t)n + 255) & ~(size_t)255;
    char *s;
    int rc = slam_scratch(ctx, 3 * pb + sb, (void **)&s);
    if (rc) return rc;
    double *d_pts = (double *)s, *d_disp = (double *)(s + pb), *d_out = (double *)(s + 2 * pb);
    uint8_t *d_st = (uint8_t *)(s + 3 * pb);
    HIP_TRY(ctx, hipMemcpyAsync(d_pts, pts_yx, pb, hipMemcpyHostToDevice, ctx->stream));
    if (disp0_yx) HIP_TRY(ctx, hipMemcpyAsync(d_disp, disp0_yx, pb, hipMemcpyHostToDevice, ctx->stream));
    LKArgs A;
    A.prev = prev->view; A.cur = cur->view;
    A.pts = d_pts; A.disp0 = disp0_yx ? d_disp : nullptr; A.n = n;
    A.pyramid_levels = pyramid_levels; A.window = window; A.iterations = iterations;
    A.eig_thr = eig_thr; A.eps = eps; A.max_distance = max_distance;
    A.out = d_out; A.status = d_st;
    { ProfScope span(ctx, "fb_track");
      hipLaunchKernelGGL(k_fb_track, dim3(n), dim3(64), 0, ctx->stream, A); }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_yx, d_out, pb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(status, d_st, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}
