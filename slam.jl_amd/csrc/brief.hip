// brief.hip -- BRIEF-256 descriptors (describe -> ImageFeatures.create_descriptor,
// src/extractor.jl:103-105; off by default in the reference: params.jl:69).
//
// Two full-frame separable FIR passes (Gaussian sigma = sqrt(2), `window` taps,
// replicate border) and one wave per keypoint: each lane evaluates the
// intensity-pair tests lane, lane+64, ... and a wave ballot packs 64 test
// results into one descriptor word.
#include "common.hpp"
#include <cmath>

#define BRIEF_MAXTAPS 41
struct Taps { double w[BRIEF_MAXTAPS]; int n; };

__global__ __launch_bounds__(256) void k_fir_y(double *dst, const double *src, int H, int W, Taps t)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int y = (int)(i % H), x = (int)(i / H), hw = t.n >> 1;
    double acc = 0.0;
    for (int j = 0; j < t.n; j++) { int yy = y + j - hw; yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); acc += src[(size_t)yy + (size_t)x * H] * t.w[j]; }
    dst[i] = acc;
}
__global__ __launch_bounds__(256) void k_fir_x(double *dst, const double *src, int H, int W, Taps t)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int y = (int)(i % H), x = (int)(i / H), hw = t.n >> 1;
    double acc = 0.0;
    for (int j = 0; j < t.n; j++) { int xx = x + j - hw; xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx); acc += src[(size_t)y + (size_t)xx * H] * t.w[j]; }
    dst[i] = acc;
}

__global__ __launch_bounds__(64) void k_brief(const double *sm, int H, const int64_t *rc, const int32_t *pattern, int n_bits, uint64_t *out)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    const long y = rc[2 * k], x = rc[2 * k + 1];
    const int words = n_bits >> 6;
    for (int w = 0; w < words; w++) {
        const int32_t *p = pattern + 4 * (w * 64 + lane);
        const double v1 = sm[(size_t)(y - 1 + p[0]) + (size_t)(x - 1 + p[1]) * H];
        const double v2 = sm[(size_t)(y - 1 + p[2]) + (size_t)(x - 1 + p[3]) * H];
        const unsigned long long m = __ballot(v1 < v2);
        if (lane == 0) out[(size_t)k * words + w] = m;
    }
}

extern "C" int slam_describe(slam_ctx *ctx, const double *image, int H, int W, const int64_t *rc, int n,
                             const int32_t *pattern, int n_bits, double sigma, int window,
                             uint64_t *out_bits, int64_t *out_rc, int *n_out)
{
    ARG_TRY(ctx, ctx != nullptr && image != nullptr && H > 0 && W > 0 && n >= 0 && n_out != nullptr);
    ARG_TRY(ctx, pattern != nullptr && n_bits > 0 && n_bits % 64 == 0 && window > 0 && window % 2 == 1 && window <= BRIEF_MAXTAPS && sigma > 0);
    *n_out = 0;
    if (n == 0) return SLAM_OK;
    ARG_TRY(ctx, rc != nullptr && out_bits != nullptr && out_rc != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // keypoints whose +-ceil(window/2) box leaves the image are dropped (order kept)
    const int lim = (window + 1) / 2;
    std::vector<int64_t> keep; keep.reserve((size_t)n * 2);
    for (int k = 0; k < n; k++) {
        int64_t y = rc[2 * k], x = rc[2 * k + 1];
        if (y - lim < 1 || y + lim > H || x - lim < 1 || x + lim > W) continue;
        keep.push_back(y); keep.push_back(x);
    }
    const int m = (int)(keep.size() / 2);
    for (int b = 0; b < n_bits; b++)
        for (int c = 0; c < 4; c++)
            if (pattern[4 * b + c] < -lim || pattern[4 * b + c] > lim)
                return slam_fail(ctx, SLAM_ERR_ARG, "slam_describe: pattern offset %d outside +-%d", pattern[4 * b + c], lim);
    if (m == 0) return SLAM_OK;
    Taps t; t.n = window;
    { const int hw = window >> 1; double s = 0; for (int i = 0; i < window; i++) { double x = i - hw; t.w[i] = std::exp(-(x * x) / (2.0 * (sigma * sigma))); s += t.w[i]; } for (int i = 0; i < window; i++) t.w[i] = t.w[i] / s; }
    const size_t N = (size_t)H * W, words = (size_t)n_bits / 64;
    const size_t img_b = (N * 8 + 255) & ~(size_t)255, rc_b = ((size_t)m * 16 + 255) & ~(size_t)255, pat_b = ((size_t)n_bits * 16 + 255) & ~(size_t)255;
    char *s;
    int r = slam_scratch(ctx, 2 * img_b + rc_b + pat_b + (size_t)m * words * 8, (void **)&s);
    if (r) return r;
    double *d_a = (double *)s, *d_b = (double *)(s + img_b);
    int64_t *d_rc = (int64_t *)(s + 2 * img_b); int32_t *d_pat = (int32_t *)(s + 2 * img_b + rc_b);
    uint64_t *d_out = (uint64_t *)(s + 2 * img_b + rc_b + pat_b);
    HIP_TRY(ctx, hipMemcpyAsync(d_a, image, N * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_rc, keep.data(), (size_t)m * 16, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_pat, pattern, (size_t)n_bits * 16, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fir_y, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, d_b, (const double *)d_a, H, W, t);
    hipLaunchKernelGGL(k_fir_x, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, d_a, (const double *)d_b, H, W, t);
    hipLaunchKernelGGL(k_brief, dim3(m), dim3(64), 0, ctx->stream, (const double *)d_a, H, (const int64_t *)d_rc, (const int32_t *)d_pat, n_bits, d_out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_bits, d_out, (size_t)m * words * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    memcpy(out_rc, keep.data(), (size_t)m * 16);
    *n_out = m;
    return SLAM_OK;
}
