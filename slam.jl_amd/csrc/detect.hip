// detect.hip -- grid-cell Shi-Tomasi keypoint extraction on gfx950.
//
// Replaces detect / get_mask / _shi_tomasi of the reference
// (src/extractor.jl:24-42, 63-95, 116-122).  One workgroup per grid cell: the
// cell's pixels, the avoidance-mask halo and every intermediate plane live in
// LDS (four cell-sized planes; the avoidance mask is rasterised per disk row; the separable Sobel and box stages are evaluated
// per pixel from the 3x3 neighbourhood with exactly the two-stage arithmetic,
// so no intermediate plane is stored and four workgroups share a CU); HBM traffic is one read of the image plus the keypoint lists, i.e. the
// algorithmic minimum (8*H*W + 16*(K + n_out) bytes).  Keypoint order and
// indices are bit-exact with the CPU oracle: every sum runs in the reference's
// order and the build uses -ffp-contract=off.
#include "common.hpp"
#include <cmath>
#include <cstddef>

#define DET_THREADS 256
#define DET_MAXCAND 512
#define DET_MAXTAPS 41
#define DET_MAXR 64                // disk radius of the avoidance mask (extractor radius, SLAM.jl:158: 17)

struct DetectArgs {
    const double *img; int H, W, pitch;
    const double *cur; int n_cur;      // device (y,x) pairs
    int radius, grid_rows, grid_cols, cs, k;
    double min_response;
    int ntaps;                         // 0: no blur of the mask
    double taps[DET_MAXTAPS];
    int64_t *cell_out;                 // n_cells * k * 2
    int *cell_cnt;                     // n_cells
    // batched launch (grid.y = stream): per-stream image offset, slice of `cur` and k; nullptr for a single image
    const int *cur_off, *k_s; size_t zs; int kmax;
    // keypoint-set launch (slam_kpset_detect): stream z's current keypoints are cur[2 * z * cur_stride ..], cur_cnt[z] of them
    // (device-side count), k follows from it
    const int *cur_cnt; int cur_stride, max_points;
    // ImageDraw's disk test ((dy/r)^2 + (dx/r)^2 < 1, f64) as a table: lim[|dy|] = largest |dx| inside the disk (-1: none), formed once
    // on the host with the per-pixel test's operations (det_disk_table; every cell used to rebuild it: ~3 k of a cell's 39 k cycles)
    signed char lim[DET_MAXR + 1];
};
static void det_disk_table(DetectArgs &A)
{
    const int r = A.radius;
    for (int dy = 0; dy <= DET_MAXR; dy++) A.lim[dy] = -1;
    for (int dy = 0; dy <= r && dy <= DET_MAXR; dy++) {
        const volatile double a = (double)dy / (double)r;
        int lim = -1;
        for (int dx = 0; dx <= r; dx++) {
            const volatile double b = (double)dx / (double)r;
            const volatile double aa = a * a, bb = b * b;       // (separately rounded products: no contraction)
            if (aa + bb < 1) lim = dx; else break;
        }
        A.lim[dy] = (signed char)lim;
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Separable 3x3 correlation evaluated at one pixel with the arithmetic of the two-pass form (ImageFiltering:
// first factor along dim 1 into an intermediate, second factor along dim 2; acc = 0; acc += v[j]*k[j], j ascending;
// replicate border at the cell edge): the three intermediate values of columns x-1, x, x+1 are recomputed in
// registers instead of being stored as a plane.
__device__ __forceinline__ double sep3(const double *src, int h, int w, int y, int x,
                                       double ky0, double ky1, double ky2, double kx0, double kx1, double kx2)
{
    const int ym = clampi(y - 1, 0, h - 1), yp = clampi(y + 1, 0, h - 1);
    double t[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int xx = clampi(x + j - 1, 0, w - 1);
        double acc = 0.0;
        acc += src[ym + xx * h] * ky0; acc += src[y + xx * h] * ky1; acc += src[yp + xx * h] * ky2;
        t[j] = acc;
    }
    double acc = 0.0;
    acc += t[0] * kx0; acc += t[1] * kx1; acc += t[2] * kx2;
    return acc;
}

// imfilter(mask, Kernel.gaussian(sigma)) as two separable passes over the halo'd byte mask, then image .* mask.
// acc = 0; acc += v[j] * k[j], j ascending (ImageFiltering order) in both passes.  NT > 0: tap count known at compile time.
// Strips: a thread produces DET_SL consecutive outputs along the filtered dimension from one run of DET_SL + NT - 1 inputs held in
// registers (NT known at compile time) -- 3.4 instead of 13 LDS reads per output; every output is still acc = 0; acc += v[j] * k[j]
// with j ascending.
#define DET_SL 5
// i / d and i % d for a divisor that is only known at run time (cell height, strips per column, disk rows): the compiler expands such a division into
// ~25 instructions at every site.  One multiplication by m = ceil(2^32 / d), formed once per workgroup, is exact for i < 2^16 <= 2^32 / d (every index of a
// cell tile is far below that).
struct DetDiv {
    unsigned m, d;
    __device__ __forceinline__ explicit DetDiv(int dd) : m(0xffffffffu / (unsigned)(dd > 0 ? dd : 1) + 1u), d((unsigned)(dd > 0 ? dd : 1)) {}
    __device__ __forceinline__ int div(int i) const { return d == 1 ? i : (int)__umulhi((unsigned)i, m); }
    __device__ __forceinline__ int mod(int i) const { return i - div(i) * (int)d; }
};
template <int NT>
__device__ __forceinline__ void blur_mask(double *bA, double *T, const unsigned char *m0, const double *taps, int h, int w, int mh, int mw, int tid, const DetDiv &dns, const DetDiv &dh, int ntaps = NT)
{
    const int nt = NT > 0 ? NT : ntaps;
    if (NT > 0) {
        constexpr int NK = NT > 0 ? NT : 1;
        double k[NK];
#pragma unroll
        for (int j = 0; j < NK; j++) k[j] = taps[j];
        // dim-1 pass: T(y, tx) = sum_j m0(y+j, tx) * k[j]; thread = (column tx, strip of DET_SL rows)
        const int ns = (h + DET_SL - 1) / DET_SL;
        for (int it = tid; it < mw * ns; it += DET_THREADS) {
            const int tx = dns.div(it), ys = (it - tx * ns) * DET_SL;
            double v[DET_SL + NK - 1];
#pragma unroll
            for (int i = 0; i < DET_SL + NK - 1; i++) { const int yy = ys + i < mh ? ys + i : mh - 1; v[i] = (double)m0[yy + tx * mh]; }
#pragma unroll
            for (int q = 0; q < DET_SL; q++) {
                if (ys + q < h) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NK; j++) acc += v[q + j] * k[j];
                    T[(ys + q) + tx * h] = acc;
                }
            }
        }
        __syncthreads();
        // dim-2 pass and image .* mask; thread = (row y, strip of DET_SL columns)
        const int nsx = (w + DET_SL - 1) / DET_SL;
        for (int it = tid; it < h * nsx; it += DET_THREADS) {
            const int sx = dh.div(it), y = it - sx * h, xs = sx * DET_SL;
            double v[DET_SL + NK - 1];
#pragma unroll
            for (int i = 0; i < DET_SL + NK - 1; i++) { const int xx = xs + i < mw ? xs + i : mw - 1; v[i] = T[y + xx * h]; }
#pragma unroll
            for (int q = 0; q < DET_SL; q++) {
                if (xs + q < w) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NK; j++) acc += v[q + j] * k[j];
                    bA[y + (xs + q) * h] = bA[y + (xs + q) * h] * acc;
                }
            }
        }
        return;
    }
    // dim-1 pass: T(y, tx) = sum_j m0(y+j, tx) * k[j]
    for (int i = tid; i < h * mw; i += DET_THREADS) {
        const int tx = dh.div(i), y = i - tx * h;
        double acc = 0.0;
        for (int j = 0; j < nt; j++) acc += (double)m0[(y + j) + tx * mh] * taps[j];
        T[i] = acc;
    }
    __syncthreads();
    // dim-2 pass and image .* mask
    for (int i = tid; i < h * w; i += DET_THREADS) {
        const int x = dh.div(i), y = i - x * h;
        double acc = 0.0;
        for (int j = 0; j < nt; j++) acc += T[y + (x + j) * h] * taps[j];
        bA[i] = bA[i] * acc;
    }
}

// A (DET_SL + 2) x 3 neighbourhood of a cell plane in registers: rows ys - 1 .. ys + DET_SL, columns x - 1 .. x + 1, replicate
// border at the cell edge (clamped indices), for the DET_SL outputs (ys .. ys + DET_SL - 1, x).
struct DetStrip { double v[3][DET_SL + 2]; };
__device__ __forceinline__ void det_strip_load(DetStrip &S, const double *src, int h, int w, int ys, int x)
{
    const int xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
#pragma unroll
    for (int i = 0; i < DET_SL + 2; i++) {
        const int yy = clampi(ys - 1 + i, 0, h - 1);
        S.v[0][i] = src[yy + xm * h]; S.v[1][i] = src[yy + x * h]; S.v[2][i] = src[yy + xp * h];
    }
}
// sep3 (above) for output q of the strip: the same operations on the same operands
__device__ __forceinline__ double det_strip_sep3(const DetStrip &S, int q, double ky0, double ky1, double ky2, double kx0, double kx1, double kx2)
{
    double t[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double acc = 0.0;
        acc += S.v[j][q] * ky0; acc += S.v[j][q + 1] * ky1; acc += S.v[j][q + 2] * ky2;
        t[j] = acc;
    }
    double acc = 0.0;
    acc += t[0] * kx0; acc += t[1] * kx1; acc += t[2] * kx2;
    return acc;
}

#ifdef DET_TRACE
#define DT(k) do { __syncthreads(); dt_clk[k] = clock64() - dt_t0; } while (0)
#else
#define DT(k)
#endif
__global__ __launch_bounds__(DET_THREADS) void detect_cells(DetectArgs A)
{
    extern __shared__ double lds[];
#ifdef DET_TRACE
    long long dt_clk[12]; const long long dt_t0 = clock64();
    for (int q = 0; q < 12; q++) dt_clk[q] = 0;
#endif
    // per-stream view of the arguments in locals: writing to the argument struct (or indexing its tap array with a lane
    // index) makes the compiler copy all 472 bytes of it to scratch and read every field back from there
    const double *a_img = A.img, *a_cur = A.cur;
    int a_ncur = A.n_cur, a_k = A.k;
    int64_t *a_cell_out = A.cell_out; int *a_cell_cnt = A.cell_cnt;
    if (A.cur_cnt) {
        const int z = blockIdx.y, nc = A.grid_rows * A.grid_cols, ncur = A.cur_cnt[z];
        a_img += (size_t)z * A.zs; a_cur += 2 * (size_t)z * A.cur_stride; a_ncur = ncur;
        a_k = ncur >= A.max_points ? 0 : (A.max_points - ncur + nc - 1) / nc;                 // extractor.jl:64-66, 74-76
        a_cell_out += (size_t)z * nc * A.kmax * 2; a_cell_cnt += (size_t)z * nc;
    } else if (A.cur_off) {
        const int z = blockIdx.y, o = A.cur_off[z], nc = A.grid_rows * A.grid_cols;
        a_img += (size_t)z * A.zs; a_cur += 2 * (size_t)o; a_ncur = A.cur_off[z + 1] - o; a_k = A.k_s[z];
        a_cell_out += (size_t)z * nc * A.kmax * 2; a_cell_cnt += (size_t)z * nc;
    }
    const int cs = A.cs, H = A.H, W = A.W;
    const int cell = blockIdx.x;
    const int cyi = cell / A.grid_cols, cxi = cell % A.grid_cols;
    const int y0 = cyi * cs, x0 = cxi * cs;
    const int h = min(H, (cyi + 1) * cs) - y0, w = min(W, (cxi + 1) * cs) - x0;
    const int tid = threadIdx.x;
    const int n = cs * cs;
    double *bA = lds, *bB = lds + n, *bC = lds + 2 * n, *bD = lds + 3 * n;
    __shared__ int s_ncand, s_cnt;
    __shared__ __attribute__((aligned(16))) int s_cand[2 * DET_MAXCAND];
    __shared__ int s_lim[DET_MAXR + 1];
    __shared__ double s_taps[DET_MAXTAPS];

    if (h <= 0 || w <= 0 || a_k <= 0) { if (tid == 0) a_cell_cnt[cell] = 0; return; }

    // taps -> LDS once.  Indexing A.taps with a lane index makes the compiler copy the argument struct to scratch, and 41 statically
    // indexed copies are 41 serialised scalar loads (6.5 k cycles): read the kernel-argument segment itself with one vector load per
    // lane instead (detect_cells has a single argument: the struct starts at offset 0 of the segment).
    if (a_ncur > 0 && A.ntaps > 0 && tid < A.ntaps) {
        const double *kt = (const double *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(DetectArgs, taps));
        s_taps[tid] = kt[tid];
    }
    // ---- image tile -> bA ---------------------------------------------------
    const DetDiv dvh(h), dnstr((h + DET_SL - 1) / DET_SL);
    for (int i0 = tid; i0 < h * w; i0 += 5 * DET_THREADS) {          // five loads per thread in flight (a 35 x 35 cell: one trip)
        double v[5];
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int i = i0 + u * DET_THREADS, ic = i < h * w ? i : i0;
            const int icx = dvh.div(ic);
            v[u] = a_img[(size_t)(y0 + ic - icx * h) + (size_t)(x0 + icx) * A.pitch];
        }
#pragma unroll
        for (int u = 0; u < 5; u++) { const int i = i0 + u * DET_THREADS; if (i < h * w) bA[i] = v[u]; }
    }

    DT(0);
    // ---- avoidance mask (get_mask + imfilter(mask, Kernel.gaussian) + .*) ---
    if (a_ncur > 0) {
        const int hw = A.ntaps >> 1;
        const int mh = h + 2 * hw, mw = w + 2 * hw;
        const int r = A.radius;
        if (tid == 0) s_ncand = 0;
        // ImageDraw's disk test ((dy/r)^2 + (dx/r)^2 < 1, f64) as a table: s_lim[|dy|] = largest |dx| inside the
        // disk (-1: none).  Same operations on the same operands as the per-pixel test, evaluated once per cell
        // instead of two f64 divisions per (pixel, keypoint) pair.
        // (the r + 1 quotients d / r are formed once, one per thread, instead of r + 1 divisions in sequence by every table thread)
        if (tid <= r) s_lim[tid] = ((const signed char *)__builtin_amdgcn_kernarg_segment_ptr())[offsetof(DetectArgs, lim) + tid];      // (a lane-indexed read of the by-value struct would copy all of it to scratch)
        __syncthreads();
        // candidate keypoints: disk (+halo) touches the clamped tile region
        const int ylo = clampi(y0 - hw, 0, H - 1) + 1, yhi = clampi(y0 + h - 1 + hw, 0, H - 1) + 1; // 1-based
        const int xlo = clampi(x0 - hw, 0, W - 1) + 1, xhi = clampi(x0 + w - 1 + hw, 0, W - 1) + 1;
        // (four keypoints per thread requested at once: one L2 round trip per 1024 keypoints instead of one per 256)
        for (int k0 = tid; k0 < a_ncur; k0 += 4 * DET_THREADS) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int k = k0 + u * DET_THREADS; v[u] = ((const double2 *)a_cur)[k < a_ncur ? k : k0]; }
            // compaction by wave ballot: a hit's slot is the wave's base (one LDS atomic per wave and pass) + the number of hits in
            // the lower lanes (popcount of the ballot below the lane) -- no per-hit atomics
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool live = k0 + u * DET_THREADS < a_ncur;
                const long py = (long)rint(v[u].x), px = (long)rint(v[u].y);
                const bool hit = live && py + r >= ylo && py - r <= yhi && px + r >= xlo && px - r <= xhi;
                const unsigned long long bal = __ballot(hit);
                if (bal) {                                                        // wave-uniform
                    int base = 0;
                    if ((tid & 63) == 0) base = atomicAdd(&s_ncand, __popcll(bal));
                    base = __shfl(base, 0);
                    const int slot = base + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                    if (hit && slot < DET_MAXCAND) { s_cand[2 * slot] = (int)py; s_cand[2 * slot + 1] = (int)px; }
                }
            }
        }
        __syncthreads();
        DT(1);
        const int ncand = s_ncand;
        const bool overflow = ncand > DET_MAXCAND;
        // raw mask over the halo'd tile (replicate = clamped coordinates) -> bB (as 0/1 doubles)
        // the halo'd raw mask is stored as bytes
        unsigned char *m0 = (unsigned char *)(lds + 4 * n) - ((mh * mw + 7) & ~7);   // byte mask at the tail of bD (sizes checked on host)
        if (!overflow) {
            // Rasterised: the mask starts as ones, every (candidate, disk row) pair clears its x-interval (ImageDraw draws
            // in-image pixels only), then halo pixels outside the image replicate the border pixel they clamp to.  Same
            // mask as testing every pixel against every candidate, ~5x fewer LDS operations.
            for (int i = tid; i < mh * mw; i += DET_THREADS) m0[i] = 1;
            __syncthreads();
            const int nrow = 2 * r + 1, ty0 = y0 - hw, tx0 = x0 - hw;
            const DetDiv dnrow(nrow), dmh(mh);
            for (int q = tid; q < ncand * nrow; q += DET_THREADS) {
                const int c = dnrow.div(q), dyi = q - c * nrow - r;
                const int yy = s_cand[2 * c] + dyi, px = s_cand[2 * c + 1];      // 1-based image coordinates
                const int lim = s_lim[abs(dyi)];
                const int ty = (yy - 1) - ty0;
                if (yy < 1 || yy > H || lim < 0 || ty < 0 || ty >= mh) continue;
                int txa = (max(px - lim, 1) - 1) - tx0, txb = (min(px + lim, W) - 1) - tx0;
                txa = txa < 0 ? 0 : txa; txb = txb > mw - 1 ? mw - 1 : txb;
                for (int tx = txa; tx <= txb; tx++) m0[ty + tx * mh] = 0;
            }
            __syncthreads();
            if (ty0 < 0 || ty0 + mh - 1 > H - 1 || tx0 < 0 || tx0 + mw - 1 > W - 1) {        // border cell: replicate
                for (int i = tid; i < mh * mw; i += DET_THREADS) {
                    const int imx = dmh.div(i);
                    const int uy = ty0 + i - imx * mh, ux = tx0 + imx;
                    if (uy < 0 || uy > H - 1 || ux < 0 || ux > W - 1)
                        m0[i] = m0[(clampi(uy, 0, H - 1) - ty0) + (clampi(ux, 0, W - 1) - tx0) * mh];
                }
            }
        } else {
            for (int i = tid; i < mh * mw; i += DET_THREADS) {
                const int yy = clampi(y0 + i % mh - hw, 0, H - 1) + 1, xx = clampi(x0 + i / mh - hw, 0, W - 1) + 1; // 1-based
                unsigned char m = 1;
                for (int c = 0; c < a_ncur; c++) {
                    long py = (long)rint(a_cur[2 * c]), px = (long)rint(a_cur[2 * c + 1]);
                    const long dy = labs(yy - py), dx = labs(xx - px);
                    if (dy <= r && dx <= s_lim[dy]) { m = 0; break; }
                }
                m0[i] = m;
            }
        }
        DT(2);
        __syncthreads();
        if (A.ntaps > 0) {
            double *T = bB;                                   // h*mw doubles: bB, bC and the part of bD below the byte mask (checked on host)
            if (A.ntaps == 13) blur_mask<13>(bA, T, m0, s_taps, h, w, mh, mw, tid, dnstr, dvh);      // sigma_mask = 3 (the default): unrolled
            else blur_mask<0>(bA, T, m0, s_taps, h, w, mh, mw, tid, dnstr, dvh, A.ntaps);
        } else {
            for (int i = tid; i < h * w; i += DET_THREADS) {
                const int x = dvh.div(i), y = i - x * h;
                bA[i] = bA[i] * (double)m0[(y + hw) + (x + hw) * mh];
            }
        }
    }
    __syncthreads();

    DT(3);
    // ---- Images.shi_tomasi on the cell view (replicate border at cell edges) -
    // imgradients(cell, KernelFactors.sobel): g1 = (d/dy along dim 1, then (1,2,1)/4 along dim 2),
    // g2 = ((1,2,1)/4 along dim 1, then d/dx along dim 2); products -> bB, bC, bD
    // thread = (column x, strip of DET_SL rows): 21 LDS reads per DET_SL pixels and plane instead of 9 per pixel
    const int nstr = (h + DET_SL - 1) / DET_SL;
    for (int it = tid; it < w * nstr; it += DET_THREADS) {
        const int x = dnstr.div(it), ys = (it - x * nstr) * DET_SL;
        DetStrip S;
        det_strip_load(S, bA, h, w, ys, x);
#pragma unroll
        for (int q = 0; q < DET_SL; q++) {
            if (ys + q < h) {
                const double g1 = det_strip_sep3(S, q, -1.0 / 2, 0.0 / 2, 1.0 / 2, 1.0 / 4, 2.0 / 4, 1.0 / 4);
                const double g2 = det_strip_sep3(S, q, 1.0 / 4, 2.0 / 4, 1.0 / 4, -1.0 / 2, 0.0 / 2, 1.0 / 2);
                const int i = (ys + q) + x * h;
                bB[i] = g1 * g1; bC[i] = g1 * g2; bD[i] = g2 * g2;
            }
        }
    }
    __syncthreads();
    DT(4);
    // 3x3 box mean of the products (1/3 x 1/3, two-pass arithmetic) and the min-eigenvalue response -> bA
    double *resp = bA;
    for (int it = tid; it < w * nstr; it += DET_THREADS) {
        const int x = dnstr.div(it), ys = (it - x * nstr) * DET_SL;
        double xx[DET_SL], xy[DET_SL], yy[DET_SL];
        DetStrip S;
        det_strip_load(S, bB, h, w, ys, x);
#pragma unroll
        for (int q = 0; q < DET_SL; q++) xx[q] = det_strip_sep3(S, q, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3);
        det_strip_load(S, bC, h, w, ys, x);
#pragma unroll
        for (int q = 0; q < DET_SL; q++) xy[q] = det_strip_sep3(S, q, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3);
        det_strip_load(S, bD, h, w, ys, x);
#pragma unroll
        for (int q = 0; q < DET_SL; q++) yy[q] = det_strip_sep3(S, q, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3, 1.0 / 3);
#pragma unroll
        for (int q = 0; q < DET_SL; q++) {
            if (ys + q < h) {
                const double dd = xx[q] - yy[q];
                resp[(ys + q) + x * h] = ((xx[q] + yy[q]) - sqrt(dd * dd + 4 * (xy[q] * xy[q]))) / 2;
            }
        }
    }
    __syncthreads();
    DT(5);

    // ---- findlocalmaxima: strict, 8-neighbourhood, edges included (neighbours outside the cell are not compared) ------------
    // The strict maxima (typically 20-40 of a 35 x 35 cell) are compacted as they are found -- wave ballot + popcount of the lower lanes,
    // one LDS atomic per wave and strip pass -- into a list of (pixel index, response); the top-k selection then runs on ONE wave
    // over that list: k rounds of a wave arg-max on the key (response descending, pixel index ascending = Julia's stable sortperm over
    // the column-major maxima, extractor.jl:27-40), no workgroup barrier and no merge thread per round.
    int *mx_idx = (int *)bC;                                  // compacted maxima: pixel index (column-major) ...
    double *mx_val = bD;                                      // ... and response
    if (tid == 0) { s_cnt = 0; s_ncand = 0; }
    __syncthreads();
    for (int it0 = 0; it0 < w * nstr; it0 += DET_THREADS) {   // (whole waves enter every pass: the ballots need them)
        const int it = it0 + tid;
        const bool act = it < w * nstr;
        const int x = act ? dnstr.div(it) : 0, ys = act ? (it - x * nstr) * DET_SL : 0;
        DetStrip S;
        det_strip_load(S, resp, h, w, ys, x);
        const bool cl = x > 0, cr = x < w - 1;
#pragma unroll
        for (int q = 0; q < DET_SL; q++) {
            const int y = ys + q;
            const double c = S.v[1][q + 1];
            bool ismax = act && y < h;
            const bool ru = y > 0, rd = y < h - 1;
            if (cl) { if (ru && !(S.v[0][q] < c)) ismax = false; if (!(S.v[0][q + 1] < c)) ismax = false; if (rd && !(S.v[0][q + 2] < c)) ismax = false; }
            if (ru && !(S.v[1][q] < c)) ismax = false;
            if (rd && !(S.v[1][q + 2] < c)) ismax = false;
            if (cr) { if (ru && !(S.v[2][q] < c)) ismax = false; if (!(S.v[2][q + 1] < c)) ismax = false; if (rd && !(S.v[2][q + 2] < c)) ismax = false; }
            const unsigned long long bal = __ballot(ismax);
            if (bal) {
                int base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&s_ncand, __popcll(bal));
                base = __shfl(base, 0);
                const int slot = base + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                if (ismax) { mx_idx[slot] = y + x * h; mx_val[slot] = c; }      // (at most h w / 4 strict maxima: the planes have room)
            }
        }
    }
    __syncthreads();

    DT(6);
    // ---- top-k by response (stable: ties keep column-major order), wave 0 only ------------
    int *sel = (int *)bB;                                     // selected pixel indices
    if (tid < 64) {
        const int nmx = s_ncand;
        int cnt = 0;
        for (int round = 0; round < a_k; round++) {
            double bv = -INFINITY; int bi = 0x7fffffff, bs = -1;
            for (int j = tid; j < nmx; j += 64) {
                const int i = mx_idx[j];
                if (i >= 0) {                                                     // (taken entries are marked -1)
                    const double v = mx_val[j];
                    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; bs = j; }
                }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(bv, off); const int oi = __shfl_xor(bi, off), os = __shfl_xor(bs, off);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; bs = os; }
            }
            if (bi == 0x7fffffff) break;                                          // wave-uniform after the butterfly
            if (tid == 0) {
                mx_idx[bs] = -1;
                if (!(bv < A.min_response)) sel[cnt] = bi;                       // `responses[mx] < min_response && continue`
            }
            if (!(bv < A.min_response)) cnt++;
        }
        if (tid == 0) s_cnt = cnt;
    }
    __syncthreads();

    DT(7);
    // ---- emit in column-major order (Keypoints(::Matrix{Bool}) = findall) ----
    if (tid == 0) {
        int cnt = s_cnt;
        for (int i = 1; i < cnt; i++) { int v = sel[i], j = i - 1; while (j >= 0 && sel[j] > v) { sel[j + 1] = sel[j]; j--; } sel[j + 1] = v; }
        int64_t *o = a_cell_out + (size_t)cell * a_k * 2;
        for (int i = 0; i < cnt; i++) {
            int y = sel[i] % h, x = sel[i] / h;
            o[2 * i] = y + 1 + y0; o[2 * i + 1] = x + 1 + x0;
        }
        a_cell_cnt[cell] = cnt;
    }
#ifdef DET_TRACE
    DT(8);
    if (tid == 0 && cell == 200 && blockIdx.y == 3) printf("detect cell: ncur %d k %d | tile %lld scan %lld raster %lld blur %lld grad %lld resp %lld lmax %lld topk %lld emit %lld cycles\n", a_ncur, a_k, dt_clk[0], dt_clk[1] - dt_clk[0], dt_clk[2] - dt_clk[1], dt_clk[3] - dt_clk[2], dt_clk[4] - dt_clk[3], dt_clk[5] - dt_clk[4], dt_clk[6] - dt_clk[5], dt_clk[7] - dt_clk[6], dt_clk[8] - dt_clk[7]);
#endif
}

// Ordered compaction of the per-cell lists (cells row-major: extractor.jl:81):
// wave-level inclusive scans (shuffle) + one LDS hop across the 16 waves.
__global__ __launch_bounds__(1024) void detect_compact(const int64_t *cell_out, const int *cell_cnt, int n_cells, int k,
                                                        int64_t *out /* [0] = n_out, then pairs */, int cap,
                                                        const int *k_s /* batched: grid.x = stream, per-stream k, lists strided by kmax = k */)
{
    if (k_s) {
        const int z = blockIdx.x;
        cell_out += (size_t)z * n_cells * k * 2; cell_cnt += (size_t)z * n_cells; out += (size_t)z * (1 + 2 * (size_t)cap);
        k = k_s[z];
    }
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_cells; c0 += 1024) {
        int c = c0 + tid;
        int cnt = c < n_cells ? cell_cnt[c] : 0;
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int wbase = 0;
        for (int i = 0; i < wv; i++) wbase += s_w[i];
        int total = 0;
        for (int i = 0; i < 16; i++) total += s_w[i];
        int start = s_base + wbase + incl - cnt;
        for (int i = 0; i < cnt; i++)
            if (start + i < cap) {
                out[1 + 2 * (start + i)] = cell_out[((size_t)c * k + i) * 2];
                out[2 + 2 * (start + i)] = cell_out[((size_t)c * k + i) * 2 + 1];
            }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) out[0] = s_base;
}

int slam_detect_device(slam_ctx *ctx, const double *img_dev, int H, int W, int pitch, const double *cur_yx, int n_cur,
                       int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                       double sigma_mask, double min_response, int64_t *out_rc, int cap, int *n_out)
{
    ARG_TRY(ctx, H > 0 && W > 0 && grid_rows > 0 && grid_cols > 0 && cell_size >= 8 && radius > 0 && radius <= DET_MAXR && n_out != nullptr);
    ARG_TRY(ctx, n_cur >= 0 && (n_cur == 0 || cur_yx != nullptr));
    *n_out = 0;
    if (n_cur >= max_points) return SLAM_OK;                      // extractor.jl:64
    const int n_cells = grid_rows * grid_cols;
    const int n_detect = max_points - n_cur;
    const int k = (n_detect + n_cells - 1) / n_cells;             // ceil(Int, n_detect / n_cells)
    DetectArgs A;
    A.img = img_dev; A.H = H; A.W = W; A.pitch = pitch; A.n_cur = n_cur; A.radius = radius; det_disk_table(A);
    A.grid_rows = grid_rows; A.grid_cols = grid_cols; A.cs = cell_size; A.k = k; A.min_response = min_response;
    A.ntaps = 0; A.cur_off = nullptr; A.k_s = nullptr; A.zs = 0; A.kmax = k; A.cur_cnt = nullptr; A.cur_stride = 0; A.max_points = max_points;
    if (n_cur > 0 && sigma_mask != 0) {
        int l = 4 * (int)std::ceil(sigma_mask) + 1;
        ARG_TRY(ctx, l <= DET_MAXTAPS);
        A.ntaps = slam_gaussian_taps(sigma_mask, A.taps);
    }
    const int hw = A.ntaps >> 1;
    const size_t n = (size_t)cell_size * cell_size;
    // LDS carve-up checks (4 planes of cell_size^2 doubles)
    {   // byte mask at the tail of the 4th plane, blur intermediate in planes 2..4 below it
        const size_t mbytes = ((size_t)(cell_size + 2 * hw) * (cell_size + 2 * hw) + 7) & ~(size_t)7;
        ARG_TRY(ctx, mbytes <= n * 8 && (size_t)cell_size * (cell_size + 2 * hw) * 8 + mbytes <= 3 * n * 8);
    }
    ARG_TRY(ctx, (size_t)k * sizeof(int) <= n * 8);
    const size_t lds_bytes = 4 * n * sizeof(double);
    ARG_TRY(ctx, lds_bytes <= 150 * 1024);

    // scratch: [cur (2*n_cur doubles)] [cell_cnt (n_cells int, padded)] [cell_out] [out header+pairs]
    const size_t cur_b = ((size_t)n_cur * 16 + 255) & ~(size_t)255;
    const size_t cnt_b = ((size_t)n_cells * 4 + 255) & ~(size_t)255;
    const size_t cout_b = ((size_t)n_cells * k * 16 + 255) & ~(size_t)255;
    const size_t out_pairs = (size_t)n_cells * k;
    const size_t out_b = 8 + out_pairs * 16;
    char *s;
    int rc = slam_scratch(ctx, cur_b + cnt_b + cout_b + out_b, (void **)&s);
    if (rc) return rc;
    double *d_cur = (double *)s; int *d_cnt = (int *)(s + cur_b);
    int64_t *d_cout = (int64_t *)(s + cur_b + cnt_b); int64_t *d_out = (int64_t *)(s + cur_b + cnt_b + cout_b);
    if (n_cur > 0) HIP_TRY(ctx, hipMemcpyAsync(d_cur, cur_yx, (size_t)n_cur * 16, hipMemcpyHostToDevice, ctx->stream));
    A.cur = d_cur; A.cell_out = d_cout; A.cell_cnt = d_cnt;
    // > 64 KB of dynamic LDS needs the opt-in; per device, so set it on every call (cheap host-side call)
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)detect_cells, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    { ProfScope span(ctx, "detect");
      hipLaunchKernelGGL(detect_cells, dim3(n_cells), dim3(DET_THREADS), lds_bytes, ctx->stream, A);
      hipLaunchKernelGGL(detect_compact, dim3(1), dim3(1024), 0, ctx->stream, d_cout, d_cnt, n_cells, k, d_out, (int)out_pairs, (const int *)nullptr); }
    HIP_TRY(ctx, hipGetLastError());
    int64_t *h_out;
    rc = slam_pinned(ctx, out_b, (void **)&h_out);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    int64_t cnt = h_out[0];
    if (cnt > cap) return slam_fail(ctx, SLAM_ERR_CAPACITY, "slam_detect: %lld keypoints but cap = %d", (long long)cnt, cap);
    memcpy(out_rc, h_out + 1, (size_t)cnt * 16);
    *n_out = (int)cnt;
    return SLAM_OK;
}

extern "C" int slam_detect(slam_ctx *ctx, const double *image, int H, int W, const double *cur_yx, int n_cur,
                           int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                           double sigma_mask, double min_response, int64_t *out_rc, int cap, int *n_out)
{
    ARG_TRY(ctx, ctx != nullptr);
    ARG_TRY(ctx, image != nullptr && H > 0 && W > 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d_img;
    int rc = slam_scratch2(ctx, (size_t)H * W * 8, &d_img);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_img, image, (size_t)H * W * 8, hipMemcpyHostToDevice, ctx->stream));
    return slam_detect_device(ctx, (const double *)d_img, H, W, H, cur_yx, n_cur, max_points, radius, grid_rows, grid_cols,
                              cell_size, sigma_mask, min_response, out_rc, cap, n_out);
}

extern "C" int slam_detect_pyr(slam_ctx *ctx, const slam_pyr *pyr, const double *cur_yx, int n_cur,
                               int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                               double sigma_mask, double min_response, int64_t *out_rc, int cap, int *n_out)
{
    ARG_TRY(ctx, ctx != nullptr && pyr != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return slam_detect_device(ctx, pyr->plane(0, 0), pyr->H[0], pyr->W[0], pyr->P[0], cur_yx, n_cur, max_points, radius, grid_rows,
                              grid_cols, cell_size, sigma_mask, min_response, out_rc, cap, n_out);
}

// detect() for the S members of a pyramid batch in one launch (grid.y = stream).  cur_yx holds the current
// keypoints of all streams back to back, cur_off[s] .. cur_off[s+1] those of stream s; the new keypoints come
// back the same way (out_off[S+1]).  Per stream identical to slam_detect_pyr on that stream's pyramid.
extern "C" int slam_detect_batch(slam_ctx *ctx, const slam_pyr *pyr0, int S, const double *cur_yx, const int32_t *cur_off,
                                 int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                                 double sigma_mask, double min_response, int64_t *out_rc, int cap, int32_t *out_off)
{
    ARG_TRY(ctx, ctx != nullptr && pyr0 != nullptr && S >= 1 && S <= 128 && cur_off != nullptr && out_off != nullptr);
    ARG_TRY(ctx, pyr0->batch_index == 0 && pyr0->batch_size == S);
    ARG_TRY(ctx, grid_rows > 0 && grid_cols > 0 && cell_size >= 8 && radius > 0 && radius <= DET_MAXR && cap >= 0 && (cap == 0 || out_rc != nullptr));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n_cells = grid_rows * grid_cols, n_tot = cur_off[S];
    ARG_TRY(ctx, cur_off[0] == 0 && n_tot >= 0 && (n_tot == 0 || cur_yx != nullptr));
    int ks[128], kmax = 0;
    for (int s = 0; s < S; s++) {
        const int nc = cur_off[s + 1] - cur_off[s];
        ARG_TRY(ctx, nc >= 0);
        ks[s] = nc >= max_points ? 0 : (max_points - nc + n_cells - 1) / n_cells;     // extractor.jl:64-66
        kmax = ks[s] > kmax ? ks[s] : kmax;
    }
    for (int s = 0; s <= S; s++) out_off[s] = 0;
    if (kmax == 0) return SLAM_OK;
    DetectArgs A;
    A.img = pyr0->plane(0, 0); A.H = pyr0->H[0]; A.W = pyr0->W[0]; A.pitch = pyr0->P[0]; A.zs = pyr0->zstride;
    A.n_cur = 0; A.radius = radius; det_disk_table(A); A.grid_rows = grid_rows; A.grid_cols = grid_cols; A.cs = cell_size; A.k = 0; A.kmax = kmax;
    A.min_response = min_response; A.ntaps = 0; A.cur_cnt = nullptr; A.cur_stride = 0; A.max_points = max_points;
    if (sigma_mask != 0) {
        ARG_TRY(ctx, 4 * (int)std::ceil(sigma_mask) + 1 <= DET_MAXTAPS);
        A.ntaps = slam_gaussian_taps(sigma_mask, A.taps);
    }
    const int hw = A.ntaps >> 1;
    const size_t n = (size_t)cell_size * cell_size;
    {   // byte mask at the tail of the 4th plane, blur intermediate in planes 2..4 below it
        const size_t mbytes = ((size_t)(cell_size + 2 * hw) * (cell_size + 2 * hw) + 7) & ~(size_t)7;
        ARG_TRY(ctx, mbytes <= n * 8 && (size_t)cell_size * (cell_size + 2 * hw) * 8 + mbytes <= 3 * n * 8);
    }
    ARG_TRY(ctx, (size_t)kmax * sizeof(int) <= n * 8);
    const size_t lds_bytes = 4 * n * sizeof(double);
    ARG_TRY(ctx, lds_bytes <= 150 * 1024);

    // host -> device in one copy: [cur_off (S+1 ints) at 0] [k_s (S ints) at 1024] [cur (2 * n_tot doubles) at 2048]
    const size_t hdr_b = 2048, cur_b = ((size_t)n_tot * 16 + 255) & ~(size_t)255;
    const size_t cnt_b = ((size_t)S * n_cells * 4 + 255) & ~(size_t)255;
    const size_t cout_b = ((size_t)S * n_cells * kmax * 16 + 255) & ~(size_t)255;
    const size_t pairs = (size_t)n_cells * kmax, out_b = (size_t)S * (8 + pairs * 16);
    char *d, *h;
    int rc = slam_scratch(ctx, hdr_b + cur_b + cnt_b + cout_b + out_b, (void **)&d);
    if (rc) return rc;
    rc = slam_pinned(ctx, hdr_b + cur_b + out_b, (void **)&h);
    if (rc) return rc;
    memcpy(h, cur_off, (size_t)(S + 1) * 4); memcpy(h + 1024, ks, (size_t)S * 4);
    if (n_tot > 0) memcpy(h + hdr_b, cur_yx, (size_t)n_tot * 16);
    HIP_TRY(ctx, hipMemcpyAsync(d, h, hdr_b + (size_t)n_tot * 16, hipMemcpyHostToDevice, ctx->stream));
    A.cur_off = (const int *)d; A.k_s = (const int *)(d + 1024); A.cur = (const double *)(d + hdr_b);
    A.cell_cnt = (int *)(d + hdr_b + cur_b); A.cell_out = (int64_t *)(d + hdr_b + cur_b + cnt_b);
    int64_t *d_out = (int64_t *)(d + hdr_b + cur_b + cnt_b + cout_b);
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)detect_cells, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    { ProfScope span(ctx, "detect");
      hipLaunchKernelGGL(detect_cells, dim3(n_cells, S), dim3(DET_THREADS), lds_bytes, ctx->stream, A);
      hipLaunchKernelGGL(detect_compact, dim3(S), dim3(1024), 0, ctx->stream, (const int64_t *)A.cell_out, (const int *)A.cell_cnt, n_cells, kmax,
                         d_out, (int)pairs, A.k_s); }
    HIP_TRY(ctx, hipGetLastError());
    int64_t *h_out = (int64_t *)(h + hdr_b + cur_b);
    HIP_TRY(ctx, hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    size_t tot = 0;
    for (int s = 0; s < S; s++) tot += (size_t)h_out[(size_t)s * (1 + 2 * pairs)];
    if (tot > (size_t)cap) return slam_fail(ctx, SLAM_ERR_CAPACITY, "slam_detect_batch: %zu keypoints but cap = %d", tot, cap);
    size_t o = 0;
    for (int s = 0; s < S; s++) {
        const int64_t *hs = h_out + (size_t)s * (1 + 2 * pairs);
        memcpy(out_rc + 2 * o, hs + 1, (size_t)hs[0] * 16);
        o += (size_t)hs[0]; out_off[s + 1] = (int32_t)o;
    }
    return SLAM_OK;
}


// detect() into a device-resident keypoint set: the avoidance list of stream z is its current list in the set, the new
// keypoints (cells row-major, column-major inside a cell: extractor.jl:81-91) are appended behind it as (row, col) Float64
// pixels with is_3d = 0 and fresh ids -- extract_keypoints! + add_keypoints_to_frame! (map_manager.jl:98-113) on arrays.
// One 1024-thread workgroup per stream: wave-level inclusive scans + one LDS hop (as detect_compact).
__global__ __launch_bounds__(1024) void detect_append(const int64_t *cell_out, const int *cell_cnt, int n_cells, int kmax, int max_points,
                                                       double *yx, double *syx, double *xyz, int64_t *id, uint8_t *is3d, uint8_t *stereo, uint8_t *haskf,
                                                       int *count, int64_t *next_id, int cap)
{
    const int z = blockIdx.x;
    cell_out += (size_t)z * n_cells * kmax * 2; cell_cnt += (size_t)z * n_cells;
    const size_t b = (size_t)z * cap;
    const int n0 = count[z];
    const int64_t id0 = next_id[z];
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    if (n0 >= max_points) return;                                 // extractor.jl:64: nothing detected (cell counts are zero as well)
    const int kz = (max_points - n0 + n_cells - 1) / n_cells;      // this stream's per-cell quota = the stride of its cell lists (detect_cells)
    for (int c0 = 0; c0 < n_cells; c0 += 1024) {
        const int c = c0 + tid;
        const int cnt = c < n_cells ? cell_cnt[c] : 0;
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int wbase = 0;
        for (int i = 0; i < wv; i++) wbase += s_w[i];
        int total = 0;
        for (int i = 0; i < 16; i++) total += s_w[i];
        const int start = s_base + wbase + incl - cnt;
        for (int i = 0; i < cnt; i++) {
            const int j = n0 + start + i;
            if (j < cap) {
                const size_t q = b + j;
                yx[2 * q] = (double)cell_out[((size_t)c * kz + i) * 2]; yx[2 * q + 1] = (double)cell_out[((size_t)c * kz + i) * 2 + 1];
                syx[2 * q] = 0.0; syx[2 * q + 1] = 0.0; xyz[3 * q] = 0.0; xyz[3 * q + 1] = 0.0; xyz[3 * q + 2] = 0.0;
                id[q] = id0 + start + i; is3d[q] = 0; stereo[q] = 0; haskf[q] = 0;
            }
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) { const int n1 = n0 + s_base; count[z] = n1 < cap ? n1 : cap; next_id[z] = id0 + s_base; }
}

extern "C" int slam_kpset_detect(slam_ctx *ctx, slam_kpset *ks, const slam_pyr *pyr0, int max_points, int radius, int grid_rows, int grid_cols,
                                 int cell_size, double sigma_mask, double min_response)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && pyr0 != nullptr);
    const int S = ks->S;
    ARG_TRY(ctx, pyr0->batch_index == 0 && pyr0->batch_size >= S);
    ARG_TRY(ctx, grid_rows > 0 && grid_cols > 0 && cell_size >= 8 && radius > 0 && radius <= DET_MAXR && max_points > 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n_cells = grid_rows * grid_cols;
    const int kmax = (max_points + n_cells - 1) / n_cells;         // n_cur = 0
    ARG_TRY(ctx, ks->cap >= max_points + n_cells);                  // a stream below max_points may receive up to n_cells * k > max_points - n_cur keypoints
    DetectArgs A;
    A.img = pyr0->plane(0, 0); A.H = pyr0->H[0]; A.W = pyr0->W[0]; A.pitch = pyr0->P[0]; A.zs = pyr0->zstride;
    A.n_cur = 0; A.radius = radius; det_disk_table(A); A.grid_rows = grid_rows; A.grid_cols = grid_cols; A.cs = cell_size; A.k = 0; A.kmax = kmax;
    A.min_response = min_response; A.ntaps = 0; A.cur_off = nullptr; A.k_s = nullptr;
    A.cur = ks->yx; A.cur_cnt = ks->count; A.cur_stride = ks->cap; A.max_points = max_points;
    if (sigma_mask != 0) {
        ARG_TRY(ctx, 4 * (int)std::ceil(sigma_mask) + 1 <= DET_MAXTAPS);
        A.ntaps = slam_gaussian_taps(sigma_mask, A.taps);
    }
    const int hw = A.ntaps >> 1;
    const size_t n = (size_t)cell_size * cell_size;
    {
        const size_t mbytes = ((size_t)(cell_size + 2 * hw) * (cell_size + 2 * hw) + 7) & ~(size_t)7;
        ARG_TRY(ctx, mbytes <= n * 8 && (size_t)cell_size * (cell_size + 2 * hw) * 8 + mbytes <= 3 * n * 8);
    }
    ARG_TRY(ctx, (size_t)kmax * sizeof(int) <= n * 8);
    const size_t lds_bytes = 4 * n * sizeof(double);
    ARG_TRY(ctx, lds_bytes <= 150 * 1024);
    const size_t cnt_b = ((size_t)S * n_cells * 4 + 255) & ~(size_t)255;
    const size_t cout_b = ((size_t)S * n_cells * kmax * 16 + 255) & ~(size_t)255;
    char *d;
    int rc = slam_scratch(ctx, cnt_b + cout_b, (void **)&d);
    if (rc) return rc;
    A.cell_cnt = (int *)d; A.cell_out = (int64_t *)(d + cnt_b);
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)detect_cells, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    { ProfScope span(ctx, "detect");
      hipLaunchKernelGGL(detect_cells, dim3(n_cells, S), dim3(DET_THREADS), lds_bytes, ctx->stream, A);
      hipLaunchKernelGGL(detect_append, dim3(S), dim3(1024), 0, ctx->stream, (const int64_t *)A.cell_out, (const int *)A.cell_cnt, n_cells, kmax, max_points,
                         ks->yx, ks->syx, ks->xyz, ks->id, ks->is3d, ks->stereo, ks->haskf, ks->count, ks->next_id, ks->cap); }
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
