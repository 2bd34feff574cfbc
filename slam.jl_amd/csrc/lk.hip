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
// Mapping: a 64-lane wave owns a point (a level visit stages its target footprint once into an LDS patch by LDS-DMA and iterates
// there: see stage_rect / lk_level); the (2w+1)^2 window is dealt to lanes
// in the reference's iteration order (q outer, p inner; element e -> lane
// e % 64, sequential per lane) and the two sums of prepare_linear_system are
// folded with a 6-step xor butterfly (wave_sum2: m = 32, 16, 1, 2, 4, 8).  That summation order is restated in the
// CPU oracle (sum_order = 1) and is bit-reproducible; it differs from the
// reference's single-accumulator order only in rounding (<= 1e-12 px observed).
#include "common.hpp"
#include <vector>
#include <cmath>
#include <cstdlib>

struct LKArgs {
    PyrView prev, cur;
    const double *pts, *disp0;
    int n, pyramid_levels, window, iterations;
    double eig_thr, eps, max_distance;
    double *out;
    uint8_t *status;
};

// -DLK_TRACE: per-phase tick totals (100 MHz wall clock, summed over waves) for scripts/probes/lk_trace.py; off in the product build
#ifdef LK_TRACE
__device__ unsigned long long g_lk_ticks[8];
#define LKT_BEGIN unsigned long long _t0 = wall_clock64()
#define LKT(k) do { const unsigned long long _t1 = wall_clock64(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_lk_ticks[k], _t1 - _t0); _t0 = _t1; } while (0)
extern "C" int slam_debug_lk_ticks(unsigned long long *out)
{
    unsigned long long z[8] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lk_ticks), sizeof z) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lk_ticks), z, sizeof z) == hipSuccess ? 0 : -1;
}
#else
#define LKT_BEGIN
#define LKT(k)
#endif

// occupancy target of the tracking kernels: 3 waves per SIMD (<= 168 VGPRs) for the 3- / 6-slot instantiations, 2 for the 9-slot one.
// Left to itself the register allocator lands on 167 or on 183 registers for the 6-slot kernel depending on unrelated details of the
// source (a wave per SIMD and 15 % of the kernel's speed); the bound makes it 168.
#define LK_OCC __attribute__((amdgpu_waves_per_eu(LK_MAXE == 9 ? 2 : 3, LK_MAXE == 9 ? 2 : 3)))
#define LK_PM 4
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

// Wave-wide sums of the iteration's TWO right-hand-side terms in the fixed butterfly order a[l] += a[l ^ m], m = 32, 16, 1, 2, 4, 8
// (restated in the CPU oracle, sum_order = 1), without touching the LDS crossbar: the reduction sits on the critical path of every LK
// iteration, and six dependent ds_bpermute round trips cost ~700 cycles.  m = 32: v_permlane32_swap (gfx950) exchanges the upper half
// of ay with the lower half of ax, so ONE add leaves ay(l) + ay(l + 32) in the lanes below 32 and ax(l - 32) + ax(l) in the lanes above
// -- from here on the two sums share every instruction; m = 16: v_permlane16_swap of the value with itself gives the rows {0, 0, 2, 2}
// and {1, 1, 3, 3}; m = 1, 2: DPP quad_perm; m = 4, 8: row_half_mirror / row_mirror (the partner group already holds one uniform value,
// so mirroring == xor).  Every lane of a half ends with the same bits (a + b == b + a); lanes 0 and 32 are read.  24 instructions for
// both sums (two separate butterflies with v_readlane row sums: 46).
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void wave_sum2(double &ay, double &ax)
{
    const auto l32 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(ay), (unsigned)__double2loint(ax), false, false);
    const auto h32 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(ay), (unsigned)__double2hiint(ax), false, false);
    double v = __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);      // l ^ 32
    const auto l16 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto h16 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);             // l ^ 16
    v = v + dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]  : l ^ 1
    v = v + dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]  : l ^ 2
    v = v + dpp_f64<0x141>(v);     // row_half_mirror      : l ^ 4 (quads are uniform)
    v = v + dpp_f64<0x140>(v);     // row_mirror           : l ^ 8 (octets are uniform)
    ay = readlane_f64(v, 0); ax = readlane_f64(v, 32);
}

// compute_spatial_gradient + svd2x2 + pinv2x2 (utils.jl:5-45)
__device__ __forceinline__ double spatial_gradient(const LevelView &v, int p0, int p1, Offs o, double Gi[4])
{
    const int y1 = p0 - o.up, y2 = p0 + o.down, x1 = p1 - o.left, x2 = p1 + o.right;
    // Images.boxdiff of the three integral images (lucas_kanade.jl:143-145): the 12 corners are ONE load -- lane l < 12 fetches
    // corner l & 3 of plane l >> 2 (corners left of / above the image read as 0, as boxdiff's index-0 rule) -- and the four terms
    // of each plane are combined in boxdiff's order from v_readlane broadcasts (12 loads held 24 registers through the set-up)
    double syy, sxx, syx;
    {
        const int lane = threadIdx.x & 63;
        const int pl = lane >> 2, cn = lane & 3;
        // (Iyy, Ixx, Iyx are consecutive planes of one allocation, pyramid.hip: plane(3..5, l); a per-lane select among the struct's
        //  pointer fields becomes an indexed read of the struct and moves it to scratch)
        const double *I = v.Iyy + (ptrdiff_t)pl * (v.Ixx - v.Iyy);
        const int yy = (cn & 2) ? y1 - 2 : y2 - 1, xx = (cn & 1) ? x1 - 2 : x2 - 1;
        const bool okc = lane < 12 && yy >= 0 && xx >= 0;
        const double val = okc ? I[(size_t)yy + (size_t)xx * v.P] : 0.0;
        double sm[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double sum = readlane_f64(val, 4 * k);
            sum -= readlane_f64(val, 4 * k + 1);
            sum -= readlane_f64(val, 4 * k + 2);
            sum += readlane_f64(val, 4 * k + 3);
            sm[k] = sum;
        }
        syy = sm[0]; sxx = sm[1]; syx = sm[2];
    }
    // M col-major: M11 = syy, M21 = syx, M12 = syx, M22 = sxx
    const double E = (syy + sxx) / 2, F = (syy - sxx) / 2, G = (syx + syx) / 2, Hh = (syx - syx) / 2;
    // M is symmetric: Hh = (syx - syx) / 2 = 0 and Q = sqrt(E * E + 0) = |E| exactly (sqrt(fl(x * x)) == |x| in binary floating point
    // unless x * x over- or underflows); the square-root sequence only runs outside that range
    double Q = fabs(E);
    if (!(Q > 1e-150 && Q < 1e150)) Q = sqrt(E * E + Hh * Hh);
    const double R = sqrt(F * F + G * G);
    const double sx = Q + R, sy = Q - R;
    const double S0 = sx, S1 = fabs(sy);
    const double tol = 1.4901161193847656e-08;
    const double cnt = (double)((long)(y2 - y1 + 1) * (long)(x2 - x1 + 1));
    if (E > 0 && sy > tol) {
        // Well-conditioned symmetric positive-definite case (every trackable window): both singular values pass
        // pinv2x2's threshold, so U*D*V' is the plain inverse; the closed form agrees with the reference's
        // atan/sincos construction to ~1e-16 relative and skips ~1500 instructions of f64 trigonometry.
        const double det = syy * sxx - syx * syx, id = 1.0 / det;
        Gi[0] = sxx * id; Gi[1] = -syx * id; Gi[2] = -syx * id; Gi[3] = syy * id;
        return fmin(S0, S1) / cnt;
    }
    // Ill-conditioned / indefinite window (rejected by eig_thr in every practical configuration, but the seam allows
    // eig_thr = 0): pinv2x2 of the symmetric M without the reference's atan/sincos construction.  M21 == M12, so the
    // SVD is the eigen-decomposition: eigenvalues E +- R with projectors (I +- [[F, G], [G, -F]] / R) / 2, and
    // pinv = sum over |eigenvalue| > tol of projector / eigenvalue (agrees with U*D*V' to 6e-16 relative).
    (void)Hh; (void)Q;
    const double l1 = E + R, l2 = E - R;
    const double c1 = fabs(l1) > tol ? 1.0 / l1 : 0.0, c2 = fabs(l2) > tol ? 1.0 / l2 : 0.0;
    if (R == 0.0) { Gi[0] = c1; Gi[1] = 0.0; Gi[2] = 0.0; Gi[3] = c1; }
    else {
        const double f = F / R, g = G / R, a = 0.5 * (c1 + c2), b = 0.5 * (c1 - c2);
        Gi[0] = a + b * f; Gi[1] = b * g; Gi[2] = b * g; Gi[3] = a - b * f;
    }
    return fmin(S0, S1) / cnt;
}

__device__ __forceinline__ double bilinear(const double *img, int H, int W, int P, double r, double c)
{
    int iy = (int)floor(r), ix = (int)floor(c);
    if (iy > H - 1) iy = H - 1;
    if (ix > W - 1) ix = W - 1;
    if (iy < 1) iy = 1;
    if (ix < 1) ix = 1;
    const double fy = r - iy, fx = c - ix;
    const double *p = img + (size_t)(iy - 1) + (size_t)(ix - 1) * P;
    const int dy = H > 1 ? 1 : 0; const size_t dx = W > 1 ? (size_t)P : 0;
    const double r0 = (1 - fx) * p[0] + fx * p[dx];
    const double r1 = (1 - fx) * p[dy] + fx * p[dy + dx];
    return (1 - fy) * r0 + fy * r1;
}

__device__ __forceinline__ bool lies_in(int H, int W, double a, double b)
{
    return 1.0 <= a && a <= (double)H && 1.0 <= b && b <= (double)W;
}

// The lane's share of the (2w+1)^2 template: element e = lane + 64*k of the reference's (q outer, p inner)
// enumeration.  Template samples, gradients and the element's window offsets depend only on the window
// geometry, so they are fetched once per geometry and stay in registers across the <= 30 iterations
// (no integer div/mod and no template loads inside the iteration).
// LK_MAXE slots per lane: the kernels are instantiated for 3 (window_size <= 6), 6 (<= 9, the default 19 x 19
// window) and 9 (<= 11) slots so that the register footprint follows the window; larger windows take the
// uncached path.
// (Rounds 1-2 kept the template samples in an LDS spill area -- 115 VGPRs, 4 waves per SIMD -- because every iteration waited for
// its footprint loads from global memory; since round 3 the iteration samples an LDS-resident target patch instead (below), the
// template lives in registers, and the kernel is VALU-bound: ~3 300 VALU instructions per point in the bench's launches, ~90 % busy.)
template <int LK_MAXE> struct Tmpl {
    double t0[LK_MAXE], t1[LK_MAXE], t2[LK_MAXE];
    int pq[LK_MAXE];               // window coordinates p | q << 16 of slot k (0 | 0 past the window)
    int ne, kmax;
};
struct __attribute__((packed, aligned(8))) D2 { double a, b; };   // two vertically adjacent samples, one 16-byte load

template <int LK_MAXE, bool TOL>
__device__ __forceinline__ void load_template(Tmpl<LK_MAXE> &T, const LevelView &first, int p0, int p1, Offs o)
{
    const int lane = threadIdx.x & 63, pitch = first.P;
    const int P = o.up + o.down + 1, Q = o.left + o.right + 1, NE = P * Q;
    T.ne = NE; T.kmax = (NE + 63) >> 6;
    // element e = lane + 64 k -> (p, q) = (e % P, e / P), advanced incrementally (one division per template; for the unclipped
    // windows of the usual sizes -- 2 w + 1 = 19 (window_size 9), 13, 23 -- by a constant: a multiply-shift, not the ~40-instruction
    // run-time division sequence every level visit used to start with)
    int pe, qe, sp, sq;
    if (P == 19) { pe = lane % 19; qe = lane / 19; sp = 64 % 19; sq = 64 / 19; }
    else if (P == 13) { pe = lane % 13; qe = lane / 13; sp = 64 % 13; sq = 64 / 13; }
    else if (P == 23) { pe = lane % 23; qe = lane / 23; sp = 64 % 23; sq = 64 / 23; }
    else { pe = lane % P; qe = lane / P; sp = 64 % P; sq = 64 / P; }
#pragma unroll
    for (int k = 0; k < LK_MAXE; k++) {
        const int e = lane + 64 * k;
        const bool in = e < NE;
        const int p = in ? pe : 0, q = in ? qe : 0;
        pe += sp; qe += sq;
        if (pe >= P) { pe -= P; qe++; }
        // (exec-masked loads: letting the slots past the window load element (0, 0) unconditionally -- straight-line code, all loads
        //  back to back -- was measured 13 % SLOWER.  Also measured and dropped: staging the window + 1 of the LAYER by LDS-DMA and
        //  evaluating the Scharr gradients in the kernel with the build's own operations (bit-identical, all tests green; 5 DMA
        //  instructions instead of 18 loads): 255 vs 183 us for 18 480 points -- the 54 LDS reads + ~350 f64 operations per level
        //  visit cost more than the loads they replace)
        const size_t a = (size_t)(p0 - o.up + p - 1) + (size_t)(p1 - o.left + q - 1) * pitch;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        if (in) { v0 = first.L[a]; v1 = first.Iy[a]; v2 = first.Ix[a]; }
        T.t0[k] = in ? v0 : 0.0; T.t1[k] = in ? v1 : 0.0; T.t2[k] = in ? v2 : 0.0;
        // tolerance mode: the element's offset inside the LDS patch (its window coordinates are not needed: the bilinear weights are
        // the estimate's own fractions there)
        T.pq[k] = TOL ? p + q * (2 * (LK_MAXE == 3 ? 6 : LK_MAXE == 6 ? 9 : 11) + 2 + 2 * LK_PM) : (p | (q << 16));
    }
}

// ---- the target footprint lives in LDS (north_star: "per-keypoint patches in LDS") -------------------------------------------
// An iteration samples the (2w + 2)^2 footprint of the target layer around the current estimate, and the estimate moves by a
// fraction of a pixel per iteration: the wave stages a PS x PS patch of the target layer (footprint + >= LK_PM pixels of
// margin on every side) ONCE per level visit -- 16-byte loads, consecutive lanes on consecutive row pairs of a column, issued
// together with the template loads -- and every iteration reads its four bilinear samples per element from LDS
// (ds_read2_b64 x 2) instead of waiting for two global round trips.  The patch is re-staged, centred on the current estimate,
// only when the footprint leaves it.  Same samples, same operations, same order: results are bit-identical to the global path.
template <int LK_MAXE> struct PatchGeom {
    static constexpr int WMAX = LK_MAXE == 3 ? 6 : LK_MAXE == 6 ? 9 : 11;        // largest window_size of the instantiation
    static constexpr int PS = 2 * WMAX + 2 + 2 * LK_PM;                           // 22 / 28 / 32 (even)
};
// top-left (cy, cx) 0-based is clipped so that the patch lies inside the image where the image is large enough; rows / columns
// past the image exist only when the image is smaller than the patch and are never sampled -- the caller samples footprints that
// lie inside the image.
template <int PS>
__device__ __forceinline__ void stage_patch(double *patch, const double *img, int H, int W, int pitch, int cy, int cx, int &pry, int &prx)
{
    const int lane = threadIdx.x & 63;
    const int hy = H - PS, hx = W - PS;
    pry = cy < 0 ? 0 : (cy > hy ? (hy > 0 ? hy : 0) : cy);
    prx = cx < 0 ? 0 : (cx > hx ? (hx > 0 ? hx : 0) : cx);
    // LDS-DMA (global_load_lds_dwordx4): one instruction moves CPI whole patch columns -- lane = (column c of the group, row pair ch)
    // in column-major order, so lane j lands at byte 16 j of the group: the lane-linear destination of the instruction IS the
    // patch's [column][row] layout; no staging registers, no ds_write pass; the per-lane source offset is the same for every group
    constexpr int CPC = PS / 2;                  // 16-byte chunks (row pairs) per column
    constexpr int CPI = 64 / CPC;                // columns per instruction
    constexpr int NIT = (PS + CPI - 1) / CPI;
    const int ch = lane % CPC, c = lane / CPC;
    const double *src = img + (size_t)pry + (size_t)prx * pitch + (2 * ch + (size_t)c * pitch);
    // images smaller than the patch (H < PS or W < PS: the coarsest levels, degenerate narrow frames): the chunks / columns past the
    // image are not requested, so the loads stay inside the layer plane (for odd H the chunk of row H - 1 also takes the pitch padding
    // row behind it); those patch cells are never sampled
    const bool rows_ok = pry + 2 * ch < H;
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        if (c < CPI && i * CPI + c < PS && rows_ok && prx + i * CPI + c < W)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)(i * CPI) * pitch),
                                             (__attribute__((address_space(3))) void *)(patch + i * CPI * PS), 16, 0, 0);
    }
}

// One pyramid level of optflow! for one point (lucas_kanade.jl:33-96).  All
// control flow is wave-uniform.  Returns the point's status.
// TOL (tolerance mode, selected for pyramids built in mode 3): the iteration's arithmetic is contracted -- bilinear() as two-operation
// lerps a + f (b - a) with the estimate's own fractions as weights (floor(r + dp) == floor(r) + dp, so the per-element fractions the
// reference forms differ from them by rounding only), fused multiply-adds in the two accumulations and the 2 x 2 solve: 9 Float64
// operations per window element instead of 26.  Positions agree with the sequential-order oracle to <= 1e-6 px (the summation order
// already differs by rounding); keypoint INDICES never pass through here (detect works on the raw frame).
template <int LK_MAXE, bool TOL>
__device__ __forceinline__ bool lk_level(const LevelView &first, const LevelView &second, int level,
                         double pty, double ptx, double &dy, double &dx,
                         int window, int iterations, double eig_thr, double eps)
{
    const int lane = threadIdx.x & 63;
    const int H = first.H, W = first.W, pitch = first.P;
    const double iscale = __hiloint2double((1023 - (level - 1)) << 20, 0);      // 2^-(level-1), built from its exponent: x * iscale == x / scale bit for bit (lucas_kanade.jl:38), no division sequence
    const int p0 = (int)floor(pty * iscale), p1 = (int)floor(ptx * iscale);
    const double pf0 = (double)p0, pf1 = (double)p1;
    LKT_BEGIN;
    const bool cached = (2 * window + 1) * (2 * window + 1) <= 64 * LK_MAXE;
    Tmpl<LK_MAXE> T;
    constexpr int PS = PatchGeom<LK_MAXE>::PS;
    __shared__ __attribute__((aligned(16))) double lds_patch[PS * PS];
    int pry = 0, prx = 0;
    const int pmarg = (PS - (2 * window + 2)) >> 1;          // margin on each side of a full footprint (>= LK_PM)
    Offs o = {0, 0, 0, 0};
    double Gi[4];
    double c0 = 0.0, c1 = 0.0;
    // it = -1 is the set-up of lucas_kanade.jl:44-52 (window geometry at the integer position, spatial gradient,
    // template); inside the iteration the same code runs again whenever the clipped window changes (:60-66).
    // One site for both keeps the kernel small enough for the instruction cache.
    for (int it = -1; it < iterations; it++) {
        double r0 = pf0, r1 = pf1;
        if (it >= 0) {
            const double f0 = dy + c0, f1 = dx + c1;
            r0 = pf0 + f0; r1 = pf1 + f1;
            if (!lies_in(H, W, r0, r1)) return false;
        }
        const Offs no = get_offsets(p0, p1, r0, r1, window, H, W);
        LKT(5);
        // where this pass samples the target (it = -1: where the first iteration will): the patch is staged at the level's
        // starting estimate -- requested BEFORE the template and the integral-image corners, one memory round trip for all three --
        // and again only when a later footprint leaves it.  Footprint strictly inside the image (always, except when
        // r0 + down == H or r1 + right == W exactly): element (p, q) reads rows iy0+dp-1, iy0+dp of columns ix0+dq-1, ix0+dq,
        // i.e. a fixed offset from the wave-uniform window origin; floor(r0 + dp) == floor(r0) + dp (and where rounding makes the
        // left side one larger, fy == 1 selects the same samples), so the results equal bilinear()'s bit for bit.
        const double e0 = it < 0 ? pf0 + dy : r0, e1 = it < 0 ? pf1 + dx : r1;
        const double fr0 = floor(e0), fr1 = floor(e1);
        const int iy0 = (int)fr0, ix0 = (int)fr1;
        const bool fast = cached && iy0 - no.up >= 1 && iy0 + no.down <= H - 1 && ix0 - no.left >= 1 && ix0 + no.right <= W - 1;
        const int fy0 = iy0 - no.up - 1, fx0 = ix0 - no.left - 1;                    // 0-based top-left sample of the footprint
        if (cached && (it < 0 || (fast && (fy0 < pry || iy0 + no.down > pry + PS - 1 || fx0 < prx || ix0 + no.right > prx + PS - 1))))
            stage_patch<PS>(lds_patch, second.L, H, W, pitch, iy0 - window - 1 - pmarg, ix0 - window - 1 - pmarg, pry, prx);
        if (it < 0 || no.up != o.up || no.down != o.down || no.left != o.left || no.right != o.right) {
            o = no;
            if (cached) load_template<LK_MAXE, TOL>(T, first, p0, p1, o);
            const double min_eig = spatial_gradient(first, p0, p1, o, Gi);
            LKT(1);
            if (min_eig < eig_thr) return false;
            LKT(2);
        }
        if (it < 0) continue;
        // prepare_linear_system (lucas_kanade.jl:159-173), wave order
        double ay = 0.0, ax = 0.0;
        if (fast) {
            // hipcc does not order these LDS reads behind an LDS-DMA still in flight (the patch re-staged in this pass: no other
            // load sits between the DMA and the reads; at the first iteration the wait for the integral-image corners has
            // already retired it) -- the wait is free whenever nothing is outstanding
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const double *pb = lds_patch + (fy0 - pry) + (fx0 - prx) * PS;
            // bilinear(): rows iy-1, iy of columns ix-1, ix -> two 16-byte LDS reads per element
            // (in two halves of LK_MAXE / 2 slots: fewer registers in flight)
            constexpr int HS = (LK_MAXE + 1) / 2;
            if (TOL) {
                const double fy = r0 - fr0, fx = r1 - fr1;
#pragma unroll
                for (int h0 = 0; h0 < LK_MAXE; h0 += HS) {
                    D2 c0v[HS], c1v[HS];
#pragma unroll
                    for (int j = 0; j < HS; j++) {
                        const int k = h0 + j;
                        if (k < LK_MAXE && k < T.kmax) { const double *ptr = pb + T.pq[k]; c0v[j] = *(const D2 *)ptr; c1v[j] = *(const D2 *)(ptr + PS); }
                    }
#pragma unroll
                    for (int j = 0; j < HS; j++) {
                        const int k = h0 + j;
                        if (k < LK_MAXE && k < T.kmax) {
                            const double t0 = __builtin_fma(fx, c1v[j].a - c0v[j].a, c0v[j].a);
                            const double t1 = __builtin_fma(fx, c1v[j].b - c0v[j].b, c0v[j].b);
                            const double dI = T.t0[k] - __builtin_fma(fy, t1 - t0, t0);
                            ay = __builtin_fma(dI, T.t1[k], ay); ax = __builtin_fma(dI, T.t2[k], ax);
                        }
                    }
                }
            } else
#pragma unroll
            for (int h0 = 0; h0 < LK_MAXE; h0 += HS) {
                D2 c0v[HS], c1v[HS];
#pragma unroll
                for (int j = 0; j < HS; j++) {
                    const int k = h0 + j;
                    if (k < LK_MAXE && k < T.kmax) {
                        const double *ptr = pb + ((T.pq[k] & 0xffff) + (T.pq[k] >> 16) * PS);
                        c0v[j] = *(const D2 *)ptr; c1v[j] = *(const D2 *)(ptr + PS);
                    }
                }
#pragma unroll
                for (int j = 0; j < HS; j++) {
                    const int k = h0 + j;
                    if (k < LK_MAXE && k < T.kmax) {
                        const double dp = (double)((T.pq[k] & 0xffff) - o.up), dq = (double)((T.pq[k] >> 16) - o.left);
                        const double fy = (r0 + dp) - (fr0 + dp), fx = (r1 + dq) - (fr1 + dq);
                        const double t0 = (1 - fx) * c0v[j].a + fx * c1v[j].a;
                        const double t1 = (1 - fx) * c0v[j].b + fx * c1v[j].b;
                        const double dI = T.t0[k] - ((1 - fy) * t0 + fy * t1);
                        ay += dI * T.t1[k];
                        ax += dI * T.t2[k];
                    }
                }
            }
        } else {
            const int P = o.up + o.down + 1, Q = o.left + o.right + 1, NE = P * Q;
            for (int e = lane; e < NE; e += 64) {
                const int p = e % P, q = e / P;
                const double r = r0 + (double)(p - o.up), c = r1 + (double)(q - o.left);
                const size_t a = (size_t)(p0 - o.up + p - 1) + (size_t)(p1 - o.left + q - 1) * pitch;
                const double dI = first.L[a] - bilinear(second.L, H, W, pitch, r, c);
                ay += dI * first.Iy[a];
                ax += dI * first.Ix[a];
            }
        }
        LKT(3);
        wave_sum2(ay, ax);
        const double fl0 = TOL ? __builtin_fma(Gi[0], ay, Gi[2] * ax) : Gi[0] * ay + Gi[2] * ax, fl1 = TOL ? __builtin_fma(Gi[1], ay, Gi[3] * ax) : Gi[1] * ay + Gi[3] * ax;
        if (fabs(fl0) < eps && fabs(fl1) < eps) break;
        c0 += fl0; c1 += fl1;
        if (!lies_in(H, W, r0 + fl0, r1 + fl1)) return false;
        LKT(4);
    }
    dy += c0; dx += c1;
    if (level > 1) { dy *= 2.0; dx *= 2.0; }
    return true;
}

// fb_tracking! for one point (tracker.jl:17-66): forward levels, level-1 backward
// pass from the forward result, consistency test.  Wave-uniform.
__device__ __forceinline__ LevelView shifted(const LevelView &v, size_t off)
{
    LevelView r = v;
    r.L += off; r.Iy += off; r.Ix += off; r.Iyy += off; r.Ixx += off; r.Iyx += off;
    return r;
}

// offP / offC: plane offset (doubles) of this point's image inside a pyramid batch (0 for single pyramids).
// The forward levels and the backward pass run through ONE inlined copy of lk_level (a loop over passes with the
// roles of the two pyramids swapped for the last one): the tracking kernels are a few KB instead of ~100 KB.
template <int LK_MAXE, bool TOL = false>
__device__ __forceinline__ bool fb_point(const PyrView &prev, const PyrView &cur, double py, double px, double dy, double dx,
                                         int pyramid_levels, int window, int iterations, double eig_thr, double eps,
                                         double max_distance, double &ny, double &nx, size_t offP = 0, size_t offC = 0)
{
    double qy = py, qx = px, cy = dy, cx = dx;
    for (int k = 0; k <= pyramid_levels + 1; k++) {
        const bool back = k == pyramid_levels + 1;
        const int level = back ? 1 : pyramid_levels + 1 - k;
        if (back) {
            ny = py + cy; nx = px + cx;                       // tracker.jl:41-42
            qy = ny; qx = nx;
            cy = -cy * 1.0; cx = -cx * 1.0;                   // back_displacement, scale = 1/2^0
        }
        // backward: pyramid_levels = 0, default eps 1e-2 (tracker.jl:34,51-57)
        const PyrView *pf = back ? &cur : &prev, *ps = back ? &prev : &cur;
        const bool ok = lk_level<LK_MAXE, TOL>(shifted(pf->lv[level - 1], back ? offC : offP), shifted(ps->lv[level - 1], back ? offP : offC),
                                          level, qy, qx, cy, cx, window, iterations, eig_thr, back ? 1e-2 : eps);
        if (!ok) return false;
    }
    const double b0 = ny + cy, b1 = nx + cx;
    const double d0 = py - b0, d1 = px - b1;
    return !(sqrt(d0 * d0 + d1 * d1) >= max_distance);
}

// Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b % 8) and every XCD has its own L2.
// Keypoint lists are spatially ordered (grid cells, row-major), and neighbouring points share cache lines of
// their windows: give each XCD one contiguous eighth of the list instead of every eighth point.
#define LK_XCDS 8
__device__ __forceinline__ int xcd_point(int n)
{
    const int per = (n + LK_XCDS - 1) / LK_XCDS;
    return (int)(blockIdx.x % LK_XCDS) * per + (int)(blockIdx.x / LK_XCDS);
}
static inline unsigned lk_grid(int n) { return (unsigned)((n + LK_XCDS - 1) / LK_XCDS) * LK_XCDS; }

template <int LK_MAXE>
__global__ __launch_bounds__(64) LK_OCC void k_fb_track(LKArgs A)
{
    const int i = xcd_point(A.n);
    if (i >= A.n) return;
    const double py = A.pts[2 * i], px = A.pts[2 * i + 1];
    const double dy = A.disp0 ? A.disp0[2 * i] : 0.0, dx = A.disp0 ? A.disp0[2 * i + 1] : 0.0;
    double ny = nan(""), nx = nan("");
    const bool ok = fb_point<LK_MAXE>(A.prev, A.cur, py, px, dy, dx, A.pyramid_levels, A.window, A.iterations, A.eig_thr, A.eps,
                             A.max_distance, ny, nx);
    if ((threadIdx.x & 63) == 0) {
        A.out[2 * i] = ny; A.out[2 * i + 1] = nx;
        A.status[i] = ok ? 1 : 0;
    }
}

// optical_flow_matching! on arrays (map_manager.jl:451-564) in ONE launch: a 3-D
// keypoint is first tracked with its projected prior on `levels3d` levels
// (:517-521); if that fails (or the point is 2-D) it is tracked without prior on
// `pyramid_levels` levels (:533-552).  Points are independent, so the reference's
// two fb_tracking! calls and the list surgery between them collapse into
// per-point control flow.
struct FlowArgs {
    LKArgs lk;
    const uint8_t *is3d; const double *proj; int levels3d;
    const int *img;            // batched call: image index of each point inside the pyramid batches (nullptr: single pyramids)
    size_t zs_from, zs_to;     // batch strides (doubles) of the from / to pyramids
};
template <int LK_MAXE>
__global__ __launch_bounds__(64) LK_OCC void k_flow_match(FlowArgs F)
{
    const LKArgs &A = F.lk;
    const int i = xcd_point(A.n);
    if (i >= A.n) return;
    LKT_BEGIN;
    const double py = A.pts[2 * i], px = A.pts[2 * i + 1];
    double ny = nan(""), nx = nan("");
    bool ok = false;
    const size_t zi = F.img ? (size_t)F.img[i] : 0;
    const size_t offP = zi * F.zs_from, offC = zi * F.zs_to;
    // attempt 0: 3-D point with its projected prior; attempt 1: no prior, all levels (one inlined fb_point)
    for (int att = F.is3d[i] ? 0 : 1; att < 2 && !ok; att++) {
        double dy = 0.0, dx = 0.0;
        if (att == 0) {
            const double scale = 1.0 / (double)(1 << F.levels3d);
            dy = scale * (F.proj[2 * i] - py); dx = scale * (F.proj[2 * i + 1] - px);                // map_manager.jl:494,504
        }
        ok = fb_point<LK_MAXE>(A.prev, A.cur, py, px, dy, dx, att == 0 ? F.levels3d : A.pyramid_levels, A.window, A.iterations, A.eig_thr, A.eps,
                               A.max_distance, ny, nx, offP, offC);
    }
    LKT(0);
    if ((threadIdx.x & 63) == 0) {
        A.out[2 * i] = ok ? ny : nan(""); A.out[2 * i + 1] = ok ? nx : nan("");
        A.status[i] = ok ? 1 : 0;
    }
}

// ---- optical_flow_matching! on a device-resident keypoint set (slam_kpset, kpset.hip) ------------------------------
// The per-point body is k_flow_match's; inputs come from, and results go to, the set's device arrays (no host lists):
//   temporal (map_manager.jl:451-564, stereo = false): a 3-D keypoint whose projection lies outside the image is skipped
//     (st = 2: kept as it is); otherwise 3-D prior attempt on `levels3d` levels, then the 2-D attempt; st = 1 + new
//     position in oyx, or st = 0 (lost: removed by the compaction that follows);
//   stereo (stereo = true): a 3-D keypoint projected outside the right image loses its observation (st = 0); a match must
//     pass maybe_stereo_update! (:579-590: |row - undistorted right row| <= epipolar_error) and is stored as
//     (left row, right column) in syx with stereo = 1; the keypoint itself always stays (st = 2).
// Per-stream parameters (32 doubles per stream, staged by the host): [0..15] Tcw of the TARGET camera (column-major),
// [16..19] fx fy cx cy, [20..23] k1 k2 p1 p2 of the target camera, [24..25] prior shift (y, x).
struct KpMatchArgs {
    PyrView from, to; size_t zs_from, zs_to;
    int pyramid_levels, levels3d, window, iterations; double eig_thr, eps, max_distance;
    double *yx, *oyx, *syx; const double *xyz; const uint8_t *is3d; uint8_t *st, *stereo; int cap;
    const int *work, *ntot;
    int prior;                 // 0: prior = the pixel itself, 1: projection of the map point (pose-based), 2: pixel + per-stream shift
    const double *par;
    int H, W, stereo_mode; double epipolar;
};
// undistort_pdn_point (camera.jl:111-125): normalised (y, x) -> pixel through the lens model
__device__ __forceinline__ void pdn_to_pixel(const double *cam, const double *dist, double ny, double nx, double &oy, double &ox)
{
    const double s0 = ny * ny, s1 = nx * nx, r2 = s0 + s1;
    const double rd = 1.0 + dist[0] * r2 + dist[1] * (r2 * r2);
    const double p = ny * nx;
    const double dtx = 2 * dist[2] * p + dist[3] * (r2 + 2 * s0);
    const double dty = dist[2] * (r2 + 2 * s1) + 2 * dist[3] * p;
    oy = (rd * ny + dty) * cam[1] + cam[3]; ox = (rd * nx + dtx) * cam[0] + cam[2];
}
template <int LK_MAXE, bool TOL>
__global__ __launch_bounds__(64) LK_OCC void k_kpset_match(KpMatchArgs M)
{
    const int ntot = M.ntot[0], per = (ntot + LK_XCDS - 1) / LK_XCDS;
    // each XCD a contiguous eighth of the (spatially ordered) list.  The grid is sized from the host's bound of the list lengths,
    // which is only a hint: a launch smaller than the lists walks them in several rounds (a keypoint that no wave visited would keep
    // the previous call's status and be compacted away or kept from stale data)
    for (int slotx = (int)(blockIdx.x / LK_XCDS); slotx < per; slotx += (int)(gridDim.x / LK_XCDS)) {
    const int idx = (int)(blockIdx.x % LK_XCDS) * per + slotx;
    if (idx >= ntot) break;
    const size_t q = (size_t)M.work[idx];
    const int s = (int)(q / M.cap);
    const double *P = M.par + 32 * (size_t)s;
    const double py = M.yx[2 * q], px = M.yx[2 * q + 1];
    const bool is3 = M.is3d[q] != 0;
    double pry = py, prx = px;
    if (is3) {
        if (M.prior == 1) {
            const double X0 = M.xyz[3 * q], X1 = M.xyz[3 * q + 1], X2 = M.xyz[3 * q + 2];
            const double cx = ((P[0] * X0 + P[4] * X1) + P[8] * X2) + P[12];
            const double cy = ((P[1] * X0 + P[5] * X1) + P[9] * X2) + P[13];
            const double cz = ((P[2] * X0 + P[6] * X1) + P[10] * X2) + P[14];
            pdn_to_pixel(P + 16, P + 20, cy / cz, cx / cz, pry, prx);                  // project_undistort (camera.jl:79-82)
        } else if (M.prior == 2) { pry = py + P[24]; prx = px + P[25]; }
    }
    const bool inside = 1 <= pry && pry <= (double)M.H && 1 <= prx && prx <= (double)M.W;   // in_image (camera.jl:90-92)
    const bool lane0 = (threadIdx.x & 63) == 0;
    if (is3 && !inside) {
        if (lane0) { M.st[q] = M.stereo_mode ? 0 : 2; if (M.stereo_mode) M.stereo[q] = 0; }
        continue;
    }
    const size_t offF = (size_t)s * M.zs_from, offT = (size_t)s * M.zs_to;
    double ny = nan(""), nx = nan("");
    bool ok = false;
    for (int att = is3 ? 0 : 1; att < 2 && !ok; att++) {
        double dy = 0.0, dx = 0.0;
        if (att == 0) {
            const double scale = 1.0 / (double)(1 << M.levels3d);
            dy = scale * (pry - py); dx = scale * (prx - px);                                        // map_manager.jl:494,504
        }
        ok = fb_point<LK_MAXE, TOL>(M.from, M.to, py, px, dy, dx, att == 0 ? M.levels3d : M.pyramid_levels, M.window, M.iterations, M.eig_thr, M.eps,
                               M.max_distance, ny, nx, offF, offT);
    }
    if (lane0 && !M.stereo_mode) {
        M.oyx[2 * q] = ok ? ny : nan(""); M.oyx[2 * q + 1] = ok ? nx : nan("");
        M.st[q] = ok ? 1 : 0;
    } else if (lane0) {
        if (ok) {                                                                                    // maybe_stereo_update!
            double uy, ux;
            pdn_to_pixel(P + 16, P + 20, (ny - P[19]) / P[17], (nx - P[18]) / P[16], uy, ux);      // undistort_point (camera.jl:98-103)
            if (fabs(py - uy) > M.epipolar) ok = false;
        }
        if (ok) { M.syx[2 * q] = py; M.syx[2 * q + 1] = nx; }
        M.stereo[q] = ok ? 1 : 0;
        M.st[q] = 2;
    }
    }
}

static int kpset_match(slam_ctx *ctx, slam_kpset *ks, const slam_pyr *from0, const slam_pyr *to0, const double *params,
                       int prior, int pyramid_levels, int levels3d, int window, int iterations, double eig_thr, double eps,
                       double max_distance, int stereo_mode, double epipolar, int n_bound)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && from0 != nullptr && to0 != nullptr);
    ARG_TRY(ctx, from0->batch_index == 0 && to0->batch_index == 0 && from0->batch_size >= ks->S && to0->batch_size >= ks->S);
    ARG_TRY(ctx, pyramid_levels >= 0 && levels3d >= 0 && window >= 0 && iterations >= 0 && prior >= 0 && prior <= 2 && (prior == 0 || params != nullptr));
    const int need = pyramid_levels > levels3d ? pyramid_levels : levels3d;
    if (!(from0->levels > need && to0->levels > need)) return slam_fail(ctx, SLAM_ERR_LAYERS, "Not enough layers in pyramids.");
    ARG_TRY(ctx, from0->H[0] == to0->H[0] && from0->W[0] == to0->W[0]);
    if (from0->target_only) return slam_fail(ctx, SLAM_ERR_ARG, "the source pyramid of a match was updated with SLAM_PYR_TARGET_ONLY: its gradient planes exist at level 0 only");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    KpMatchArgs M;
    M.from = from0->view; M.to = to0->view; M.zs_from = from0->zstride; M.zs_to = to0->zstride;
    M.pyramid_levels = pyramid_levels; M.levels3d = levels3d; M.window = window; M.iterations = iterations;
    M.eig_thr = eig_thr; M.eps = eps; M.max_distance = max_distance;
    M.yx = ks->yx; M.oyx = ks->oyx; M.syx = ks->syx; M.xyz = ks->xyz; M.is3d = ks->is3d; M.st = ks->st; M.stereo = ks->stereo; M.cap = ks->cap;
    M.work = ks->work; M.ntot = ks->ntot; M.prior = prior; M.H = from0->H[0]; M.W = from0->W[0]; M.stereo_mode = stereo_mode; M.epipolar = epipolar;
    std::vector<double> zero;
    if (!params) { zero.assign((size_t)ks->S * 32, 0.0); for (int s = 0; s < ks->S; s++) { zero[32 * s + 16] = zero[32 * s + 17] = 1.0; } params = zero.data(); }
    int rc = kpset_stage_params(ctx, ks, params, (size_t)ks->S * 32, &M.par);
    if (rc) return rc;
    rc = kpset_build_worklist(ctx, ks);
    if (rc) return rc;
    const int nmax = ks->S * ks->cap;
    int nb = n_bound > 0 && n_bound < nmax ? n_bound : nmax;
    // (measurement knob) cap the launch: every wave then walks several keypoints (the kernel loops) instead of one wave being dispatched per keypoint
    static const int grid_cap = [] { const char *v = getenv("SLAMHIP_LK_GRID"); return v ? atoi(v) : 0; }();
    if (grid_cap > 0 && nb > grid_cap) nb = grid_cap;
    { ProfScope span(ctx, "fb_track");
      const int ne = (2 * window + 1) * (2 * window + 1);
      // both pyramids built in tolerance mode (slam_pyr_update* mode 3): the contracted-arithmetic instantiation (positions <= 1e-6 px)
      static const bool no_tol_lk = getenv("SLAMHIP_NO_TOL_LK") != nullptr;
      const bool tol = from0->tol_planes && to0->tol_planes && !no_tol_lk;
      if (tol) {
          if (ne <= 192) hipLaunchKernelGGL((k_kpset_match<3, true>), dim3(lk_grid(nb)), dim3(64), 0, ctx->stream, M);
          else if (ne <= 384) hipLaunchKernelGGL((k_kpset_match<6, true>), dim3(lk_grid(nb)), dim3(64), 0, ctx->stream, M);
          else hipLaunchKernelGGL((k_kpset_match<9, true>), dim3(lk_grid(nb)), dim3(64), 0, ctx->stream, M);
      }
      else if (ne <= 192) hipLaunchKernelGGL((k_kpset_match<3, false>), dim3(lk_grid(nb)), dim3(64), 0, ctx->stream, M);
      else if (ne <= 384) hipLaunchKernelGGL((k_kpset_match<6, false>), dim3(lk_grid(nb)), dim3(64), 0, ctx->stream, M);
      else hipLaunchKernelGGL((k_kpset_match<9, false>), dim3(lk_grid(nb)), dim3(64), 0, ctx->stream, M); }
    HIP_TRY(ctx, hipGetLastError());
    return kpset_compact(ctx, ks, 0, nullptr);                   // lost keypoints (st = 0) leave the lists; stable
}

// optical_flow_matching!(map_manager, frame, from, to, false) for the S streams of the set (enqueue only).
// params: S x 32 doubles (layout above) or NULL (prior 0); n_bound: an upper bound of the number of live keypoints known to the
// host (sizes the launch; <= 0: S x cap)
extern "C" int slam_kpset_flow_match(slam_ctx *ctx, slam_kpset *ks, const slam_pyr *from0, const slam_pyr *to0, const double *params, int prior,
                                     int pyramid_levels, int pyramid_levels_3d, int window, int iterations, double eig_thr, double eps,
                                     double max_distance, int n_bound)
{
    return kpset_match(ctx, ks, from0, to0, params, prior, pyramid_levels, pyramid_levels_3d, window, iterations, eig_thr, eps, max_distance, 0, 0.0, n_bound);
}
// optical_flow_matching!(..., stereo = true): left0 -> right0; params describe the RIGHT camera
extern "C" int slam_kpset_stereo_match(slam_ctx *ctx, slam_kpset *ks, const slam_pyr *left0, const slam_pyr *right0, const double *params, int prior,
                                       int pyramid_levels, int pyramid_levels_3d, int window, int iterations, double eig_thr, double eps,
                                       double max_distance, double epipolar_error, int n_bound)
{
    return kpset_match(ctx, ks, left0, right0, params, prior, pyramid_levels, pyramid_levels_3d, window, iterations, eig_thr, eps, max_distance, 1, epipolar_error, n_bound);
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
#define LK_STAGE_MIN_POINTS 4096

// shared host path.  Keypoint lists are tiny (16-33 B per point): instead of staging
// them through HBM (H2D copy -> kernel -> D2H copy: two extra dependent DMA hops per
// call) the kernel reads its inputs from, and writes its results to, the context's
// pinned, device-mapped, coherent host block directly over PCIe.  One launch + one
// stream sync per call.
static int run_tracking(slam_ctx *ctx, const slam_pyr *prev, const slam_pyr *cur, const double *pts_yx, const double *aux_yx,
                        const uint8_t *is3d, int n, int pyramid_levels, int levels3d, int window, int iterations,
                        double eig_thr, double eps, double max_distance, double *out_yx, uint8_t *status, bool flow,
                        const int32_t *img_index = nullptr)
{
    const size_t pb = al256((size_t)n * 16), sb = al256((size_t)n), ib = al256((size_t)n * 4);
    const size_t in_b = 2 * pb + sb + ib, out_b = pb + sb;
    char *h, *d;
    int rc = slam_pinned(ctx, in_b + out_b, (void **)&h);
    if (rc) return rc;
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&d, h, 0));
    memcpy(h, pts_yx, (size_t)n * 16);
    if (aux_yx) memcpy(h + pb, aux_yx, (size_t)n * 16);
    if (is3d) memcpy(h + 2 * pb, is3d, (size_t)n);
    if (img_index) memcpy(h + 2 * pb + sb, img_index, (size_t)n * 4);
    // Large batches: tens of thousands of 8-byte reads and writes over PCIe (every wave starts with dependent
    // reads of its point and ends with three small stores) are slower than one DMA of the whole block each way.
    const bool staged = n >= LK_STAGE_MIN_POINTS;
    if (staged) {
        void *dbuf;
        rc = slam_scratch(ctx, in_b + out_b, &dbuf);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(dbuf, h, in_b, hipMemcpyHostToDevice, ctx->stream));
        d = (char *)dbuf;
    }
    FlowArgs F;
    LKArgs &A = F.lk;
    A.prev = prev->view; A.cur = cur->view;
    A.pts = (const double *)d; A.disp0 = aux_yx ? (const double *)(d + pb) : nullptr; A.n = n;
    A.pyramid_levels = pyramid_levels; A.window = window; A.iterations = iterations;
    A.eig_thr = eig_thr; A.eps = eps; A.max_distance = max_distance;
    A.out = (double *)(d + in_b); A.status = (uint8_t *)(d + in_b + pb);
    F.is3d = (const uint8_t *)(d + 2 * pb); F.proj = (const double *)(d + pb); F.levels3d = levels3d;
    F.img = img_index ? (const int *)(d + 2 * pb + sb) : nullptr; F.zs_from = prev->zstride; F.zs_to = cur->zstride;
    { ProfScope span(ctx, "fb_track");
      const int ne = (2 * window + 1) * (2 * window + 1);
      if (flow) {
          if (ne <= 192) hipLaunchKernelGGL(k_flow_match<3>, dim3(lk_grid(n)), dim3(64), 0, ctx->stream, F);
          else if (ne <= 384) hipLaunchKernelGGL(k_flow_match<6>, dim3(lk_grid(n)), dim3(64), 0, ctx->stream, F);
          else hipLaunchKernelGGL(k_flow_match<9>, dim3(lk_grid(n)), dim3(64), 0, ctx->stream, F);
      } else {
          if (ne <= 192) hipLaunchKernelGGL(k_fb_track<3>, dim3(lk_grid(n)), dim3(64), 0, ctx->stream, A);
          else if (ne <= 384) hipLaunchKernelGGL(k_fb_track<6>, dim3(lk_grid(n)), dim3(64), 0, ctx->stream, A);
          else hipLaunchKernelGGL(k_fb_track<9>, dim3(lk_grid(n)), dim3(64), 0, ctx->stream, A);
      } }
    HIP_TRY(ctx, hipGetLastError());
    if (staged) HIP_TRY(ctx, hipMemcpyAsync(h + in_b, d + in_b, out_b, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    memcpy(out_yx, h + in_b, (size_t)n * 16);
    memcpy(status, h + in_b + pb, (size_t)n);
    return SLAM_OK;
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
    if (prev->target_only) return slam_fail(ctx, SLAM_ERR_ARG, "the source pyramid of a match was updated with SLAM_PYR_TARGET_ONLY: its gradient planes exist at level 0 only");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return run_tracking(ctx, prev, cur, pts_yx, disp0_yx, nullptr, n, pyramid_levels, 0, window, iterations, eig_thr, eps,
                        max_distance, out_yx, status, false);
}

extern "C" int slam_flow_match(slam_ctx *ctx, const slam_pyr *from, const slam_pyr *to,
                               const double *pts_yx, const uint8_t *is_3d, const double *proj_yx, int n,
                               int pyramid_levels, int pyramid_levels_3d, int window, int iterations,
                               double eig_thr, double eps, double max_distance,
                               double *out_yx, uint8_t *status)
{
    ARG_TRY(ctx, ctx != nullptr && from != nullptr && to != nullptr);
    ARG_TRY(ctx, n >= 0 && pyramid_levels >= 0 && pyramid_levels_3d >= 0 && window >= 0 && iterations >= 0);
    if (n == 0) return SLAM_OK;
    ARG_TRY(ctx, pts_yx != nullptr && is_3d != nullptr && proj_yx != nullptr && out_yx != nullptr && status != nullptr);
    const int need = pyramid_levels > pyramid_levels_3d ? pyramid_levels : pyramid_levels_3d;
    if (!(from->levels > need && to->levels > need))
        return slam_fail(ctx, SLAM_ERR_LAYERS, "Not enough layers in pyramids.");
    ARG_TRY(ctx, from->H[0] == to->H[0] && from->W[0] == to->W[0]);
    if (from->target_only) return slam_fail(ctx, SLAM_ERR_ARG, "the source pyramid of a match was updated with SLAM_PYR_TARGET_ONLY: its gradient planes exist at level 0 only");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return run_tracking(ctx, from, to, pts_yx, proj_yx, is_3d, n, pyramid_levels, pyramid_levels_3d, window, iterations, eig_thr, eps,
                        max_distance, out_yx, status, true);
}

// optical_flow_matching! for S lock-stepped streams in one launch: point i belongs to stream img_index[i]; from0 / to0 are
// member 0 of two pyramid batches (slam_pyr_create_batch) of the same size.  Per point identical to slam_flow_match on that
// stream's own pyramids (points are independent); one launch and one stream sync for all streams.
extern "C" int slam_flow_match_batch(slam_ctx *ctx, const slam_pyr *from0, const slam_pyr *to0, int S, const int32_t *img_index,
                                     const double *pts_yx, const uint8_t *is_3d, const double *proj_yx, int n,
                                     int pyramid_levels, int pyramid_levels_3d, int window, int iterations,
                                     double eig_thr, double eps, double max_distance, double *out_yx, uint8_t *status)
{
    ARG_TRY(ctx, ctx != nullptr && from0 != nullptr && to0 != nullptr && S >= 1);
    ARG_TRY(ctx, from0->batch_index == 0 && to0->batch_index == 0 && from0->batch_size >= S && to0->batch_size >= S);
    ARG_TRY(ctx, n >= 0 && pyramid_levels >= 0 && pyramid_levels_3d >= 0 && window >= 0 && iterations >= 0);
    if (n == 0) return SLAM_OK;
    ARG_TRY(ctx, img_index != nullptr && pts_yx != nullptr && is_3d != nullptr && proj_yx != nullptr && out_yx != nullptr && status != nullptr);
    for (int i = 0; i < n; i++) if (img_index[i] < 0 || img_index[i] >= S) return slam_fail(ctx, SLAM_ERR_ARG, "slam_flow_match_batch: img_index[%d] = %d outside [0,%d)", i, img_index[i], S);
    const int need = pyramid_levels > pyramid_levels_3d ? pyramid_levels : pyramid_levels_3d;
    if (!(from0->levels > need && to0->levels > need))
        return slam_fail(ctx, SLAM_ERR_LAYERS, "Not enough layers in pyramids.");
    ARG_TRY(ctx, from0->H[0] == to0->H[0] && from0->W[0] == to0->W[0]);
    if (from0->target_only) return slam_fail(ctx, SLAM_ERR_ARG, "the source pyramid of a match was updated with SLAM_PYR_TARGET_ONLY: its gradient planes exist at level 0 only");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return run_tracking(ctx, from0, to0, pts_yx, proj_yx, is_3d, n, pyramid_levels, pyramid_levels_3d, window, iterations, eig_thr, eps,
                        max_distance, out_yx, status, true, img_index);
}

// slam_flow_match_batch followed by the list surgery of optical_flow_matching! (map_manager.jl:523-560: keypoints whose
// tracking failed are removed, the others take their new position), as a stable compaction on the host side of the
// call: kept_* hold the surviving keypoints in input order (kept_src[k] = their index in the input lists).  The
// arrays must have room for n entries.  status (n bytes, nullable) still reports every input keypoint.
extern "C" int slam_flow_match_batch_kept(slam_ctx *ctx, const slam_pyr *from0, const slam_pyr *to0, int S, const int32_t *img_index,
                                          const double *pts_yx, const uint8_t *is_3d, const double *proj_yx, int n,
                                          int pyramid_levels, int pyramid_levels_3d, int window, int iterations,
                                          double eig_thr, double eps, double max_distance,
                                          double *kept_yx, uint8_t *kept_is3d, int32_t *kept_img, int32_t *kept_src, int *n_kept, uint8_t *status)
{
    ARG_TRY(ctx, ctx != nullptr && n_kept != nullptr);
    *n_kept = 0;
    if (n == 0) return SLAM_OK;
    ARG_TRY(ctx, n > 0 && kept_yx != nullptr && kept_is3d != nullptr && kept_img != nullptr && kept_src != nullptr);
    // run_tracking leaves positions and status in the context's pinned block; compact straight out of it
    std::vector<uint8_t> st_local;
    uint8_t *st = status;
    if (!st) { st_local.resize((size_t)n); st = st_local.data(); }
    int rc = slam_flow_match_batch(ctx, from0, to0, S, img_index, pts_yx, is_3d, proj_yx, n, pyramid_levels, pyramid_levels_3d, window, iterations,
                                   eig_thr, eps, max_distance, kept_yx, st);
    if (rc) return rc;
    int k = 0;
    for (int i = 0; i < n; i++)
        if (st[i]) {
            kept_yx[2 * k] = kept_yx[2 * i]; kept_yx[2 * k + 1] = kept_yx[2 * i + 1];      // k <= i: in-place stable compaction
            kept_is3d[k] = is_3d[i]; kept_img[k] = img_index[i]; kept_src[k] = i;
            k++;
        }
    *n_kept = k;
    return SLAM_OK;
}
