// ba.hip -- local bundle adjustment and single-pose refinement on gfx950.
//
// Replaces bundle_adjustment! / _ba_detect_outliers! / pnp_bundle_adjustment of
// the reference (src/bundle_adjustment.jl:1-171) on the flat LocalBACache
// arrays (src/estimator.jl:16-40).  The Levenberg-Marquardt outer loop is
// LeastSquaresOptim's (trust-region radius update, step-quality test,
// diagonal clamping); the step itself is the EXACT solution of the damped
// normal equations, obtained by eliminating the map points (Schur complement)
// and factorising the 6P x 6P reduced camera system, instead of the
// reference's inexact LSMR solve on the full sparse system.
//
// Per LM iteration (all on device, fixed launch sequence, no host sync; every
// kernel early-outs once the device-side state says "converged"):
//   k_schur_groups   one workgroup per group of map points with the same first free observer: residuals, analytic
//                    2x6 / 2x3 Jacobians, V = Jl'Jl + D, V^-1, bl, W = Jp'Jl and the group's window of pose blocks
//                    -(W V^-1) W' (+ Jp'Jp, gradient, diag U), all in LDS
//   k_schur_reduce   S, g, diag(U) = fixed-order sums of the window partials (deterministic, no atomics)
//   k_band_solve     damped banded Cholesky of S in one workgroup, dp (wide systems: the tiled k_chol_* chain)
//   k_update_groups  per group: dl = V^-1 (bl - W' dp), trial parameters, trial and predicted residuals
//   k_control        rho, accept/reject, radius update, convergence (LeastSquaresOptim's rules)
//   (commit)         accept: the committed and the trial parameter buffers swap roles (LMState::cur, flipped by lm_decide); k_commit is
//                    the host-paced protocol's flip
// Fallback for systems the groups do not cover (block half-bandwidth > 20, a point with > 448 observations):
//   k_linearize, k_points, k_obs_factors, k_blocks (pair lists sorted by pose block), k_backsub, k_trial.
// Observations are re-ordered by map point at upload (map points by first free observer) so a point's observations
// and a group's points are contiguous.
#include "common.hpp"
#include <atomic>
#include <algorithm>
#include <type_traits>
#include <chrono>
#include <thread>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <cmath>

#define LM_MAX_DELTA 1e16
#define LM_MIN_DELTA 1e-16
#define LM_MIN_STEP_QUALITY 1e-3
#define LM_MIN_DIAGONAL 1e-6
#define LM_MAX_DIAGONAL 1e32
#define LM_DELTA0 10.0
#define LM_XTOL 1e-8
#define LM_FTOL 1e-8
#define SOLVE_MAX_N 1536   /* 6 * 256 key-frames */

struct LMState {
    double delta, decrease_factor, ssr, trial_ssr, pred_ssr, maxdx;
    double ssr_init, ssr_pass1, ssr_final;
    int converged, accept, iters, n_outliers, chol_fail, iters_pass1, iters_pass2;
    int cur;                     // which of the two parameter buffers is the committed one: an accepted step SWAPS them (lm_decide) -- no copy
                                 // kernel per iteration (k_commit cost the iteration a launch: ~5 us of its 127)
};


struct BADev {
    Cam cam;
    int P, M, O, n;              // n = 6P
    double *pose, *pose_t, *pts, *pts_t;
    const uint8_t *pconst;
    const double *pix;           // SoA: py[O], px[O]
    const int *opose, *opoint, *pt_start;
    uint8_t *outl, *hasp;
    double *f, *ft;              // AoS O x 2
    double *Jp, *Jl;             // AoS O x 12, O x 6 (a lane reads its observation's block contiguously)
    double *Vinv, *bl;           // SoA 6 x M, 3 x M
    double *T, *Wm;              // AoS O x 18 each
    const int2 *pairs; const int *blk_start; const int2 *blk_pq; int nblk;
    // pt_start / pt_id: observations are sorted by map point, the map points by (first free observing pose, id); opk = the
    // sorted position of an observation's point.  grp / fgrp / wpart: the point groups of k_schur_groups (below).
    const int *pt_id, *opk;
    const int4 *grp; const int *fgrp; int ngrp, whb, wstride;
    const int *fobs;             // the observations of free poses, grouped by map point in sorted order (pfs[M] entries)
    const int *pfs;              // pfs[k]: observations of free poses of the map points (sorted order) before point k, M + 1 entries (k_ba_window's chunks)
    const int *ohp; int sg_hp;   // ohp[i]: index of observation i among its group's observations of FREE poses (or -1): the phase 2-3 records (W, Jp, gradient:
                                 // 36 doubles) exist for those only -- the reference's window is 80 % observations of constant poses; sg_hp: room for that many
    int sg_ob, sg_sb;            // k_schur_groups' LDS layout: room for sg_ob observations / sg_sb points per group (SG_OB / SG_SB; a batch of small
                                 // windows sizes it to its largest group, so that several workgroups share a compute unit)
    double *wpart;
    double *S, *g, *udiag;       // reduce buffer views
    double *Swork, *dp, *dl;
    double *sc0, *sc1;           // [P][6] each: sin / cos of the angles of the poses in d.pose / in d.pose_t (batches: formed once per window and iteration by
                                 // k_pass_start_b / k_trial_poses_b and swapped with the parameter buffers; every point group used to form them for itself)
    double *part;                // reduction partials
    LMState *st;
};
// the committed parameters and the trial ones: d.pose / d.pts hold the committed set while st->cur == 0, d.pose_t / d.pts_t while it is 1
struct ParamBufs { double *pose, *pts, *pose_t, *pts_t; double *sc, *sc_t; };     // sc / sc_t: sin / cos of the committed / trial poses (batches)
__device__ __forceinline__ ParamBufs param_bufs(const BADev &d)
{
    const bool sw = d.st->cur != 0;
    return ParamBufs{sw ? d.pose_t : d.pose, sw ? d.pts_t : d.pts, sw ? d.pose : d.pose_t, sw ? d.pts : d.pts_t, sw ? d.sc1 : d.sc0, sw ? d.sc0 : d.sc1};
}


struct slam_ba {
    int device = 0;
    BADev d;
    void *arena = nullptr;       // one device allocation
    bool owns_arena = true;      // false: the arena is the calling context's scratch (slam_local_ba)
    double *reduce = nullptr;    // internal reduce buffer (single-GPU path)
    int *chol_flag = nullptr;    // device flag: a pivot was not positive
    double *linv = nullptr;      // inverses of the factored diagonal tiles, nbc x 32 x 32
    double *lfac = nullptr;      // finished factor tiles + forward-substituted rhs row, (n+1) x n
    int hb = 0;                  // block half-bandwidth of the reduced system: S_pq = 0 for |p - q| > hb
    int p0 = 0, pspan = 0;       // the banded solve runs on the poses p0 .. p0 + pspan - 1 = first .. last FREE pose: the constant poses outside
                                 // that span (the reference's window: <= 5 free key-frames + their constant observers, estimator.jl:327-331) have
                                 // identity blocks and dp = 0 -- each of them used to be a block column of the factorisation all the same
    bool grouped = false;        // the reduced system is built by k_schur_groups / k_schur_reduce (else: pair lists, k_blocks)
    int nparts = 0;              // partial sums k_control folds after a linearisation inside a build
    const double *zeroed = nullptr;   // reduce buffer whose out-of-band part is known to be zero
    double *band = nullptr;      // factor store of k_band_solve, P x ((hb + 1) x 36 + 8)
    double *xchg = nullptr;      // twisted factorisation: the trailing window one side hands to the other
    int epoch = 0;               // launch counter of k_band_solve (value of its hand-over flags)
    std::vector<int> perm;       // sorted position -> original observation index
    std::vector<int> pose_order; // solver's pose k = the caller's pose pose_order[k]; empty: the caller's order (see ba_pose_order)
    int nblocks_obs = 0, nblocks_pts = 0;
};

// ---------------------------------------------------------------------------------
// residual of one observation + analytic Jacobian (bundle_adjustment.jl:23-30;
// RotZYX = Rz(t1) Ry(t2) Rx(t3)).  Jp: 2x6 row-major, Jl: 2x3 row-major.
// sc = (sin, cos) of the three angles, tr = the translation: the group kernels form sc ONCE per pose and workgroup (sincos in Float64 is a few
// hundred instructions; an observation evaluated it three times) -- the same function of the same argument, so the same bits
__device__ __forceinline__ void pose_sincos(const double *pose, double sc[6])
{
    sincos(pose[0], &sc[0], &sc[1]); sincos(pose[1], &sc[2], &sc[3]); sincos(pose[2], &sc[4], &sc[5]);
}
__device__ __forceinline__ void obs_eval_sc(const double *sc, const double *tr, const double *X, double py, double px, const Cam &c,
                                            double r[2], double *Jp, double *Jl, double *depth)
{
    const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3], s3 = sc[4], c3 = sc[5];
    const double pose[6] = {0.0, 0.0, 0.0, tr[0], tr[1], tr[2]};
    const double R[9] = {c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3,
                         s1 * c2, s1 * s2 * s3 + c1 * c3, s1 * s2 * c3 - c1 * s3,
                         -s2, c2 * s3, c2 * c3};
    const double x = (R[0] * X[0] + R[1] * X[1] + R[2] * X[2]) + pose[3];
    const double y = (R[3] * X[0] + R[4] * X[1] + R[5] * X[2]) + pose[4];
    const double z = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2]) + pose[5];
    const double iz = 1.0 / z;
    r[0] = py - (c.fy * y * iz + c.cy);
    r[1] = px - (c.fx * x * iz + c.cx);
    if (depth) *depth = z;
    if (!Jl) return;
    const double dy[3] = {0.0, -c.fy * iz, c.fy * y * iz * iz};
    const double dx[3] = {-c.fx * iz, 0.0, c.fx * x * iz * iz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Jl[k] = dy[0] * R[k] + dy[1] * R[3 + k] + dy[2] * R[6 + k];
        Jl[3 + k] = dx[0] * R[k] + dx[1] * R[3 + k] + dx[2] * R[6 + k];
    }
    if (!Jp) return;
    const double d1[9] = {-s1 * c2, -s1 * s2 * s3 - c1 * c3, -s1 * s2 * c3 + c1 * s3,
                          c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3, 0, 0, 0};
    const double d2[9] = {-c1 * s2, c1 * c2 * s3, c1 * c2 * c3, -s1 * s2, s1 * c2 * s3, s1 * c2 * c3, -c2, -s2 * s3, -s2 * c3};
    const double d3[9] = {0, c1 * s2 * c3 + s1 * s3, -c1 * s2 * s3 + s1 * c3, 0, s1 * s2 * c3 - c1 * s3, -s1 * s2 * s3 - c1 * c3,
                          0, c2 * c3, -c2 * s3};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double *D = k == 0 ? d1 : (k == 1 ? d2 : d3);
        const double vx = D[0] * X[0] + D[1] * X[1] + D[2] * X[2];
        const double vy = D[3] * X[0] + D[4] * X[1] + D[5] * X[2];
        const double vz = D[6] * X[0] + D[7] * X[1] + D[8] * X[2];
        Jp[k] = dy[0] * vx + dy[1] * vy + dy[2] * vz;
        Jp[6 + k] = dx[0] * vx + dx[1] * vy + dx[2] * vz;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { Jp[3 + k] = dy[k]; Jp[9 + k] = dx[k]; }
}
__device__ __forceinline__ void obs_eval(const double *pose, const double *X, double py, double px, const Cam &c,
                                         double r[2], double *Jp, double *Jl, double *depth)
{
    double sc[6];
    pose_sincos(pose, sc);
    obs_eval_sc(sc, pose + 3, X, py, px, c, r, Jp, Jl, depth);
}

// deterministic block reduction (256 threads): wave butterfly, then wave order
// workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the outstanding global loads / stores: after a store
// that is a full memory round trip (k_schur_groups: its "barrier" phases were mostly the Jacobian / partial stores being acknowledged).
// For barriers that only hand LDS data (or nothing) between the threads of a workgroup.
__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ double block_sum_lds(double v, double *sh)       // block_sum with LDS-only barriers
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    lds_sync();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    lds_sync();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += sh[w];
    return t;
}
__device__ __forceinline__ double block_max_lds(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    lds_sync();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    lds_sync();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t = fmax(t, sh[w]);
    return t;
}
__device__ __forceinline__ double block_sum(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += sh[w];
    return t;
}
__device__ __forceinline__ double block_max(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t = fmax(t, sh[w]);
    return t;
}

// ---------------------------------------------------------------------------------
// An observation's records (Jp 12, Jl 6, T 18, W 18 doubles) are 16-byte aligned AoS blocks: move them as 16-byte vectors
// (half the memory instructions of scalar loads).
template <int N> __device__ __forceinline__ void ld_rec(const double *p, double *v)
{
    static_assert(N % 2 == 0, "even record length");
    const double2 *q = (const double2 *)p;
#pragma unroll
    for (int k = 0; k < N / 2; k++) { const double2 t = q[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
}
template <int N> __device__ __forceinline__ void st_rec(double *p, const double *v)
{
    static_assert(N % 2 == 0, "even record length");
    double2 *q = (double2 *)p;
#pragma unroll
    for (int k = 0; k < N / 2; k++) q[k] = make_double2(v[2 * k], v[2 * k + 1]);
}

template <bool STORE = true>          // STORE = false (batches): the cost only -- the grouped build evaluates every observation again and keeps what it needs
__device__ __forceinline__ void linearize_body(const BADev &d, int ignore_outliers, int respect_done)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    __shared__ double sh[4];
    if (respect_done && d.st->converged) return;
    const int i = blockIdx.x * 256 + threadIdx.x, O = d.O;
    double ss = 0.0;
    if (i < O) {
        const int p = d.opose[i], j = d.opoint[i];
        const bool active = !(ignore_outliers && d.outl[i]);
        const bool hp = active && !d.pconst[p];
        double r[2] = {0.0, 0.0}, Jp[12], Jl[6];
#pragma unroll
        for (int k = 0; k < 12; k++) Jp[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) Jl[k] = 0.0;
        if (active) {
            const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
            double pose[6];
#pragma unroll
            for (int k = 0; k < 6; k++) pose[k] = pb.pose[6 * p + k];
            if (STORE) obs_eval(pose, X, d.pix[i], d.pix[O + i], d.cam, r, Jp, Jl, nullptr);
            else obs_eval(pose, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
            if (!hp) {
#pragma unroll
                for (int k = 0; k < 12; k++) Jp[k] = 0.0;
            }
        }
        if (STORE) {
            d.hasp[i] = hp ? 1 : 0;
            st_rec<2>(d.f + 2 * (size_t)i, r);
            st_rec<12>(d.Jp + (size_t)i * 12, Jp);
            st_rec<6>(d.Jl + (size_t)i * 6, Jl);
        }
        ss = r[0] * r[0] + r[1] * r[1];
    }
    const double t = block_sum(ss, sh);
    if (threadIdx.x == 0) d.part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void k_linearize(BADev d, int ignore_outliers, int respect_done) { linearize_body(d, ignore_outliers, respect_done); }

__device__ __forceinline__ void inv3_sym(const double V[6], double I[6])
{
    const double a = V[0], b = V[1], c = V[2], dd = V[3], e = V[4], f = V[5];
    const double A = dd * f - e * e, B = c * e - b * f, C = b * e - c * dd;
    const double det = a * A + b * B + c * C, id = 1.0 / det;
    I[0] = A * id; I[1] = B * id; I[2] = C * id;
    I[3] = (a * f - c * c) * id; I[4] = (b * c - a * e) * id; I[5] = (a * dd - b * b) * id;
}

__global__ __launch_bounds__(256) void k_points(BADev d, double inv_delta_host, int use_state)
{
    if (use_state && d.st->converged) return;
    const int k = blockIdx.x * 256 + threadIdx.x, M = d.M;
    if (k >= M) return;
    const int j = d.pt_id[k];
    const double inv_delta = use_state ? 1.0 / d.st->delta : inv_delta_host;
    double V[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    const int t0 = d.pt_start[k], t1 = d.pt_start[k + 1];
    for (int i = t0; i < t1; i++) {
        double jl[6], ff[2];
        ld_rec<6>(d.Jl + (size_t)i * 6, jl); ld_rec<2>(d.f + 2 * (size_t)i, ff);
        const double f0 = ff[0], f1 = ff[1];
        V[0] += jl[0] * jl[0] + jl[3] * jl[3]; V[1] += jl[0] * jl[1] + jl[3] * jl[4]; V[2] += jl[0] * jl[2] + jl[3] * jl[5];
        V[3] += jl[1] * jl[1] + jl[4] * jl[4]; V[4] += jl[1] * jl[2] + jl[4] * jl[5]; V[5] += jl[2] * jl[2] + jl[5] * jl[5];
#pragma unroll
        for (int k = 0; k < 3; k++) bl[k] += jl[k] * f0 + jl[3 + k] * f1;
    }
    V[0] += fmin(fmax(V[0], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    V[3] += fmin(fmax(V[3], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    V[5] += fmin(fmax(V[5], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    double Vi[6];
    inv3_sym(V, Vi);
#pragma unroll
    for (int k = 0; k < 6; k++) d.Vinv[(size_t)k * M + j] = Vi[k];
#pragma unroll
    for (int k = 0; k < 3; k++) d.bl[(size_t)k * M + j] = bl[k];
}

// per observation: W = Jp'Jl (6x3), T = W V^-1 of its point
__global__ __launch_bounds__(256) void k_obs_factors(BADev d, int use_state)
{
    if (use_state && d.st->converged) return;
    const int i = blockIdx.x * 256 + threadIdx.x, M = d.M;
    if (i >= d.O) return;
    if (d.pconst[d.opose[i]]) return;                 // never referenced by a pair list
    double *Wo = d.Wm + (size_t)i * 18, *To = d.T + (size_t)i * 18;
    if (!d.hasp[i]) {                                  // ignored outlier: its pair-list entries must contribute nothing
#pragma unroll
        for (int k = 0; k < 18; k++) { Wo[k] = 0.0; To[k] = 0.0; }
        return;
    }
    const int j = d.opoint[i];
    double jp[12], jl[6], Vi[6], wv[18], tv[18];
    ld_rec<12>(d.Jp + (size_t)i * 12, jp); ld_rec<6>(d.Jl + (size_t)i * 6, jl);
#pragma unroll
    for (int k = 0; k < 6; k++) Vi[k] = d.Vinv[(size_t)k * M + j];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const double w0 = jp[a] * jl[0] + jp[6 + a] * jl[3];
        const double w1 = jp[a] * jl[1] + jp[6 + a] * jl[4];
        const double w2 = jp[a] * jl[2] + jp[6 + a] * jl[5];
        wv[3 * a] = w0; wv[3 * a + 1] = w1; wv[3 * a + 2] = w2;
        tv[3 * a] = w0 * Vi[0] + w1 * Vi[1] + w2 * Vi[2];
        tv[3 * a + 1] = w0 * Vi[1] + w1 * Vi[3] + w2 * Vi[4];
        tv[3 * a + 2] = w0 * Vi[2] + w1 * Vi[4] + w2 * Vi[5];
    }
    st_rec<18>(Wo, wv); st_rec<18>(To, tv);
}

// One wave per non-zero upper block (p <= q) of the reduced camera system.
__global__ __launch_bounds__(256) void k_blocks(BADev d, int use_state)
{
    __shared__ double s_red[4][48];
    if (use_state && d.st->converged) return;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The blocks are sorted by (p, q) and the blocks of
    // one pose row p read the same T records (and neighbouring rows the same W records): give every XCD one contiguous
    // eighth of the block list, so that those re-reads are L2 hits instead of 8 separate fetches of the T / W arrays.
    const int per = (d.nblk + 7) / 8;
    const int b = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (b >= d.nblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, M = d.M, n = d.n;
    const int2 pq = d.blk_pq[b];
    const int e0 = d.blk_start[b], e1 = d.blk_start[b + 1];
    double acc[36], gg[6], ud[6];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) { gg[k] = 0.0; ud[k] = 0.0; }
    for (int e = e0 + tid; e < e1; e += 256) {
        const int2 tt = d.pairs[e];
        double T[18], W2[18];
        ld_rec<18>(d.T + (size_t)tt.x * 18, T); ld_rec<18>(d.Wm + (size_t)tt.y * 18, W2);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++)
                acc[a + 6 * c] -= T[3 * a] * W2[3 * c] + T[3 * a + 1] * W2[3 * c + 1] + T[3 * a + 2] * W2[3 * c + 2];
        if (tt.x == tt.y) {
            const int i = tt.x, j = d.opoint[i];
            double jp[12];
#pragma unroll
            for (int k = 0; k < 12; k++) jp[k] = d.Jp[(size_t)i * 12 + k];
            const double f0 = d.f[2 * (size_t)i], f1 = d.f[2 * (size_t)i + 1];
            const double b0 = d.bl[j], b1 = d.bl[(size_t)M + j], b2 = d.bl[(size_t)2 * M + j];
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int c = 0; c < 6; c++) acc[a + 6 * c] += jp[a] * jp[c] + jp[6 + a] * jp[6 + c];
                gg[a] += (jp[a] * f0 + jp[6 + a] * f1) - (T[3 * a] * b0 + T[3 * a + 1] * b1 + T[3 * a + 2] * b2);
                ud[a] += jp[a] * jp[a] + jp[6 + a] * jp[6 + a];
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] += __shfl_xor(acc[k], m);
        if (pq.x == pq.y) {
#pragma unroll
            for (int k = 0; k < 6; k++) { gg[k] += __shfl_xor(gg[k], m); ud[k] += __shfl_xor(ud[k], m); }
        }
    }
    // fold the 4 waves in a fixed order through LDS (deterministic)
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 36; k++) s_red[wv][k] = acc[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { s_red[wv][36 + k] = gg[k]; s_red[wv][42 + k] = ud[k]; }
    }
    __syncthreads();
    if (tid < 36) {
        const int a = tid % 6, c = tid / 6;
        const double v = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
        d.S[(size_t)(6 * pq.x + a) + (size_t)(6 * pq.y + c) * n] = v;
        if (pq.x != pq.y) d.S[(size_t)(6 * pq.y + c) + (size_t)(6 * pq.x + a) * n] = v;
    }
    if (pq.x == pq.y && tid >= 64 && tid < 70) {
        const int a = tid - 64;
        d.g[6 * pq.x + a] = ((s_red[0][36 + a] + s_red[1][36 + a]) + s_red[2][36 + a]) + s_red[3][36 + a];
        d.udiag[6 * pq.x + a] = ((s_red[0][42 + a] + s_red[1][42 + a]) + s_red[2][42 + a]) + s_red[3][42 + a];
    }
}

// ---- the reduced camera system of a windowed problem, built point group by point group ---------------------------------
// A map point seen by free poses f .. f + hb only touches the (hb + 1) x (hb + 1) window of 6 x 6 blocks that starts at its first
// free observer f.  The map points are sorted by f; one 512-thread workgroup takes a group of <= SG_SB points with the same f
// (<= SG_OB observations, contiguous) and does, without leaving LDS, what k_linearize + k_points + k_obs_factors + k_blocks do
// through HBM (T / W records: 29 MB written, 160 MB gathered per iteration at O = 1e5):
//   phase 0  thread = observation: residual + Jacobians (stored for k_backsub / k_trial), Jl'Jl and Jl'f into LDS
//   phase 1  thread = point: V = sum Jl'Jl + D, V^-1, bl (fixed order)
//   phase 2  thread = observation: W = Jp'Jl, Jp, Jp'f - W V^-1 bl into LDS
//   phase 3  thread = (window block (a, b), row r): sum over the group's points of -(W_a V^-1) W_b' (+ Jp'Jp on the diagonal);
//            thread = (window slot a, r): gradient and diagonal of U
// and writes the window as a partial (wstride doubles per group).  k_schur_reduce adds the partials of every band block in a
// fixed order (f ascending, groups ascending): deterministic, no atomics.  Inactive observations (ignored outliers, constant
// poses) have no window slot and contribute nothing, as in the pair lists.
#define SG_T 512
#define SG_OB 448
#define SG_SB 56
// byte offset of s_dg = end of the phase 0-2 arrays, or of the fold buffers of phase 3 that overlay them (whichever is larger)
__host__ __device__ __forceinline__ size_t sg_w_doubles(int ob, int hp) { const size_t a = (size_t)ob * 9, b = (size_t)hp * 18; return ((a > b ? a : b) + 1) & ~(size_t)1; }
__host__ __device__ __forceinline__ size_t sg_dg_off(int whb, int ob, int sb, int nthreads, int hp)
{
    const int hbw = whb + 1, nwin = hbw * (hbw + 1) / 2, LPS = (nwin + 63) & ~63, NS = nthreads / LPS;
    const size_t lay = ((sg_w_doubles(ob, hp) + (size_t)hp * 18 + (size_t)sb * 10 + 8) * 8 + (size_t)sb * hbw * 2 + (size_t)nwin * 2 + 15) & ~(size_t)15;
    const size_t fold = NS >= 1 ? ((((size_t)(NS - 1) * nwin * 36 + (size_t)(NS - 1) * hbw * 6 * 7) * 8 + 15) & ~(size_t)15) : 0;
    return lay > fold ? lay : fold;
}
static size_t sg_lds_bytes(int whb, int P, int ob = SG_OB, int sb = SG_SB, int nthreads = 512, int hp = SG_OB)
{
    return sg_dg_off(whb, ob, sb, nthreads, hp) + (size_t)(whb + 1) * 36 * 8 + (size_t)P * 6 * 8 + 16;
}

// sum over NS adjacent lanes (NS a power of two, uniform): DPP moves up to 16 lanes -- a ds_bpermute butterfly of the 36 block
// entries costs more LDS issue slots than the block products themselves
template <int CTRL> __device__ __forceinline__ double sg_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sg_fold(double v, int NS)
{
    if (NS >= 2) v = v + sg_dpp<0xB1>(v);       // quad_perm [1,0,3,2]
    if (NS >= 4) v = v + sg_dpp<0x4E>(v);       // quad_perm [2,3,0,1]
    if (NS >= 8) v = v + sg_dpp<0x141>(v);      // row_half_mirror (quads are uniform by now)
    if (NS >= 16) v = v + sg_dpp<0x140>(v);     // row_mirror (octets are uniform)
    if (NS >= 32) v = v + __shfl_xor(v, 16);
    if (NS >= 64) v = v + __shfl_xor(v, 32);
    return v;
}

#ifdef SG_TRACE
#define SG_CLK_DECL long long sg_clk[12]; const long long sg_t0 = clock64()
#define SG_CLK(k) sg_clk[k] = clock64() - sg_t0
#define SG_DUMP() do { if (threadIdx.x == 0 && blockIdx.x == (gridDim.y > 1 ? 10 : 100) && blockIdx.y == (gridDim.y > 1 ? 5 : 0) && d.st->iters == 3) printf("schur group: npts %d nobs %d | init %lld ph0 %lld bar %lld ph1 %lld ph2 %lld ph3 %lld bar %lld fold %lld tail %lld cycles\n", npts, nobs, sg_clk[0], sg_clk[1] - sg_clk[0], sg_clk[2] - sg_clk[1], sg_clk[3] - sg_clk[2], sg_clk[4] - sg_clk[3], sg_clk[8] - sg_clk[4], sg_clk[9] - sg_clk[8], sg_clk[5] - sg_clk[9], sg_clk[6] - sg_clk[5]); } while (0)
#define SGM_DUMP() do { if (threadIdx.x == 0 && blockIdx.x == 10 && blockIdx.y == 5 && d.st->iters == 3) printf("schur group (mfma): npts %d nobs %d | init %lld ph0 %lld bar %lld ph1 %lld ph2a %lld ph2x %lld ph2b %lld mfma+out %lld cycles\n", npts, nobs, sg_clk[0], sg_clk[1] - sg_clk[0], sg_clk[2] - sg_clk[1], sg_clk[3] - sg_clk[2], sg_clk[4] - sg_clk[3], sg_clk[5] - sg_clk[4], sg_clk[6] - sg_clk[5], sg_clk[7] - sg_clk[6]); } while (0)
#else
#define SG_CLK_DECL
#define SG_CLK(k)
#define SG_DUMP()
#define SGM_DUMP()
#endif
// the fold buffers of phase 3 (the partial blocks and slot rows of the subsets 1 .. NS - 1) overlay everything below s_dg
static bool sg_fold_fits(int whb)
{
    const int hbw = whb + 1, nwin = hbw * (hbw + 1) / 2, LPS = (nwin + 63) & ~63, NS = SG_T / LPS;
    const size_t dg_off = (((size_t)SG_OB * 36 + (size_t)SG_SB * 10 + 8) * 8 + (size_t)SG_SB * hbw * 2 + (size_t)nwin * 2 + 15) & ~(size_t)15;
    return NS >= 1 && ((size_t)(NS - 1) * nwin * 36 + (size_t)(NS - 1) * hbw * 6 * 7) * 8 <= dg_off;
}

template <int TT>      // threads per workgroup: SG_T, or 256 for a batch of windows whose groups all have <= 256 observations (two to three workgroups per compute unit)
__device__ __forceinline__ void schur_groups_body(const BADev &d, double inv_delta_host, int ignore_outliers, int use_state)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    extern __shared__ __attribute__((aligned(16))) double sg_lds[];
    SG_CLK_DECL;
    if (use_state && d.st->converged) return;
    const int tid = threadIdx.x, M = d.M, O = d.O;
    const int4 G = d.grp[blockIdx.x];                       // first point, first observation, f | points << 16, observations
    const int k0 = G.x, o0 = G.y, f = G.z & 0xffff, npts = G.z >> 16, nobs = G.w;
    const int hbw = d.whb + 1, nwin = hbw * (hbw + 1) / 2;
    const int OBc = d.sg_ob, SBc = d.sg_sb;        // layout capacities (SG_OB / SG_SB, or the largest group of a batch of small windows)
    const int HPc = d.sg_hp;                       // room for that many observations of free poses (the only ones with W / Jp / gradient records)
    double *s_W = sg_lds;                          // phases 0-1: [OBc][9] = Jl'Jl (6), Jl'f (3) of every observation; phases 2-3: [HPc][18] W = Jp'Jl
    double *s_Jp = s_W + sg_w_doubles(OBc, HPc);   // [HPc][12]
    double *s_g = s_Jp + HPc * 12;                 // [HPc][6]
    double *s_pt = s_g + HPc * 6;                  // [SBc][10]   V^-1 (6), bl (3), pad
    double *s_red = s_pt + SBc * 10;               // [8]
    short *s_slot = (short *)(s_red + 8);          // [SBc][hbw]  observation (index in the group) of point x in window slot y, or -1
    unsigned char *s_ab = (unsigned char *)(s_slot + SBc * hbw);   // [nwin][2]
    double *s_dg = sg_lds + (sg_dg_off(d.whb, OBc, SBc, TT, HPc) >> 3);     // [hbw][36] Jp'Jp per window slot
    double *s_sc = s_dg + hbw * 36;                                    // [P][6] sin / cos of every pose's angles (pose_sincos)
    for (int p = tid; p < d.P; p += TT) pose_sincos(pb.pose + 6 * p, s_sc + 6 * p);
    for (int x = tid; x < npts * hbw; x += TT) s_slot[x] = -1;
    for (int w = tid; w < nwin; w += TT) {
        int a = 0, r = w;
        while (r >= hbw - a) { r -= hbw - a; a++; }
        s_ab[2 * w] = (unsigned char)a; s_ab[2 * w + 1] = (unsigned char)(a + r);
    }
    lds_sync();
    SG_CLK(0);
    // ---- phase 0
    double r2[2] = {0.0, 0.0}, Jp[12], Jl[6];
#pragma unroll
    for (int k = 0; k < 12; k++) Jp[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) Jl[k] = 0.0;
    int pl = 0, hpi = -1;
    if (tid < nobs) {
        const int i = o0 + tid;
        const int p = d.opose[i], j = d.opoint[i];
        pl = d.opk[i] - k0;
        hpi = d.ohp[i];
        const bool active = !(ignore_outliers && d.outl[i]);
        const bool hp = active && !d.pconst[p];
        if (active) {
            const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
            double sc[6], tr[3];
#pragma unroll
            for (int k = 0; k < 6; k++) sc[k] = s_sc[6 * p + k];
#pragma unroll
            for (int k = 0; k < 3; k++) tr[k] = pb.pose[6 * p + 3 + k];
            obs_eval_sc(sc, tr, X, d.pix[i], d.pix[O + i], d.cam, r2, Jp, Jl, nullptr);
            if (!hp) {
#pragma unroll
                for (int k = 0; k < 12; k++) Jp[k] = 0.0;
            }
        }
        d.hasp[i] = hp ? 1 : 0;
        st_rec<2>(d.f + 2 * (size_t)i, r2);
        if (hp) st_rec<12>(d.Jp + (size_t)i * 12, Jp);     // (k_update_groups takes zeros where hasp is clear: the reference's window is 80 % observations of constant poses)
        st_rec<6>(d.Jl + (size_t)i * 6, Jl);
        if (hp) s_slot[pl * hbw + (p - f)] = (short)hpi; else hpi = -1;
        double *v = s_W + tid * 9;
        v[0] = Jl[0] * Jl[0] + Jl[3] * Jl[3]; v[1] = Jl[0] * Jl[1] + Jl[3] * Jl[4]; v[2] = Jl[0] * Jl[2] + Jl[3] * Jl[5];
        v[3] = Jl[1] * Jl[1] + Jl[4] * Jl[4]; v[4] = Jl[1] * Jl[2] + Jl[4] * Jl[5]; v[5] = Jl[2] * Jl[2] + Jl[5] * Jl[5];
#pragma unroll
        for (int k = 0; k < 3; k++) v[6 + k] = Jl[k] * r2[0] + Jl[3 + k] * r2[1];
    }
    SG_CLK(1);
    lds_sync();
    SG_CLK(2);
    // ---- phase 1
    if (tid < npts) {
        const int k = k0 + tid, j = d.pt_id[k];
        const double inv_delta = use_state ? 1.0 / d.st->delta : inv_delta_host;
        double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int t0 = d.pt_start[k] - o0, t1 = d.pt_start[k + 1] - o0;
        for (int t = t0; t < t1; t++) {
#pragma unroll
            for (int c = 0; c < 9; c++) V[c] += s_W[t * 9 + c];
        }
        V[0] += fmin(fmax(V[0], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        V[3] += fmin(fmax(V[3], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        V[5] += fmin(fmax(V[5], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        double Vi[6];
        inv3_sym(V, Vi);
#pragma unroll
        for (int c = 0; c < 6; c++) { d.Vinv[(size_t)c * M + j] = Vi[c]; s_pt[tid * 10 + c] = Vi[c]; }
#pragma unroll
        for (int c = 0; c < 3; c++) { d.bl[(size_t)c * M + j] = V[6 + c]; s_pt[tid * 10 + 6 + c] = V[6 + c]; }
    }
    lds_sync();
    SG_CLK(3);
    // ---- phase 2
    if (hpi >= 0) {                                        // (observations of constant poses and ignored outliers have no records: nothing reads them)
        double Vi[6], bl[3];
#pragma unroll
        for (int c = 0; c < 6; c++) Vi[c] = s_pt[pl * 10 + c];
#pragma unroll
        for (int c = 0; c < 3; c++) bl[c] = s_pt[pl * 10 + 6 + c];
        const double vb0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
        const double vb1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
        const double vb2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double w0 = Jp[a] * Jl[0] + Jp[6 + a] * Jl[3];
            const double w1 = Jp[a] * Jl[1] + Jp[6 + a] * Jl[4];
            const double w2 = Jp[a] * Jl[2] + Jp[6 + a] * Jl[5];
            s_W[hpi * 18 + 3 * a] = w0; s_W[hpi * 18 + 3 * a + 1] = w1; s_W[hpi * 18 + 3 * a + 2] = w2;
            s_g[hpi * 6 + a] = (Jp[a] * r2[0] + Jp[6 + a] * r2[1]) - (w0 * vb0 + w1 * vb1 + w2 * vb2);
        }
#pragma unroll
        for (int k = 0; k < 12; k++) s_Jp[hpi * 12 + k] = Jp[k];
    }
    lds_sync();
    SG_CLK(4);
    double *out = d.wpart + (size_t)blockIdx.x * d.wstride;
    // ---- phase 3: the window blocks.  A wave (a run of LPS lanes) is one point subset: its lanes are the blocks, all on the same
    //      point at the same time -- V^-1 and the slot row are broadcast reads, the W_a rows are shared by up to hb + 1 lanes, the
    //      W_b rows of neighbouring lanes are neighbouring records (conflict-free b128 reads).  The whole 6 x 6 block stays in
    //      registers (42 LDS doubles per 162 fused multiply-adds).  The first 6 (hb + 1) lanes of a subset then take one row of
    //      Jp'Jp and one gradient entry of a window slot each.  The subsets are folded through LDS in subset order by subset 0.
    {
        const int LPS = (nwin + 63) & ~63, NS = TT / LPS;                          // 8 subsets for hb <= 9, 4 up to 14, 2 up to 20
        const int sub = tid / LPS, w = tid - sub * LPS;
        const bool live = sub < NS && w < nwin, xl = sub < NS && w < hbw * 6;      // (LPS = 192 leaves 128 threads over: they are no subset)
        const int a = s_ab[live ? 2 * w : 0], b = s_ab[live ? 2 * w + 1 : 1];
        const int a2 = xl ? w / 6 : 0, r2 = w - 6 * (w / 6);
        double acc[36], ex[7];
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++) ex[k] = 0.0;
        if (live)
            for (int x = sub; x < npts; x += NS) {
                const int ta = s_slot[x * hbw + a], tb = s_slot[x * hbw + b];
                if (ta < 0 || tb < 0) continue;
                double Vi[6], Wa[18], Wb[18], T[18];
                ld_rec<6>(s_pt + x * 10, Vi); ld_rec<18>(s_W + ta * 18, Wa); ld_rec<18>(s_W + tb * 18, Wb);
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    T[3 * r] = fma(Wa[3 * r + 2], Vi[2], fma(Wa[3 * r + 1], Vi[1], Wa[3 * r] * Vi[0]));
                    T[3 * r + 1] = fma(Wa[3 * r + 2], Vi[4], fma(Wa[3 * r + 1], Vi[3], Wa[3 * r] * Vi[1]));
                    T[3 * r + 2] = fma(Wa[3 * r + 2], Vi[5], fma(Wa[3 * r + 1], Vi[4], Wa[3 * r] * Vi[2]));
                }
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 6; c++)
                        acc[6 * r + c] = fma(-T[3 * r + 2], Wb[3 * c + 2], fma(-T[3 * r + 1], Wb[3 * c + 1], fma(-T[3 * r], Wb[3 * c], acc[6 * r + c])));
            }
        if (xl)
            for (int x = sub; x < npts; x += NS) {
                const int ta = s_slot[x * hbw + a2];
                if (ta < 0) continue;
                double J[12];
                ld_rec<12>(s_Jp + ta * 12, J);
                const double j0 = r2 == 0 ? J[0] : r2 == 1 ? J[1] : r2 == 2 ? J[2] : r2 == 3 ? J[3] : r2 == 4 ? J[4] : J[5];
                const double j1 = r2 == 0 ? J[6] : r2 == 1 ? J[7] : r2 == 2 ? J[8] : r2 == 3 ? J[9] : r2 == 4 ? J[10] : J[11];
#pragma unroll
                for (int c = 0; c < 6; c++) ex[c] = fma(j1, J[6 + c], fma(j0, J[c], ex[c]));
                ex[6] += s_g[ta * 6 + r2];
            }
        SG_CLK(8);
        lds_sync();                                           // every read of W / Jp / g / V^-1 / the slots is done: the region becomes the fold buffer
        SG_CLK(9);
        double *fold = sg_lds, *efold = sg_lds + (size_t)(NS - 1) * nwin * 36;
        if (sub >= 1) {
            if (live) st_rec<36>(fold + ((size_t)(sub - 1) * nwin + w) * 36, acc);
            if (xl) {
#pragma unroll
                for (int k = 0; k < 7; k++) efold[((size_t)(sub - 1) * hbw * 6 + w) * 7 + k] = ex[k];
            }
        }
        lds_sync();
        if (sub == 0) {
            if (live)
                for (int q = 0; q < NS - 1; q++) {
                    double o[36];
                    ld_rec<36>(fold + ((size_t)q * nwin + w) * 36, o);
#pragma unroll
                    for (int k = 0; k < 36; k++) acc[k] += o[k];
                }
            if (xl) {
                for (int q = 0; q < NS - 1; q++) {
#pragma unroll
                    for (int k = 0; k < 7; k++) ex[k] += efold[((size_t)q * hbw * 6 + w) * 7 + k];
                }
#pragma unroll
                for (int c = 0; c < 6; c++) s_dg[a2 * 36 + r2 * 6 + c] = ex[c];
                const double ud = r2 == 0 ? ex[0] : r2 == 1 ? ex[1] : r2 == 2 ? ex[2] : r2 == 3 ? ex[3] : r2 == 4 ? ex[4] : ex[5];
                out[nwin * 36 + a2 * 12 + r2] = ex[6]; out[nwin * 36 + a2 * 12 + 6 + r2] = ud;
            }
        }
        lds_sync();
        if (live && sub == 0) {
            if (a == b) {
#pragma unroll
                for (int k = 0; k < 36; k++) acc[k] += s_dg[a * 36 + k];
            }
            st_rec<36>(out + w * 36, acc);
        }
    }
    SG_CLK(5);
    const double t = block_sum_lds(r2[0] * r2[0] + r2[1] * r2[1], s_red);
    if (tid == 0) d.part[blockIdx.x] = t;
    SG_CLK(6);
    SG_DUMP();
}
__global__ __launch_bounds__(SG_T) void k_schur_groups(BADev d, double inv_delta_host, int ignore_outliers, int use_state) { schur_groups_body<SG_T>(d, inv_delta_host, ignore_outliers, use_state); }

// ---- the same build for a BATCH of windows, with the Schur products on the matrix cores (round 6) -------------------------------------------
// For a group of map points the window blocks are  S_ab -= sum_x W_xa V_x^-1 W_xb'  over every pair a <= b of window slots.  With the Cholesky
// factor V_x^-1 = L_x L_x' and Y_xa = W_xa L_x (6 x 3) this is  -(Y Y')  for the matrix Y whose rows are (slot, pose parameter) and whose columns are
// (point, coordinate): a symmetric rank-k update with k = 3 x points -- the one place of the path that IS a dense contraction.  It runs as
// v_mfma_f64_16x16x4_f64 tiles (A[i][k] from lane i + 16 k, B[k][j] from lane j + 16 k, D[4 r + lane / 16][lane % 16] in accumulator r:
// scripts/ubench/mfma_f64_layout.hip), upper-triangular tiles dealt to the waves, no partial blocks to fold.  The peak of the matrix cores in
// Float64 equals the vector peak on this chip (64 cycles per 2048 multiply-adds); what the instruction removes is the issue and LDS traffic of
// the vector form (42 LDS doubles per 162 multiply-adds and lane, a quarter of them re-forming W V^-1 in every lane of a slot row) and the
// fold of four partial block sets through LDS: phase 3 + fold 30 k -> ~5 k cycles of a group's 68 k, and the workgroup needs 47 instead of
// 80 KB of LDS (three per compute unit).  Association differs from the vector kernel (Y Y' instead of (W V^-1) W'): results to rounding.
// LDS: one region R, used in turn as [ob][9] products Jl'Jl / Jl'f, as [hp][18] records (Jp, gradient) of the free-pose observations, and as
// the matrix Y -- element (row, k) at ((k >> 1) RP + row) 2 + (k & 1): the 32 lanes of a half-wave read 256 contiguous bytes.
typedef double sgm_d4 __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ int sgm_rp(int whb) { return 16 * ((6 * (whb + 1) + 15) / 16); }
__host__ __device__ __forceinline__ size_t sgm_r_doubles(int whb, int ob, int sb, int hp)
{
    const size_t a = (size_t)ob * 9, b = (size_t)hp * 18, c = (size_t)((3 * sb + 3) & ~3) * sgm_rp(whb);
    return ((a > b ? (a > c ? a : c) : (b > c ? b : c)) + 1) & ~(size_t)1;
}
static size_t sgm_lds_bytes(int whb, int P, int ob, int sb, int hp)
{
    return (sgm_r_doubles(whb, ob, sb, hp) + (size_t)sb * 16 + 8 + (size_t)(whb + 1) * 36 + (size_t)P * 6) * 8 + (size_t)sb * (whb + 1) * 2 + 16;
}
template <int TT>
__device__ __forceinline__ void schur_groups_mfma_body(const BADev &d, int ignore_outliers)
{
    extern __shared__ __attribute__((aligned(16))) double sg_lds[];
    SG_CLK_DECL;
    if (d.st->converged) return;
    const ParamBufs pb = param_bufs(d);
    const int tid = threadIdx.x, M = d.M, O = d.O, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (the wave index in a scalar register: the tile loop branches on it)
    constexpr int NW = TT / 64;
    const int4 G = d.grp[blockIdx.x];                       // first point, first observation, f | points << 16, observations
    const int k0 = __builtin_amdgcn_readfirstlane(G.x), o0 = __builtin_amdgcn_readfirstlane(G.y), f = __builtin_amdgcn_readfirstlane(G.z & 0xffff),
              npts = __builtin_amdgcn_readfirstlane(G.z >> 16), nobs = __builtin_amdgcn_readfirstlane(G.w);      // (uniform by construction; told to the compiler: scalar loop counters)
    const int hbw = d.whb + 1, nwin = hbw * (hbw + 1) / 2, RP = sgm_rp(d.whb), nrow = 6 * hbw;
    const int OBc = d.sg_ob, SBc = d.sg_sb, HPc = d.sg_hp;
    double *s_R = sg_lds;
    double *s_pt = s_R + sgm_r_doubles(d.whb, OBc, SBc, HPc);   // [SBc][16]  V^-1 (6), bl (3), L (6: l00 l10 l20 l11 l21 l22)
    double *s_dg = s_pt + SBc * 16 + 8;                          // [hbw][36]  Jp'Jp per window slot
    double *s_sc = s_dg + hbw * 36;                              // [P][6]     sin / cos of every pose's angles
    short *s_slot = (short *)(s_sc + 6 * d.P);                   // [SBc][hbw] record of point x in window slot y, or -1
    // the observation's scalars are requested BEFORE the set-up work below (their latency hides behind the sin / cos of the poses)
    int i = 0, p = 0, j = 0, pl = 0, hpi = -1; bool active = false, hp = false; double py = 0.0, px = 0.0;
    if (tid < nobs) {
        i = o0 + tid; p = d.opose[i]; j = d.opoint[i]; pl = d.opk[i] - k0; hpi = d.ohp[i];
        active = !(ignore_outliers && d.outl[i]); hp = active && !d.pconst[p];
        py = d.pix[i]; px = d.pix[O + i];
    }
    for (int a = tid; a < 6 * d.P; a += TT) s_sc[a] = pb.sc[a];      // (formed once per window: k_pass_start_b / k_trial_poses_b)
    for (int x = tid; x < npts * hbw; x += TT) s_slot[x] = -1;
    lds_sync();
    SG_CLK(0);
    // ---- phase 0: residual + Jacobians of the observation, Jl'Jl / Jl'f -> R
    double r2[2] = {0.0, 0.0}, Jp[12], Jl[6];
#pragma unroll
    for (int k = 0; k < 12; k++) Jp[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) Jl[k] = 0.0;
    if (tid < nobs) {
        if (active) {
            const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
            double sc[6], tr[3];
#pragma unroll
            for (int k = 0; k < 6; k++) sc[k] = s_sc[6 * p + k];
#pragma unroll
            for (int k = 0; k < 3; k++) tr[k] = pb.pose[6 * p + 3 + k];
            obs_eval_sc(sc, tr, X, py, px, d.cam, r2, Jp, Jl, nullptr);
            if (!hp) {
#pragma unroll
                for (int k = 0; k < 12; k++) Jp[k] = 0.0;
            }
        }
        // (nothing of the evaluation is stored: k_update_groups_b<.., RECOMP> forms it again -- 160 bytes per observation not written here, not read there)
        if (hp) s_slot[pl * hbw + (p - f)] = (short)hpi; else hpi = -1;
        double *v = s_R + tid * 9;
        v[0] = Jl[0] * Jl[0] + Jl[3] * Jl[3]; v[1] = Jl[0] * Jl[1] + Jl[3] * Jl[4]; v[2] = Jl[0] * Jl[2] + Jl[3] * Jl[5];
        v[3] = Jl[1] * Jl[1] + Jl[4] * Jl[4]; v[4] = Jl[1] * Jl[2] + Jl[4] * Jl[5]; v[5] = Jl[2] * Jl[2] + Jl[5] * Jl[5];
#pragma unroll
        for (int k = 0; k < 3; k++) v[6 + k] = Jl[k] * r2[0] + Jl[3 + k] * r2[1];
    }
    SG_CLK(1);
    lds_sync();
    SG_CLK(2);
    // ---- phase 1: thread = map point: V = sum + D, V^-1, its Cholesky factor, bl
    if (tid < npts) {
        const int k = k0 + tid, jj = d.pt_id[k];
        const double inv_delta = 1.0 / d.st->delta;
        double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int t0 = d.pt_start[k] - o0, t1 = d.pt_start[k + 1] - o0;
        for (int t = t0; t < t1; t++) {
#pragma unroll
            for (int c = 0; c < 9; c++) V[c] += s_R[t * 9 + c];
        }
        V[0] += fmin(fmax(V[0], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        V[3] += fmin(fmax(V[3], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        V[5] += fmin(fmax(V[5], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        double Vi[6];
        inv3_sym(V, Vi);
#pragma unroll
        for (int c = 0; c < 6; c++) { d.Vinv[(size_t)c * M + jj] = Vi[c]; s_pt[tid * 16 + c] = Vi[c]; }
#pragma unroll
        for (int c = 0; c < 3; c++) { d.bl[(size_t)c * M + jj] = V[6 + c]; s_pt[tid * 16 + 6 + c] = V[6 + c]; }
        // V^-1 = L L' (V^-1 is positive definite with V; a breakdown gives NaN, which the solve reports as a failed factorisation)
        const double l00 = sqrt(Vi[0]), l10 = Vi[1] / l00, l20 = Vi[2] / l00;
        const double l11 = sqrt(Vi[3] - l10 * l10), l21 = (Vi[4] - l20 * l10) / l11;
        const double l22 = sqrt(Vi[5] - l20 * l20 - l21 * l21);
        double *L = s_pt + tid * 16 + 9;
        L[0] = l00; L[1] = l10; L[2] = l20; L[3] = l11; L[4] = l21; L[5] = l22;
    }
    lds_sync();
    SG_CLK(3);
    // ---- phase 2a: free-pose observations: W = Jp'Jl, the gradient term, Y = W L (kept in registers), record (Jp, gradient) -> R
    double Y[18];
#pragma unroll
    for (int k = 0; k < 18; k++) Y[k] = 0.0;
    if (hpi >= 0) {
        double Vi[6], bl[3], L[6];
#pragma unroll
        for (int c = 0; c < 6; c++) Vi[c] = s_pt[pl * 16 + c];
#pragma unroll
        for (int c = 0; c < 3; c++) bl[c] = s_pt[pl * 16 + 6 + c];
#pragma unroll
        for (int c = 0; c < 6; c++) L[c] = s_pt[pl * 16 + 9 + c];
        const double vb0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
        const double vb1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
        const double vb2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
        double *E = s_R + hpi * 18;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double w0 = Jp[a] * Jl[0] + Jp[6 + a] * Jl[3];
            const double w1 = Jp[a] * Jl[1] + Jp[6 + a] * Jl[4];
            const double w2 = Jp[a] * Jl[2] + Jp[6 + a] * Jl[5];
            Y[3 * a] = w0 * L[0] + w1 * L[1] + w2 * L[2]; Y[3 * a + 1] = w1 * L[3] + w2 * L[4]; Y[3 * a + 2] = w2 * L[5];
            E[12 + a] = (Jp[a] * r2[0] + Jp[6 + a] * r2[1]) - (w0 * vb0 + w1 * vb1 + w2 * vb2);
        }
#pragma unroll
        for (int k = 0; k < 12; k++) E[k] = Jp[k];
    }
    lds_sync();
    SG_CLK(4);
    // ---- phase 2x: Jp'Jp, the gradient and diag U per window slot: lane = (slot a2, row rr, point class xq); no partials through LDS --
    //      the XQ lanes of a task sit next to each other and are folded by DPP
    double *out = d.wpart + (size_t)blockIdx.x * d.wstride;
    {
        const int TPW = (nrow + NW - 1) / NW, XQ = TPW <= 16 ? 4 : TPW <= 32 ? 2 : 1;
        const int task = wv * TPW + lane / XQ, xq = lane - (lane / XQ) * XQ;
        const bool xl = lane / XQ < TPW && task < nrow;
        const int a2 = xl ? task / 6 : 0, rr = task - 6 * (task / 6);
        double ex[7];
#pragma unroll
        for (int k = 0; k < 7; k++) ex[k] = 0.0;
        if (xl)
            for (int x = xq; x < npts; x += XQ) {
                const int ta = s_slot[x * hbw + a2];
                if (ta < 0) continue;
                double J[12];
                ld_rec<12>(s_R + ta * 18, J);
                const double j0 = rr == 0 ? J[0] : rr == 1 ? J[1] : rr == 2 ? J[2] : rr == 3 ? J[3] : rr == 4 ? J[4] : J[5];
                const double j1 = rr == 0 ? J[6] : rr == 1 ? J[7] : rr == 2 ? J[8] : rr == 3 ? J[9] : rr == 4 ? J[10] : J[11];
#pragma unroll
                for (int c = 0; c < 6; c++) ex[c] = fma(j1, J[6 + c], fma(j0, J[c], ex[c]));
                ex[6] += s_R[ta * 18 + 12 + rr];
            }
#pragma unroll
        for (int k = 0; k < 7; k++) ex[k] = sg_fold(ex[k], XQ);
        if (xl && xq == 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) s_dg[a2 * 36 + rr * 6 + c] = ex[c];
            const double ud = rr == 0 ? ex[0] : rr == 1 ? ex[1] : rr == 2 ? ex[2] : rr == 3 ? ex[3] : rr == 4 ? ex[4] : ex[5];
            out[nwin * 36 + a2 * 12 + rr] = ex[6]; out[nwin * 36 + a2 * 12 + 6 + rr] = ud;
        }
    }
    lds_sync();                                               // every record is read: R becomes the matrix Y
    SG_CLK(5);
    // ---- phase 2b: Y -> R; cells nobody owns (a slot without observation, the padding rows and columns) are zeroed by whoever comes by
    const int K3 = 3 * npts, K4 = (K3 + 3) & ~3;
    if (hpi >= 0) {
        const int rb = 6 * (p - f);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int k = 3 * pl + c;
            double *col = s_R + ((size_t)(k >> 1) * RP + rb) * 2 + (k & 1);
#pragma unroll
            for (int r = 0; r < 6; r++) col[2 * r] = Y[3 * r + c];
        }
    }
    for (int c = tid; c < npts * hbw; c += TT)
        if (s_slot[c] < 0) {
            const int x = c / hbw, rb = 6 * (c - x * hbw);
            for (int cc = 0; cc < 3; cc++) {
                const int k = 3 * x + cc;
                double *col = s_R + ((size_t)(k >> 1) * RP + rb) * 2 + (k & 1);
#pragma unroll
                for (int r = 0; r < 6; r++) col[2 * r] = 0.0;
            }
        }
    for (int c = tid; c < (RP - nrow) * K4; c += TT) { const int k = c / (RP - nrow), row = nrow + (c - k * (RP - nrow)); s_R[((size_t)(k >> 1) * RP + row) * 2 + (k & 1)] = 0.0; }
    for (int c = tid; c < (K4 - K3) * nrow; c += TT) { const int k = K3 + c / nrow, row = c - (c / nrow) * nrow; s_R[((size_t)(k >> 1) * RP + row) * 2 + (k & 1)] = 0.0; }
    lds_sync();
    SG_CLK(6);
    // ---- phase 3: -(Y Y') on the matrix cores: the upper-triangular 16 x 16 tiles dealt to the waves, three tiles (three independent
    //      accumulator chains) at a time
    {
        const int NT = RP / 16, ntiles = NT * (NT + 1) / 2, q = lane >> 4, c16 = lane & 15, nks = K4 >> 2;
        const size_t lane_off = ((size_t)(q >> 1) * RP + c16) * 2 + (q & 1), kstep = (size_t)4 * RP;
        auto emit = [&](int I, int J, const sgm_d4 &acc) {
            const int col = J * 16 + c16, b = col / 6, cc = col - 6 * b;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = I * 16 + 4 * r + q, a = row / 6, rr = row - 6 * a;
                if (a >= hbw || b >= hbw || a > b) continue;
                const int w = a * hbw - a * (a - 1) / 2 + (b - a);
                if (a < b) out[w * 36 + rr * 6 + cc] = -acc[r];
                else {
                    const double v = s_dg[a * 36 + rr * 6 + cc] - acc[r];
                    out[w * 36 + rr * 6 + cc] = v;
                    if (I < J) out[w * 36 + cc * 6 + rr] = v;      // a diagonal block cut by a tile boundary: its mirror half lies in a tile below the diagonal, which nobody computes
                }
            }
        };
        for (int t0 = wv; t0 < ntiles; t0 += 3 * NW) {
            int I[3], J[3]; bool ok[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                int t = t0 + u * NW; ok[u] = t < ntiles; if (!ok[u]) t = 0;
                int ii = 0; while (t >= NT - ii) { t -= NT - ii; ii++; }
                I[u] = ii; J[u] = ii + t;
            }
            const double *pa0 = s_R + lane_off + (size_t)I[0] * 32, *pb0 = s_R + lane_off + (size_t)J[0] * 32;
            const double *pa1 = s_R + lane_off + (size_t)I[1] * 32, *pb1 = s_R + lane_off + (size_t)J[1] * 32;
            const double *pa2 = s_R + lane_off + (size_t)I[2] * 32, *pb2 = s_R + lane_off + (size_t)J[2] * 32;
            sgm_d4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0, c2 = c0;
            for (int ks = 0; ks < nks; ks++) {
                const size_t o = (size_t)ks * kstep;
                const double a0 = pa0[o], b0 = pb0[o], a1 = pa1[o], b1 = pb1[o], a2m = pa2[o], b2 = pb2[o];
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2m, b2, c2, 0, 0, 0);
            }
            emit(I[0], J[0], c0);
            if (ok[1]) emit(I[1], J[1], c1);
            if (ok[2]) emit(I[2], J[2], c2);
        }
    }
    SG_CLK(7);
    SGM_DUMP();
}

// S, g, diag(U) from the window partials: thread = (band block (p, p + dq), entry) / (pose, gradient or diagonal entry)
__device__ __forceinline__ void schur_reduce_body(const BADev &d, int use_state)
{
    if (use_state && d.st->converged) return;
    const int idx = blockIdx.x * 256 + threadIdx.x, P = d.P, n = d.n;
    const int hbw = d.whb + 1, nwin = hbw * (hbw + 1) / 2;
    const int nS = P * hbw * 36;
    if (idx < nS) {
        const int bb = idx / 36, e = idx - 36 * bb, p = bb / hbw, q = p + (bb - p * hbw);
        if (q >= P) return;
        // the contributing groups are one contiguous run (groups are sorted by f); sixteen loads in flight (their groups' f first),
        // summed in order
        double sum = 0.0;
        const int g1 = d.fgrp[p + 1];
        for (int g0 = d.fgrp[max(0, q - d.whb)]; g0 < g1; g0 += 16) {
            int fz[16]; double v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) fz[u] = d.grp[min(g0 + u, g1 - 1)].z;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int gi = min(g0 + u, g1 - 1);
                const int f = fz[u] & 0xffff, a = p - f, b = q - f, w = a * hbw - a * (a - 1) / 2 + (b - a);
                v[u] = d.wpart[(size_t)gi * d.wstride + w * 36 + e];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) sum += g0 + u < g1 ? v[u] : 0.0;
        }
        const int r = e / 6, c = e - 6 * r;
        d.S[(size_t)(6 * p + r) + (size_t)(6 * q + c) * n] = sum;
        if (p != q) d.S[(size_t)(6 * q + c) + (size_t)(6 * p + r) * n] = sum;
        return;
    }
    const int v = idx - nS;
    if (v >= P * 12) return;
    const int p = v / 12, r = v - 12 * p;
    double sum = 0.0;
    const int g1 = d.fgrp[p + 1];
    for (int g0 = d.fgrp[max(0, p - d.whb)]; g0 < g1; g0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int gi = min(g0 + u, g1 - 1);
            v[u] = d.wpart[(size_t)gi * d.wstride + nwin * 36 + (p - (d.grp[gi].z & 0xffff)) * 12 + r];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) sum += g0 + u < g1 ? v[u] : 0.0;
    }
    if (r < 6) d.g[6 * p + r] = sum; else d.udiag[6 * p + r - 6] = sum;
}
__global__ __launch_bounds__(256) void k_schur_reduce(BADev d, int use_state) { schur_reduce_body(d, use_state); }

// ---- damped solve of the reduced camera system ----------------------------------
// Tiled right-looking Cholesky over 32x32 tiles, one launch per tile column, every
// tile of the trailing matrix on its own workgroup.  The right-hand side rides
// along as row n of the (n+1) x n working matrix, so the forward substitution
// L y = g falls out of the factorisation (y' = last row of L); a single blocked
// back-substitution kernel finishes L' dp = y.
#define CT 32
struct CholArgs { double *A; double *Lf; int n, ld; int *fail; };   // A: working matrix (updated in place); Lf: finished factor tiles

// copy S -> work (lower triangle + rhs row), add the LM damping to the diagonal
__global__ __launch_bounds__(256) void k_chol_prepare(BADev d, const double *Sin, const double *gin, const double *udin,
                                                      double inv_delta_host, int use_state)
{
    if (use_state && d.st->converged) return;
    const int n = d.n, ld = n + 1;
    const double inv_delta = use_state ? 1.0 / d.st->delta : inv_delta_host;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)ld * n) return;
    const int i = (int)(idx % ld), j = (int)(idx / ld);
    double v;
    if (i == n) v = gin[j];
    else {
        v = Sin[(size_t)i + (size_t)j * n];
        if (i == j) v += fmin(fmax(udin[j], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    }
    d.Swork[idx] = v;
}

// Factor a diagonal tile and invert its triangle, by ONE wave, rows in registers.
// t (LDS, 32x33): in = tile (lower part, h rows x w valid columns, rows >= w are
// panel rows riding along), out = L.  inv (LDS): out = L^-1 (w x w lower).
// Lane i owns row i of the tile, which stays in LDS: at step j every lane forms its element of column j
// left-looking, l_ij = (a_ij - sum_{m<j} l_im l_jm) / l_jj, with its own l_im in registers and the l_jm
// (and a_jj) fetched as LDS broadcast reads, which issue back to back -- the v_readlane form of the same
// algorithm paid the scalar-register hazard on every one of its ~1000 broadcasts and was 3x slower.  The
// dot products are formed as four interleaved partial sums (dependent chain j / 4 instead of j).
// Every lane recomputes the pivot l_jj from row j (no communication).  Fully unrolled; tile extents
// (h rows, w columns, wave-uniform) are predicates.
// Called by ALL threads of the workgroup (it contains a barrier); the first two waves work: wave 0 factors, wave 1 inverts one step behind it
// (row i of L and 1 / l_ii are final after factor step i; published through LDS with a step counter), so the
// forward substitution hides behind the factorisation instead of following it.
__device__ __forceinline__ void tile_potrf_inv(double (*t)[CT + 1], double (*inv)[CT + 1], int h, int w, int *fail)
{
    __shared__ double s_rdiag[CT];
    __shared__ int s_prog;                                 // (relaxed workgroup-scope atomics = ds_read / ds_write_b32; a volatile LDS int is a FLAT access
                                                           //  with sc0 sc1 and a vmcnt(0) + lgkmcnt(0) drain in front of every poll)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane < CT ? lane : CT - 1;
    if (threadIdx.x == 0) s_prog = 0;
    __syncthreads();
    if (wv == 0) {
        double lrow[CT];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < CT; j++) {
            const bool active = j < w;
            double acc = t[li][j];                                   // a_ij
            double dj = t[j][j];                                     // a_jj (broadcast)
            {   // four interleaved partial sums each: the dependent chain is j / 4 adds instead of j
                double pa[4] = {0.0, 0.0, 0.0, 0.0}, pd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < j; m++) {
                    const double ljm = t[j][m];                      // broadcast, final since step m
                    pa[m & 3] += lrow[m] * ljm;
                    pd[m & 3] += ljm * ljm;
                }
                acc -= (pa[0] + pa[1]) + (pa[2] + pa[3]);
                dj -= (pd[0] + pd[1]) + (pd[2] + pd[3]);
            }
            bad = bad || (active && !(dj > 0));
            dj = (active && dj > 0) ? dj : 1.0;
            const double rd = rsqrt(dj);
            const double l = (lane == j) ? dj * rd : acc * rd;
            lrow[j] = l;
            if (active && lane >= j && lane < CT) t[lane][j] = l;
            if (lane == 0) s_rdiag[j] = rd;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_store(&s_prog, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (bad && lane == 0) *fail = 1;
    } else if (wv == 1) {
        // inverse: lane c solves L x = e_c by forward substitution, x in registers, row i of L as broadcast reads
        double x[CT];
#pragma unroll
        for (int i = 0; i < CT; i++) {
            while (__hip_atomic_load(&s_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= i) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            double ps[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < i; m++) ps[m & 3] += t[i][m] * x[m];
            const double sacc = ((i == lane) ? 1.0 : 0.0) - ((ps[0] + ps[1]) + (ps[2] + ps[3]));
            x[i] = (i < w && lane <= i) ? sacc * s_rdiag[i] : 0.0;
        }
        if (lane < CT) {
#pragma unroll
            for (int i = 0; i < CT; i++) inv[i][lane] = (lane < w) ? x[i] : 0.0;
        }
    }
}
// zero everything outside the factor's lower-triangular extent (after both waves are done with the tile)
__device__ __forceinline__ void tile_mask_lower(double (*t)[CT + 1], int h, int w)
{
    for (int e = threadIdx.x; e < CT * CT; e += blockDim.x) {
        const int i = e % CT, m = e / CT;
        if (!(m <= i && m < w && i < h)) t[i][m] = 0.0;
    }
}

__global__ __launch_bounds__(256) void k_chol_step(BADev d, CholArgs C, double *Linv, int k, int nbr, int use_state)
{
    if (use_state && d.st->converged) return;
    __shared__ double Li[CT][CT + 1], Lr[CT][CT + 1], Lc[CT][CT + 1], Arc[CT][CT + 1], Tmp[CT][CT + 1];
    const int n = C.n, ld = C.ld, tid = threadIdx.x;
    int r, c;
    {
        int b = blockIdx.x, cc = k;
        const int nbc = (n + CT - 1) / CT;
        while (true) { const int cnt = nbr - cc; if (b < cnt || cc == nbc - 1) { r = cc + b; c = cc; break; } b -= cnt; cc++; }
    }
    if (r == k && c == k) return;                                   // the panel's diagonal tile is already factored
    const int wk = min(CT, n - CT * k);
    const int hr = min(CT, n + 1 - CT * r), hc = min(CT, n + 1 - CT * c);
    const double *Lik = Linv + (size_t)k * CT * CT;
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        const int gj = CT * k + j;
        Li[i][j] = Lik[i + CT * j];
        const int ri = CT * r + i;
        Lr[i][j] = (i < hr && j < wk) ? C.A[(size_t)ri + (size_t)gj * ld] : 0.0;
        if (c > k) {
            const int ci = CT * c + i;
            Lc[i][j] = (c != r && i < hc && j < wk) ? C.A[(size_t)ci + (size_t)gj * ld] : 0.0;
            const int aj = CT * c + j;
            Arc[i][j] = (i < hr && aj < n && ri >= aj) ? C.A[(size_t)ri + (size_t)aj * ld] : 0.0;
        }
    }
    __syncthreads();
    // panel solves as small GEMMs: X[i][j] = sum_{m<=j} B[i][m] * Linv[j][m]
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        double s0 = 0.0;
        for (int m = 0; m <= j; m++) s0 += Lr[i][m] * Li[j][m];
        Tmp[i][j] = s0;
    }
    __syncthreads();
    for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; Lr[i][j] = Tmp[i][j]; }
    if (c > k && c != r) {
        __syncthreads();
        for (int e = tid; e < CT * CT; e += 256) {
            const int i = e % CT, j = e / CT;
            double s0 = 0.0;
            for (int m = 0; m <= j; m++) s0 += Lc[i][m] * Li[j][m];
            Tmp[i][j] = s0;
        }
        __syncthreads();
        for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; Lc[i][j] = Tmp[i][j]; }
    }
    __syncthreads();
    if (c == k) {                                                   // panel tile: store L_rk
        for (int e = tid; e < CT * CT; e += 256) {
            const int i = e % CT, j = e / CT;
            // NOT in place: other workgroups of this launch still read the un-solved panel from A
            if (i < hr && j < wk) C.Lf[(size_t)(CT * r + i) + (size_t)(CT * k + j) * ld] = Lr[i][j];
        }
        return;
    }
    const double (*Lcc)[CT + 1] = (c == r) ? Lr : Lc;
    const int wc = min(CT, n - CT * c);
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        if (i < hr && j < wc && CT * r + i >= CT * c + j) {
            double s0 = Arc[i][j];
            for (int m = 0; m < wk; m++) s0 -= Lr[i][m] * Lcc[j][m];
            Arc[i][j] = s0;
        }
    }
    __syncthreads();
    if (r == k + 1 && c == k + 1) {                                 // next panel's diagonal tile is final now
        tile_potrf_inv(Arc, Tmp, hr, wc, C.fail);
        __syncthreads();
        tile_mask_lower(Arc, hr, wc);
        __syncthreads();
        double *Lo = Linv + (size_t)(k + 1) * CT * CT;
        for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; Lo[i + CT * j] = Tmp[i][j]; }
    }
    double *dstm = (r == k + 1 && c == k + 1) ? C.Lf : C.A;        // a factored diagonal tile is final
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        if (i < hr && j < wc && CT * r + i >= CT * c + j) dstm[(size_t)(CT * r + i) + (size_t)(CT * c + j) * ld] = Arc[i][j];
    }
}

__global__ __launch_bounds__(256) void k_chol_first(BADev d, CholArgs C, double *Linv, int use_state)
{
    if (use_state && d.st->converged) return;
    __shared__ double t[CT][CT + 1], inv[CT][CT + 1];
    const int n = C.n, ld = C.ld, tid = threadIdx.x;
    const int h = min(CT, n + 1), w = min(CT, n);
    for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; t[i][j] = (i < h && j < w && i >= j) ? C.A[(size_t)i + (size_t)j * ld] : 0.0; }
    if (tid == 0) *C.fail = 0;
    __syncthreads();
    tile_potrf_inv(t, inv, h, w, C.fail);
    __syncthreads();
    tile_mask_lower(t, h, w);
    __syncthreads();
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        if (i < h && j < w && i >= j) C.Lf[(size_t)i + (size_t)j * ld] = t[i][j];
        Linv[i + CT * j] = inv[i][j];
    }
}

// L' dp = y (y = row n of the factor), blocked from the last tile column upwards;
// the diagonal solves are mat-vecs with the stored tile inverses.
__global__ __launch_bounds__(256) void k_chol_backsolve(BADev d, CholArgs C, const double *Linv, int use_state)
{
    if (use_state && d.st->converged) return;
    __shared__ double x[SOLVE_MAX_N];
    __shared__ double xb[CT];
    __shared__ double Lis[CT][CT + 1];
    const int n = C.n, ld = C.ld, tid = threadIdx.x;
    for (int a = tid; a < n; a += 256) x[a] = C.Lf[(size_t)n + (size_t)a * ld];
    const int nbc = (n + CT - 1) / CT;
    // The 10 block steps are a dependent chain through x, but what they read from HBM (the tile inverse and the block
    // row of L) does not depend on x: the data of step kb-1 is requested before step kb is computed, so the chain
    // only pays LDS latency and barriers instead of two global round trips per step.
    constexpr int RPT = 2;                                       // prefetched rows per thread (a < 512); larger systems read the rest directly
    double lf[RPT][CT], lfn[RPT][CT], li[4], lin[4];
    auto fetch = [&](int kb, double (&rf)[RPT][CT], double (&ri)[4]) {
        const int j0 = CT * kb, w = min(CT, n - j0);
        const double *Li = Linv + (size_t)kb * CT * CT;
#pragma unroll
        for (int q = 0; q < 4; q++) ri[q] = Li[tid + 256 * q];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int a = tid + 256 * r;
#pragma unroll
            for (int j = 0; j < CT; j++) rf[r][j] = (a < j0 && j < w) ? C.Lf[(size_t)(j0 + j) + (size_t)a * ld] : 0.0;
        }
    };
    fetch(nbc - 1, lf, li);
    __syncthreads();
    for (int kb = nbc - 1; kb >= 0; kb--) {
        const int j0 = CT * kb, w = min(CT, n - j0);
#pragma unroll
        for (int q = 0; q < 4; q++) { const int e = tid + 256 * q; Lis[e % CT][e / CT] = li[q]; }      // Linv tile, [i][j] = Li[i + CT j]
        if (kb > 0) fetch(kb - 1, lfn, lin);
        __syncthreads();
        if (tid < w) {                                              // x_k = Linv' y_k : x[a] = sum_{i>=a} Linv[i][a] y[i]
            double s0 = 0.0;
            for (int i = tid; i < w; i++) s0 += Lis[i][tid] * x[j0 + i];
            xb[tid] = s0;
        }
        __syncthreads();
        if (tid < w) x[j0 + tid] = xb[tid];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; r++) {                             // y_a -= sum_j L[j0+j][a] x[j0+j]
            const int a = tid + 256 * r;
            if (a < j0) {
                double s0 = 0.0;
#pragma unroll
                for (int j = 0; j < CT; j++) s0 += lf[r][j] * x[j0 + j < n ? j0 + j : n - 1];
                x[a] -= s0;
            }
        }
        for (int a = tid + 256 * RPT; a < j0; a += 256) {            // rows beyond the prefetched ones (n > 512)
            double s0 = 0.0;
            for (int j = 0; j < w; j++) s0 += C.Lf[(size_t)(j0 + j) + (size_t)a * ld] * x[j0 + j];
            x[a] -= s0;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; r++)
#pragma unroll
            for (int j = 0; j < CT; j++) lf[r][j] = lfn[r][j];
#pragma unroll
        for (int q = 0; q < 4; q++) li[q] = lin[q];
    }
    for (int a = tid; a < n; a += 256) d.dp[a] = x[a];
    if (tid == 0 && *C.fail) d.st->chol_fail = 1;
}

// ---- banded solve of the reduced camera system in ONE launch --------------------------------------------------------
// A windowed problem couples pose p only with poses p - hb .. p + hb (a map point is seen by a run of consecutive
// key-frames: hb = 9 for the 10-observer scenes): S is block-banded, and the factorisation of block column k only touches
// the (hb+1) x (hb+1) window of 6x6 blocks below / right of it.  One 256-thread workgroup keeps that window in LDS as a
// ring (block (i, j) in slot [i mod (hb+1)][j mod (hb+1)]), walks the block columns left to right and replaces the
// launch chain k_chol_prepare / k_chol_first / k_chol_step x nbc / k_chol_backsolve:
//   P1  every thread factors the 6x6 diagonal block D_k = L L' and inverts L in registers (redundantly: no hand-off),
//       then thread (i, r) forms row r of L_ik = A_ik L^-T for the <= hb blocks below it; the right-hand side rides along
//       as one more row (forward substitution for free); the panel goes to LDS and to the global factor store;
//   P2  trailing update A_ij -= L_ik L_jk' of the window (<= hb (hb+1) / 2 block pairs), the block row k + hb + 1
//       (requested from S one step earlier, damping added on the way) enters the slot row k just vacated.
// Back-substitution L' dp = y then walks the block columns right to left with the stored L_ik and L_kk^-1.
// Systems whose half-bandwidth exceeds BS_MAXHB blocks (dense windows of > 21 poses) keep the tiled path.
#define BS_MAXHB 20
struct BandArgs { const double *S, *g, *ud; double *Lg; int nb, hb; double inv_delta_host; int *fail; long long *trace; int lds_bytes; double *xchg; int epoch; int shift; int p0; };
#define BS_PF 6      // prefetch registers per prefetch thread: ceil(((BS_MAXHB + 1) * 36 + 6) / BS_PT)
#define BS_WS 38     // doubles per 6 x 6 block in the window ring and the panel: 36 + 2, so that the blocks the lanes of a wave read at the
                     // same time start 12 banks apart (a stride of 36 doubles = 8 banks puts every fourth block on the same ones)

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the outstanding global loads / stores (the
// prefetch of the next block row, the factor store), which put a full memory round trip into every step
__device__ __forceinline__ void bs_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#define BS_T 512      /* eight waves, two per SIMD (256 registers each, no spills); roles in the column loop: see there */
#define BS_PT 128     /* threads of the prefetch waves (3 and 7) */
#define BS_UT 128     /* threads of the update waves (1-2) */
__device__ __forceinline__ void band_solve_body(const BADev &d, const BandArgs &B, int use_state)
{
    if (use_state && d.st->converged) return;
    extern __shared__ __attribute__((aligned(16))) double bs_sm[];
    __shared__ int s_bad, s_step;
    // Twisted factorisation (two workgroups at work): side 0 eliminates the poses 0 .. own - 1 top-down, side 1 the poses P - 1 ..
    // P - own' bottom-up (the same algorithm on the block-reversed matrix: pose pi(i) = P - 1 - i) -- at the same time, on two CUs.
    // The hb poses in the middle receive the Schur updates of both: side 1 hands its trailing window over through global memory
    // (B.xchg, flag = launch epoch), side 0 adds it to its own (M = ringA + ringB - S), factors the middle and back-substitutes it,
    // publishes dp of the middle poses, and both sides run their back-substitution outwards.  Sequential block columns: P / 2 + hb / 2
    // instead of P, both ways.  A single workgroup (grid 1) runs the plain factorisation.
    const int hb = B.hb, hb1 = hb + 1, nbT = B.nb, n = d.n, tid = threadIdx.x;
    // The twisted launch has NINE workgroups: the two sides are workgroups 0 and 8 -- workgroups go to the eight XCDs round-robin, so
    // these two share an L2 and their two hand-overs (the trailing window, the middle dp) are L2 round trips instead of trips through
    // the fabric; workgroups 1-7 leave at once.
    // (a launch of TWO workgroups -- SLAMHIP_TWIST_SPREAD=1, a test knob -- puts the sides on neighbouring XCDs: the hand-overs then take
    //  the agent-scope path below)
    const bool tw = gridDim.x > 1;
    if (gridDim.x == 9 && (blockIdx.x & 7) != 0) { if (B.trace && threadIdx.x == 0) B.trace[80 + blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20); return; }
    const int side = gridDim.x == 9 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    // side 0 also eliminates the middle, after side 1's window has arrived: side 1 gets fewer columns so that it is there in time
    int ownA = (nbT - hb) / 2 + B.shift; if (ownA > nbT - hb - 4) ownA = nbT - hb - 4;
    const int ownB = nbT - hb - ownA;
    const int own = tw ? (side ? ownB : ownA) : nbT;      // block columns this side eliminates
    const int nb = tw ? own + hb : nbT;                   // its local system: own columns, then the middle
    auto gi = [&](int i) { return side ? nbT - 1 - i : i; };          // local block index -> pose
    double *x = bs_sm;                                   // [n]: y, then dp
    double *damp = x + n;                                // [n]: LM damping of the diagonal (k_chol_prepare)
    double *chat = damp + n;                             // [n]: L_kk^-T y_k (narrow bands)
    double *LiAll = chat + n;                            // [nb][36]: L_kk of every block column (the factor wave's; 1 / L_jj in the upper triangle), turned into L_kk^-1 for the back-substitution
    double *Wn = LiAll + (size_t)nb * 36;                // [hb1][hb1][BS_WS] window ring, blocks row-major 6x6
    double *rhs = Wn + (size_t)hb1 * hb1 * BS_WS;        // [hb1][6]
    double *Lp = rhs + hb1 * 6;                          // [hb1][BS_WS]: Lp[di] = L_{k+di,k}
    double *yk = Lp + hb1 * BS_WS;                       // [8]
    double *part = yk + 8;                               // [hb1][6] partial sums of the back-substitution
    double *Dn = part + hb1 * 6;                         // [36]: the next diagonal block, updated
    double *Gs = Wn;                                     // narrow bands, after the factorisation: the staged G blocks (see the back-substitution)
    const bool narrow = hb * 6 <= 58;                    // hb <= 9: the back-substitution is a one-wave recurrence
    unsigned char *ptab = (unsigned char *)(Dn + 36);    // [hb (hb+1) / 2][2] pair table (di, dj), dj <= di, ordered by di
    const double inv_delta = use_state ? 1.0 / d.st->delta : B.inv_delta_host;
    const size_t lgs = (size_t)hb1 * 36 + 8;             // doubles per block column in the global factor store
    double *const Lg = B.Lg + (size_t)side * nbT * lgs;
    double *const dpo = d.dp + 6 * B.p0;                   // dp of the solve's first pose (the span of the free poses)
    // The two sides of a twisted solve hand data to each other through global memory.  On the same XCD (the normal case, see above) the
    // L2 is common: the producer's stores only have to have arrived there (s_waitcnt vmcnt(0)) and the consumer reads with sc1 loads, past
    // its own L1 -- no agent-scope fence, whose L2 write-back / invalidate costs microseconds.  Each side publishes its XCC_ID (tagged
    // with the launch epoch) at the start and compares the other's with its own at its hand-over; a side that does not see a matching
    // id takes the agent-scope fence.
    const long long tr_in = B.trace ? clock64() : 0;
    const int myxcc = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15);
    const int xcc_tag = (int)(((unsigned)B.epoch << 5) + 16u);       // (unsigned: the epoch counts launches for the life of the solver object)
    if (tw && tid == 0) __hip_atomic_store(B.fail + 3 + side, xcc_tag + myxcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto same_xcd = [&]() { return __hip_atomic_load(B.fail + 3 + (1 - side), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc_tag + myxcc; };
    if (tid == 0) { s_bad = 0; s_step = 0; }
    if (tid < hb * hb1 / 2) {                                // pair tid = (di, dj), 1 <= dj <= di <= hb, ordered by di
        int di = 1; while (di * (di + 1) / 2 <= tid) di++;
        ptab[2 * tid] = (unsigned char)di; ptab[2 * tid + 1] = (unsigned char)(tid - di * (di - 1) / 2 + 1);
    }
    if (B.trace && side == 0 && tid == 0) B.trace[103] = clock64() - tr_in;
    // ---- set-up: damping and the first window (block rows 0 .. hb).  Every global load of the set-up is requested before any of them
    //      is consumed -- S was written by other kernels from all eight XCDs: first touches are HBM round trips, and row after row
    //      (load, wait, store) the set-up took 34 k cycles, a sixth of the kernel.  A thread owns one slot (r, jb, c) -- or a right-hand-side
    //      entry -- of the generic band row [hb + 1 blocks | 6] and loads it for all hb + 1 rows (block column i - hb + jb: the rows of the
    //      first window lack their leading blocks).  The values are consumed further down, after the other requests of the set-up (the next
    //      row, the L2 warm-up) have gone out too ----
        const double udv = tid < 6 * nb ? B.ud[6 * gi(tid / 6) + tid % 6] : 0.0;          // (6 nb <= BS_T up to nb = 85; the entries beyond follow below)
        constexpr int NS = ((BS_MAXHB + 1) * 36 + 6 + BS_T - 1) / BS_T;                    // slots per thread (2)
        double wv[NS][BS_MAXHB + 1];
        // per slot: global index of row 0 and its stride per row (linear in the row number on either side), first valid row, LDS word
        // of row 0 (the ring advances by one row and one column per row: (hb + 2) blocks), the diagonal's damping entry or -1
        int s_g[NS], s_st[NS], s_i0[NS], s_l[NS], s_ls[NS], s_dm[NS], s_kind[NS];           // s_kind: 0 none, 1 block entry, 2 right-hand side
        const int nrow0 = nb - 1 < hb ? nb : hb1;                                            // rows of the first window
#pragma unroll
        for (int q = 0; q < NS; q++) {
            const int e = tid + q * BS_T;
            s_g[q] = 0; s_st[q] = 0; s_i0[q] = 1 << 20; s_l[q] = 0; s_ls[q] = 0; s_dm[q] = -1; s_kind[q] = 0;
            if (e < hb1 * 36) {
                int r = 0; while (e >= (r + 1) * 6 * hb1) r++;
                const int t = e - r * 6 * hb1, jb = t / 6, c = t - 6 * jb;
                auto idx = [&](int i) { return (6 * gi(i - hb + jb) + c) + (6 * gi(i) + r) * n; };
                s_kind[q] = 1; s_i0[q] = hb - jb; s_g[q] = idx(hb) - hb * (idx(hb + 1) - idx(hb)); s_st[q] = idx(hb + 1) - idx(hb);
                s_l[q] = (int)(Wn - bs_sm) + (jb - hb) * BS_WS + r * 6 + c; s_ls[q] = (hb1 + 1) * BS_WS;
                if (jb == hb && r == c) s_dm[q] = r;
            } else if (e < hb1 * 36 + 6) {
                const int c = e - hb1 * 36;
                s_kind[q] = 2; s_i0[q] = 0; s_g[q] = 6 * gi(0) + c; s_st[q] = 6 * (gi(1) - gi(0));
                s_l[q] = (int)(rhs - bs_sm) + c; s_ls[q] = 6;
            }
            auto request = [&](auto R) {                          // rows 0 .. R - 1 (straight-line: every load goes out before anything waits)
#pragma unroll
                for (int i = 0; i < decltype(R)::value; i++) {
                    wv[q][i] = 0.0;
                    if (i >= s_i0[q] && i < nrow0) wv[q][i] = (s_kind[q] == 1 ? B.S : B.g)[s_g[q] + i * s_st[q]];
                }
            };
            if (hb1 <= 10) request(std::integral_constant<int, 10>()); else request(std::integral_constant<int, BS_MAXHB + 1>());
        }
    if (B.trace && side == 0 && tid == 0) B.trace[104] = clock64() - tr_in;
    // Rows i > hb all have hb + 1 blocks.  The prefetch lanes (waves 3 and 7) own fixed elements of such a row; the global index of an
    // element is linear in the row number on either side (pose = i or P - 1 - i), so a lane keeps its elements' indices for row hb + 1
    // and per row only adds a stride; the ring slot moves with the row.
    double pf[BS_PF];
    int2 *etab = (int2 *)(Dn + 36 + 56);                      // [BS_PF][BS_PT], behind the pair table's <= 420 bytes (in LDS: registers are what the update waves are short of)
    // .x: jb * 64 + r * 6 + c (+ 4096 on the diagonal block's diagonal), or -(1 + c) for a right-hand-side entry, -1000: none;  .y: index in S / g for row hb + 1
    int strideS = 0, strideG = 0;
    const bool pl = (tid >> 6) == 3 || (tid >> 6) == 7;
    const int pidx = ((tid >> 6) == 7 ? 64 : 0) + (tid & 63);
    {
        const int cnt = hb1 * 36 + 6, i0 = hb + 1;
        auto idxS = [&](int i, int jb, int r, int c) { return (6 * gi(i - hb + jb) + c) + (6 * gi(i) + r) * n; };
        strideS = idxS(i0 + 1, 0, 0, 0) - idxS(i0, 0, 0, 0); strideG = 6 * (gi(i0 + 1) - gi(i0));
        if (pl)
            for (int q = 0; q < BS_PF; q++) {
                const int e = pidx + BS_PT * q;
                int2 v = make_int2(-1000, 0);
                if (e < hb1 * 36) {
                    int r = 0; while (e >= (r + 1) * 6 * hb1) r++;
                    const int t = e - r * 6 * hb1, jb = t / 6, c = t - 6 * jb;
                    v = make_int2(jb * 64 + r * 6 + c + ((jb == hb && r == c) ? 4096 : 0), idxS(i0, jb, r, c));
                } else if (e < cnt) v = make_int2(-(1 + (e - hb1 * 36)), 6 * gi(i0) + (e - hb1 * 36));
                etab[q * BS_PT + pidx] = v;
            }
    }
    // (the table is read in one go -- BS_PF independent LDS reads, one wait -- and the entries' work is branch-free where it can be: a
    //  read, a wait and a branch per entry made this the longest wave of a step)
    auto load_tab = [&](int2 (&e)[BS_PF]) {
#pragma unroll
        for (int q = 0; q < BS_PF; q++) e[q] = etab[q * BS_PT + pidx];
    };
    auto fetch_row = [&](int i, const int2 (&e)[BS_PF]) {    // i > hb; prefetch lanes only
        const int di = i - (hb + 1);
#pragma unroll
        for (int q = 0; q < BS_PF; q++) {
            const double *src = e[q].x >= 0 ? B.S + (e[q].y + di * strideS) : B.g + (e[q].y + di * strideG);
            pf[q] = 0.0;
            if (e[q].x > -1000) pf[q] = *src;
        }
    };
    auto put_row = [&](int i, int ri, const int2 (&e)[BS_PF], const double (&v)[BS_PF], bool blocks, bool rhs_rows) {      // i > hb, ri = i mod (hb + 1)
#pragma unroll
        for (int q = 0; q < BS_PF; q++) {
            const int ew = e[q].x;
            if (ew >= 0) {
                const int jb = (ew >> 6) & 63, rc = ew & 63;
                int sl = ri + 1 + jb; if (sl >= hb1) sl -= hb1;            // (i - hb + jb) mod (hb + 1)
                if (blocks) Wn[(ri * hb1 + sl) * BS_WS + rc] = ew >= 4096 ? v[q] + damp[6 * i + rc / 7] : v[q];
            } else if (rhs_rows && ew > -1000) rhs[ri * 6 - 1 - ew] = v[q];
        }
    };
    bs_barrier();                                             // (the table: its lanes only read their own entries, but ptab above is everybody's; LDS only: the set-up's loads stay in flight)
    if (B.trace && side == 0 && tid == 0) B.trace[105] = clock64() - tr_in;
    if (pl && hb + 1 < nb) { int2 e[BS_PF]; load_tab(e); fetch_row(hb + 1, e); }
    double warm_acc = 0.0;
    {   // Pull the rest of the band into this XCD's L2 now.  k_blocks wrote S from all eight XCDs, so the first touch of a line
        // is an HBM round trip (~2.5 us, about a factorisation step): with the lines resident, the one-step-ahead request
        // of the prefetch wave is an L2 hit.  A scalar row's band segment is 6 (hb + 1) contiguous doubles: one load per 128-byte line.
        double acc = 0.0;
        const int seg = 6 * hb1, lines = (seg + 15) / 16 + 1, nsr = 6 * (nb - hb - 2);
        if (tid >= 256 && !pl)                                  // waves 4-6: nothing in the column loop makes them wait for memory
        for (int e = tid - 256 - (tid >= 448 ? 64 : 0); e < nsr * lines; e += 192) {
            const int sr = e / lines, l = e - sr * lines, i = hb + 2 + sr / 6, r = sr - 6 * (sr / 6);
            const int c0 = 6 * gi(i - hb) < 6 * gi(i) ? 6 * gi(i - hb) : 6 * gi(i);             // first column of the segment on either side
            int off = 16 * l; if (off > seg - 1) off = seg - 1;
            acc += B.S[(size_t)(c0 + off) + (size_t)(6 * gi(i) + r) * n];
        }
        warm_acc = acc;
    }
    if (B.trace && side == 0 && tid == 0) B.trace[106] = clock64() - tr_in;
    // ---- the first window goes into the ring ----
        if (tid < 6 * nb) damp[tid] = fmin(fmax(udv, LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        for (int a = tid + BS_T; a < 6 * nb; a += BS_T) damp[a] = fmin(fmax(B.ud[6 * gi(a / 6) + a % 6], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
        __syncthreads();
        if (B.trace && side == 0 && tid == 0) B.trace[107] = clock64() - tr_in;
#pragma unroll
        for (int q = 0; q < NS; q++) {
            auto place = [&](auto R) {                             // ring row i, ring column i - hb + jb (no wrap inside the first window)
#pragma unroll
                for (int i = 0; i < decltype(R)::value; i++)
                    if (i >= s_i0[q] && i < nrow0) bs_sm[s_l[q] + i * s_ls[q]] = s_dm[q] >= 0 ? wv[q][i] + damp[6 * i + s_dm[q]] : wv[q][i];
            };
            if (hb1 <= 10) place(std::integral_constant<int, 10>()); else place(std::integral_constant<int, BS_MAXHB + 1>());
        }
    __syncthreads();
    bool bad = false;
    long long tr0 = B.trace ? clock64() : 0, trT = tr0, trA = 0, trB = 0, trC = 0, trD = 0, trW = 0;
#ifdef BS_TRACE_ACC      /* per-phase sums: every clock read costs the wave ~200 cycles, so only on request; the step-10 stamps below are always there */
#define BS_TR(acc) if (B.trace) { const long long t_ = clock64(); acc += t_ - trT; trT = t_; }
#else
#define BS_TR(acc)
#endif
    // Roles in the column loop -- one busy wave per SIMD (the waves w and w + 4 of a workgroup share a SIMD, and the second of two busy
    // waves only gets the issue slots the first one leaves), and as few LDS instructions as possible (the LDS takes a fixed number of
    // cycles per wave instruction however many lanes are active; the trailing update's loads are what a step's other LDS traffic queues behind):
    //   wave 0  the factor wave;   waves 1-2  trailing update;   wave 3 (+ wave 7, a few instructions)  ring prefetch;
    //   wave 4  copies L_kk^-1 to the factor store;   waves 5-6  idle.
    const bool fwave = tid < 64;
    const bool uwave = tid >= 64 && tid < 64 + BS_UT;
    const bool pwave = (tid >> 6) == 3 || (tid >> 6) == 7;
    const int flane = tid;
    // Factor wave.  Every lane holds L_kk (Lr) and 1 / diag(L_kk) (invd) of the block column being eliminated.  Step k: lane t forms row
    // t of the panel L_ik = A_ik L_kk^-T by forward substitution from those registers (the right-hand side rides along as one more row),
    // the panel is published with an LDS flag (the update and prefetch waves poll it: no workgroup barrier); lane (r, c) reads rows r and c
    // of L_{k+1,k} back and forms its entry of D_{k+1} = A_{k+1,k+1} - L L'; D_{k+1} reaches every lane through v_readlane (no LDS
    // round trip behind the update waves' loads) and is factored there.  One barrier per step, at its end: the critical path of a step
    // stays in this wave.
    //
    // What the wave's instructions cost (scripts/ubench/f64_issue.hip, one wave on its SIMD): a dependent v_fma_f64 / v_mul_f64 36 / 32
    // cycles, an independent one 9.5 (a lone wave gets every other f64 issue slot), rsqrt() 88-120.  The textbook Cholesky loop has ~10
    // dependent operations per pivot: ~2 000 cycles for a 6 x 6 block.  Here the elimination runs DIVISION-FREE on scaled entries --
    // m_ik <- m_ik p_j - m_ij m_kj with p_j the scaled pivot; every second pivot the entries are rescaled by the power of two that
    // brings the pivot to [0.5, 1) (exact; it bounds the magnitudes at the 4th power of the block's dynamic range) -- two (three)
    // dependent operations per pivot.  If s_j is the scale the entries carry at step j (s_0 = 1, s_{j+1} = s_j p_j c_j), the true pivot is
    // p_j / s_j and the Cholesky column is L_ij = m_ij rsqrt(p_j s_j): the six rsqrt (v_rsq_f64 + one Newton step: 4e-15 relative) are
    // independent of each other.  L_kk^-1 is not formed here at all: the back-substitution inverts the blocks it needs, all at once.
    double Lr[6][6], invd[6];
    auto factor = [&](const double (&dd)[21], double *LOut) {        // dd: lower triangle, row-major
#pragma clang fp contract(fast)
        double M[21], ps[6], sj[6];
#pragma unroll
        for (int q = 0; q < 21; q++) M[q] = dd[q];
        double sc = 1.0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double pj = M[j * (j + 1) / 2 + j];
            bad = bad || !(pj > 0);
            pj = pj > 0 ? pj : 1.0;
            ps[j] = pj * sc;                                  // p_j s_j
            sj[j] = sc;
            Lr[j][j] = pj;
#pragma unroll
            for (int i = j + 1; i < 6; i++) Lr[i][j] = M[i * (i + 1) / 2 + j];       // (unscaled column: times rsqrt(p_j s_j) below)
            if ((j & 1) == 0) {
                const int e = -__builtin_amdgcn_frexp_exp(pj);
                sc *= __builtin_amdgcn_frexp_mant(pj);
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int k2 = j + 1; k2 <= i; k2++) {
                        const double t = M[i * (i + 1) / 2 + k2] * pj - M[i * (i + 1) / 2 + j] * M[k2 * (k2 + 1) / 2 + j];
                        M[i * (i + 1) / 2 + k2] = __builtin_amdgcn_ldexp(t, e);
                    }
            } else {
                sc *= pj;
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int k2 = j + 1; k2 <= i; k2++)
                        M[i * (i + 1) / 2 + k2] = M[i * (i + 1) / 2 + k2] * pj - M[i * (i + 1) / 2 + j] * M[k2 * (k2 + 1) / 2 + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double y0 = __builtin_amdgcn_rsq(ps[j]);
            const double r0 = fma(-(ps[j] * y0), y0, 1.0), rd = fma(y0 * 0.5, r0, y0);
            invd[j] = sj[j] * rd;                             // 1 / L_jj = s_j rsqrt(p_j s_j)
#pragma unroll
            for (int i = j; i < 6; i++) Lr[i][j] *= rd;
        }
        if (flane == 0) {                                     // L_kk for the back-substitution (lower triangle; zeros above)
            double lo[36];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int c = 0; c < 6; c++) lo[i * 6 + c] = c <= i ? Lr[i][c] : 0.0;
            lo[1] = invd[0]; lo[2] = invd[1]; lo[3] = invd[2]; lo[4] = invd[3]; lo[5] = invd[4]; lo[8] = invd[5];      // 1 / L_jj ride in the upper triangle (bs_invd_slot)
            st_rec<36>(LOut, lo);
        }
    };
    auto factor_lds = [&](const double *D, double *LiOut) {
        double dd[21];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) dd[i * (i + 1) / 2 + j] = D[i * 6 + j];
        factor(dd, LiOut);
    };
    const int u_di = ptab[2 * (tid & 63) < hb * hb1 ? 2 * (tid & 63) : 0], u_dj = ptab[2 * (tid & 63) + 1 < hb * hb1 ? 2 * (tid & 63) + 1 : 0];    // an update lane's pair (first 64 pairs)
    bool has_rhs_el = false;                                     // a prefetch lane that carries right-hand-side entries of the incoming row
    if (pl) for (int q = 0; q < BS_PF; q++) { const int ew = etab[q * BS_PT + pidx].x; has_rhs_el = has_rhs_el || (ew < 0 && ew > -1000); }
    if (fwave) factor_lds(Wn, LiAll);                            // D_0 = block (0, 0), ring slot [0][0]
    bs_barrier();
    if (B.trace && side == 0 && tid == 0) { B.trace[16] = clock64() - tr0; B.trace[26] = tr0 - tr_in; }
    if (B.trace && tid == 0) B.trace[20 + side] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    int kbeg = 0, kend = own, kk = 0;                            // kk = k mod (hb + 1), kept by hand (a run-time division costs ~40 scalar instructions)
    for (int phase = 0; ; phase++) {
    for (int k = kbeg; k < kend; k++, kk = kk + 1 == hb1 ? 0 : kk + 1) {
        const int np = nb - 1 - k < hb ? nb - 1 - k : hb;        // blocks below the diagonal in this column
        double *Lgk = Lg + (size_t)k * lgs;
        const bool stamp = B.trace && side == 0 && k == 10 && (tid & 63) == 0;
        if (stamp) B.trace[32 + (tid >> 6)] = clock64();
        if (fwave) {
#pragma clang fp contract(fast)
            // ---- the panel rows L_ik = A_ik L_kk^-T and the right-hand side (lane t: row t) ----
            const int nrow = np * 6 + 1;
            int r1 = kk + 1; if (r1 >= hb1) r1 -= hb1;
            auto fwd = [&](const double (&a)[6], double (&o)[6]) {       // o L' = a: o_q = (a_q - sum_{m < q} o_m L_qm) / L_qq
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    double t = a[q];
#pragma unroll
                    for (int m = 0; m < q; m++) t -= o[m] * Lr[q][m];
                    o[q] = t * invd[q];
                }
            };
            const int fl = flane < 36 ? flane : 0, fr = fl / 6, fc = fl - 6 * fr;
            const double dv = Wn[(r1 * hb1 + r1) * BS_WS + fl];
            for (int t = flane; t < nrow; t += 64) {             // (one trip up to hb = 10)
                const bool rh = t == np * 6;
                const int di = t / 6 + 1, r = t - 6 * (di - 1);
                int ri = kk + di; if (ri >= hb1) ri -= hb1;
                const double *Arow = rh ? rhs + kk * 6 : Wn + (ri * hb1 + kk) * BS_WS + r * 6;
                double a[6], o[6];
                ld_rec<6>(Arow, a);
                if (stamp) B.trace[96] = clock64();
                fwd(a, o);
                if (stamp) B.trace[97] = clock64();
                if (rh) { st_rec<6>(yk, o); st_rec<6>(x + 6 * k, o); }
                else { st_rec<6>(Lp + di * BS_WS + r * 6, o); st_rec<6>(Lgk + di * 36 + r * 6, o); }
            }
            asm volatile("" ::: "memory");                       // the LDS executes a wave's instructions in order: the flag lands after the panel
            if (flane == 0) __hip_atomic_store(&s_step, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            // ---- D_{k+1}: lane (r, c) < 36 reads rows r and c of L_{k+1,k} back (this wave's own stores: in order) ----
            double xr[6], xc[6];
            ld_rec<6>(Lp + BS_WS + fr * 6, xr); ld_rec<6>(Lp + BS_WS + fc * 6, xc);
            xr[0] = dv - ((xr[0] * xc[0] + xr[1] * xc[1] + xr[2] * xc[2]) + (xr[3] * xc[3] + xr[4] * xc[4] + xr[5] * xc[5]));
            if (flane < 36 && k + 1 < nb) Dn[flane] = xr[0];     // (the hand-over to / from the other side reads it after the loop)
            if (stamp) B.trace[98] = clock64();
            BS_TR(trB)
            if (stamp) B.trace[40] = clock64();
            if (k + 1 < nb) {
                double dd[21];
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        const int lo = __builtin_amdgcn_readlane(__double2loint(xr[0]), i * 6 + j), hi = __builtin_amdgcn_readlane(__double2hiint(xr[0]), i * 6 + j);
                        dd[i * (i + 1) / 2 + j] = __hiloint2double(hi, lo);
                    }
                if (stamp) B.trace[99] = clock64();
                factor(dd, LiAll + 36 * (k + 1));
            }
            BS_TR(trA)
        } else if (uwave) {
#pragma clang fp contract(fast)
            // ---- trailing update of the window: lane = block pair (di, dj), dj <= di; wave 1 takes rows 0-2 of the block, wave 2 rows 3-5 ----
            while (__hip_atomic_load(&s_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            BS_TR(trW)
            const int npair = np * (np + 1) / 2, h = (tid >> 6) - 1;
            for (int pr = tid & 63; pr < npair; pr += 64) {
                if (pr == 0) continue;                           // pair (1, 1) = the next diagonal block: the factor wave's
                const int di = pr < 64 ? u_di : ptab[2 * pr], dj = pr < 64 ? u_dj : ptab[2 * pr + 1];
                int ri = kk + di; if (ri >= hb1) ri -= hb1;
                int rj = kk + dj; if (rj >= hb1) rj -= hb1;
                const double *Ai = Lp + di * BS_WS + h * 18, *Lj = Lp + dj * BS_WS;
                double *Wb = Wn + (ri * hb1 + rj) * BS_WS + h * 18;
                double a[18], wb[18];
                ld_rec<18>(Ai, a);                                // 16-byte LDS accesses (ds_read_b128: 256 B per clock; ds_read2_b64: half that)
                ld_rec<18>(Wb, wb);
#pragma unroll
                for (int hc = 0; hc < 2; hc++) {                 // L_dj in two halves (the registers: 256 per lane with two waves per SIMD)
                    double lj[18];
                    ld_rec<18>(Lj + hc * 18, lj);
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            double tt = 0.0;
#pragma unroll
                            for (int m = 0; m < 6; m++) tt += a[r * 6 + m] * lj[c * 6 + m];
                            wb[r * 6 + hc * 3 + c] -= tt;
                        }
                }
                st_rec<18>(Wb, wb);
            }
            BS_TR(trC)
        } else if ((tid >> 6) == 5) {
#pragma clang fp contract(fast)
            // ---- the right-hand side rows of the window: lane = dj ----
            while (__hip_atomic_load(&s_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            for (int dj = (tid & 63) + 1; dj <= np; dj += 64) {
                int rj = kk + dj; if (rj >= hb1) rj -= hb1;
                const double *Lj = Lp + dj * BS_WS;
                double a[6], lj[36], wb[6];
                ld_rec<6>(yk, a);
                ld_rec<36>(Lj, lj);
                ld_rec<6>(rhs + rj * 6, wb);
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double tt = 0.0;
#pragma unroll
                    for (int m = 0; m < 6; m++) tt += a[m] * lj[c * 6 + m];
                    wb[c] -= tt;
                }
                st_rec<6>(rhs + rj * 6, wb);
            }
        } else if (pwave) {
            // ---- the next block row enters the ring row that block row k vacated ((k + 1 + hb) mod (hb + 1) = k mod (hb + 1)): nobody reads
            //      that row's blocks in this step, so they go in right away, while the LDS is idle (the update waves' bursts start at the
            //      flag), and the row after it is requested; the row's right-hand side waits for the flag -- the factor wave reads the old
            //      one for the panel ----
            double pg[BS_PF];
            int2 e[BS_PF];
            load_tab(e);
#pragma unroll
            for (int q = 0; q < BS_PF; q++) pg[q] = pf[q];
            if (stamp && tid == 192) B.trace[72] = clock64();
            if (k + 1 + hb < nb) put_row(k + 1 + hb, kk, e, pg, true, false);
            if (stamp && tid == 192) B.trace[73] = clock64();
            if (k + 2 + hb < nb) fetch_row(k + 2 + hb, e);
            if (stamp && tid == 192) B.trace[74] = clock64();
            if (has_rhs_el) {
                while (__hip_atomic_load(&s_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                if (k + 1 + hb < nb) put_row(k + 1 + hb, kk, e, pg, false, true);
            }
            BS_TR(trC)
        }
        if (stamp) B.trace[48 + (tid >> 6)] = clock64();
        bs_barrier();
        BS_TR(trD)
        if (stamp) { B.trace[56 + (tid >> 6)] = clock64(); B.trace[64 + (tid >> 6)] = __builtin_amdgcn_s_getreg(63492); }
    }
    if (B.trace && side == 0 && tid == 0) B.trace[17 + 2 * phase] = clock64() - tr0;
    if (!tw || phase == 1) break;
    // ---- the middle: rows own .. own + hb - 1 of the ring hold S - (this side's updates); the diagonal block (own, own) is in Dn
    //      (the factor wave keeps the next diagonal block to itself).  Lower blocks (i >= j), row-major hb x hb triangle + rhs.
    const int ntri = hb * (hb + 1) / 2;
    if (side == 1) {
        for (int e = tid; e < ntri * 36 + hb * 6; e += BS_T) {
            double v;
            if (e < ntri * 36) {
                const int bq = e / 36, rc = e - 36 * bq;
                int i = 0, q = bq; while (q > i) { q -= i + 1; i++; }            // bq = i (i + 1) / 2 + j
                const int li = own + i, lj = own + q;
                v = (i == 0) ? Dn[rc] : Wn[((size_t)(li % hb1) * hb1 + (lj % hb1)) * BS_WS + rc];
            } else {
                const int i = (e - ntri * 36) / 6, r = e - ntri * 36 - 6 * i;
                v = rhs[((own + i) % hb1) * 6 + r];
            }
            __hip_atomic_store(B.xchg + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (an agent-scope store: no data race with the other workgroup's atomic loads)
        }
        if (bad) s_bad = 1;
        // hand-over on a common L2, made explicit (it does not lean on how the compiler lowers a workgroup-scope fence): every store
        // of this wave has ARRIVED at the L2 (the L1 is write-through; vmcnt counts stores until they are acknowledged) before the
        // barrier, the flag store follows the barrier, and the consumer reads flag and data with agent-scope (sc1) loads, past its L1
        if (same_xcd()) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(B.fail + 1, B.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
    }
    // M = ringA + ringB' - S for the middle.  What does not depend on the other side is done BEFORE the wait: each thread's (<= 4) entries --
    // where they come from and go to -- and their S / g values (requested now, in flight during the wait); after the flag the other side's
    // values are requested together: one L2 round trip, not one per entry.
    // (a thread keeps one position (r, c) inside the blocks and takes block t / 36 + 14 qq in pass qq; the blocks' (i, q) come from the
    //  pair table; the 6 hb right-hand-side entries ride in the last pass's spare slots: no run-time division, no search)
    constexpr int ME = 4;                                          // 14 blocks per pass: 56 >= 45 (twisted solves have hb <= 9) + the right-hand side
    int m_x[ME], m_w[ME], m_s[ME]; double m_sd[ME];               // index in xchg; LDS word (of bs_sm) written / read; S (+ damping) or g value
    {
        const int rc = tid % 36, r = rc / 6, c = rc - 6 * r, b0 = tid / 36, ob = own % hb1;
#pragma unroll
        for (int qq = 0; qq < ME; qq++) {
            const int bq = b0 + 14 * qq;
            m_x[qq] = -1; m_w[qq] = 0; m_s[qq] = 0; m_sd[qq] = 0.0;
            if (tid >= 504) continue;
            if (bq < ntri) {
                const int i = ptab[2 * bq] - 1, q = ptab[2 * bq + 1] - 1;          // middle block (own + i, own + q), i >= q
                // the other side numbers the middle backwards and holds the transposed block: its (hb - 1 - q, hb - 1 - i), entry (c, r)
                const int oi = hb - 1 - q, oj = hb - 1 - i;
                m_x[qq] = (oi * (oi + 1) / 2 + oj) * 36 + c * 6 + r;
                const int I = own + i, J = own + q;
                double sd = B.S[(6 * J + c) + (6 * I + r) * n];                     // (S is symmetric, both halves written: this is the half the start of the kernel pulled into the L2)
                if (i == q && r == c) sd += damp[6 * I + r];
                m_sd[qq] = sd;
                int si = ob + i; if (si >= hb1) si -= hb1;
                int sj = ob + q; if (sj >= hb1) sj -= hb1;
                m_w[qq] = (int)(Wn - bs_sm) + (si * hb1 + sj) * BS_WS + rc;
                m_s[qq] = bq == 0 ? (int)(Dn - bs_sm) + rc : m_w[qq];                // (the factor wave keeps the next diagonal block in Dn)
            } else {
                const int er = (bq - ntri) * 36 + rc;
                if (er < hb * 6) {
                    const int i = er / 6, rr = er - 6 * i;
                    m_x[qq] = ntri * 36 + (hb - 1 - i) * 6 + rr;
                    int si = ob + i; if (si >= hb1) si -= hb1;
                    m_w[qq] = m_s[qq] = (int)(rhs - bs_sm) + si * 6 + rr;
                    m_sd[qq] = B.g[6 * (own + i) + rr];
                }
            }
        }
    }
    if (B.trace && side == 0 && tid == 0) B.trace[22] = clock64() - tr0;
    if (tid == 0) while (__hip_atomic_load(B.fail + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != B.epoch) __builtin_amdgcn_s_sleep(2);
    if (B.trace && side == 0 && tid == 0) B.trace[23] = clock64() - tr0;
    __syncthreads();
    if (same_xcd()) asm volatile("" ::: "memory"); else __threadfence();           // (same L2: the loads below are agent-scope atomics, served by the L2)
    if (B.trace && side == 0 && tid == 0) B.trace[24] = clock64() - tr0;
    {
        double vb[ME];
#pragma unroll
        for (int qq = 0; qq < ME; qq++) vb[qq] = __hip_atomic_load(B.xchg + (m_x[qq] >= 0 ? m_x[qq] : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (sc1: from the L2)
#pragma unroll
        for (int qq = 0; qq < ME; qq++) if (m_x[qq] >= 0) bs_sm[m_w[qq]] = (bs_sm[m_s[qq]] + vb[qq]) - m_sd[qq];
    }
    __syncthreads();
    if (B.trace && side == 0 && tid == 0) B.trace[25] = clock64() - tr0;
    if (fwave) factor_lds(Wn + ((size_t)(own % hb1) * hb1 + (own % hb1)) * BS_WS, LiAll + 36 * own);
    bs_barrier();
    if (B.trace && side == 0 && tid == 0) B.trace[18] = clock64() - tr0;
    kbeg = own; kend = nb;
    }
    if (bad) s_bad = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the factor store is re-read below by other threads (of this workgroup)
    __syncthreads();
    const int kfac2 = (tw && side) ? own : nb;               // columns this side has factored (side 1: not the middle)
    // ---- L_kk -> L_kk^-1, all columns at once: thread = (block column k, column c of the inverse), forward substitution with the
    //      1 / L_jj the factor wave left in the upper triangle; results are written after a barrier (column c overwrites what the
    //      threads of the columns before it read), the upper triangle is cleared ----
    for (int base = 0; base < 6 * kfac2; base += 510) {
#pragma clang fp contract(fast)
        const int e = base + tid, k = e / 6, c = e - 6 * k;
        const bool act = tid < 510 && e < 6 * kfac2;
        double xi[6];
        if (act) {
            const double *Lk = LiAll + 36 * k;
            const int slot[6] = {1, 2, 3, 4, 5, 8};
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double a = i == c ? -1.0 : 0.0;
#pragma unroll
                for (int m = 0; m < i; m++) a += (m >= c ? Lk[i * 6 + m] * xi[m] : 0.0);
                xi[i] = i >= c ? -a * Lk[slot[i]] : 0.0;
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int i = 0; i < 6; i++) LiAll[36 * k + i * 6 + c] = xi[i];       // (zero above the diagonal: i < c)
        }
        __syncthreads();
    }
    if (!narrow) {                                           // the wide-band back-substitution reads L_kk^-1 from the factor store
        for (int e = tid; e < 36 * kfac2; e += BS_T) Lg[(size_t)(e / 36) * lgs + e % 36] = LiAll[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
    }
    // ---- back-substitution L' dp = y, block columns right to left ----
    // step k: (A) thread (di, c): sum_r L_{k+di,k}[r][c] x_{k+di}[r]; (B) thread c: t_c = y_k[c] - the partial sums, in fixed order;
    // (C) thread c: x_k[c] = sum_{m >= c} L_kk^-1[m][c] t_m.  The factor blocks of step k - 1 are requested while step k computes.
    double lreg[6], lic[6];
    auto fetch_back = [&](int k) {
        const int np = nb - 1 - k < hb ? nb - 1 - k : hb;
        const double *Lgk = B.Lg + (size_t)k * lgs;
        if (tid < np * 6) {
            const int di = tid / 6 + 1, c = tid - 6 * (di - 1);
#pragma unroll
            for (int r = 0; r < 6; r++) lreg[r] = Lgk[di * 36 + r * 6 + c];
        }
        if (tid >= BS_T - 6) {
            const int c = tid - (BS_T - 6);
#pragma unroll
            for (int m = 0; m < 6; m++) lic[m] = Lgk[m * 6 + c];          // column c of L_kk^-1 (zero above the diagonal)
        }
    };
    double *tv = yk;                                         // [6] t of the current step
    if (narrow) {
        // narrow bands (hb <= 9): dp_k = chat_k - sum_j G_{k,j} dp_{k+j} with G_{k,j} = L_kk^-T L_{k+j,k}^T, chat_k = L_kk^-T y_k
        // -- nothing in the recurrence but the products with the newest dp.  One wave, lane = (slot s = k' mod (hb+1),
        // row c), holds the running sum of the hb + 1 columns in flight.  Step k: the six lanes of slot k mod (hb+1) add chat_k and
        // hold dp_k; it is broadcast with v_readlane, every other slot does base -= G_{k',k-k'}[c][:] . dp_k (six multiply-adds),
        // the slot of k restarts at zero for column k - hb - 1.  ~150 cycles per step instead of three LDS round trips and a 21-deep chain (1 600).
        // G and chat do not depend on dp: all threads form them first -- thread (k', j, r) one column of G_{k',j}
        // (G[c][r] = sum_{m >= c} L_{k'+j,k'}[r][m] L_k'k'^-1[m][c]) from the factor store, into LDS by the step k = k' + j that uses it.
        const int per_step = hb * 36, cap = (B.lds_bytes - (3 * n + 36 * nb) * 8) / (per_step * 8);
        const int kfac = (tw && side) ? own : nb;              // columns this side has factored (side 1: not the middle)
        const int lane = tid, s_ = lane / 6, c_ = lane - 6 * s_;
        const bool act = tid < hb1 * 6;
        auto linv_times = [&](const double *Li, const double (&o)[6], double (&gq)[6]) {       // gq[c] = sum_{m >= c} o[m] L^-1[m][c]
            double li[21];
            {
                int q = 0;
#pragma unroll
                for (int m = 0; m < 6; m++)
#pragma unroll
                    for (int c = 0; c <= m; c++) li[q++] = Li[m * 6 + c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double t = 0.0;
#pragma unroll
                for (int m = c; m < 6; m++) t += o[m] * li[m * (m + 1) / 2 + c];
                gq[c] = t;
            }
        };
        long long trb0 = B.trace ? clock64() : 0;
        for (int k = tid; k < kfac; k += BS_T) {
            double o[6], gq[6];
#pragma unroll
            for (int m = 0; m < 6; m++) o[m] = x[6 * k + m];
            linv_times(LiAll + 36 * k, o, gq);
#pragma unroll
            for (int c = 0; c < 6; c++) chat[6 * k + c] = gq[c];
        }
        __syncthreads();
        if (B.trace && side == 0 && tid == 0) { const long long t_ = clock64(); B.trace[10] = t_ - trb0; trb0 = t_; }
        double base = 0.0;                                     // - sum_j G_{k',j} dp_{k'+j} so far, of the column in this lane's slot
        for (int kb = nb; kb > 0; ) {
            const int ka = kb > cap ? kb - cap : 0;               // steps ka .. kb - 1 (step 0 has nothing to update: its G rows are never read)
            const int tot = (kb - ka) * hb * 6;
            for (int e0 = tid; e0 < tot; e0 += 3 * BS_T) {      // three items per thread in flight: the rows of L are L2 round trips
                double o[3][6];
                int ob[3], kq[3];
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const int e = e0 + b * BS_T;
                    const int blk = e / 6, r = e - 6 * blk, st = blk / hb, j = blk - st * hb + 1, kp = ka + st - j;
                    const bool ok = e < tot && kp >= 0 && kp < kfac;
                    ob[b] = ok ? blk * 36 + r : -1; kq[b] = ok ? kp : 0;
                    ld_rec<6>(Lg + (size_t)kq[b] * lgs + (ok ? j * 36 + r * 6 : 0), o[b]);
                }
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    if (ob[b] < 0) continue;
                    double gq[6];
                    linv_times(LiAll + 36 * kq[b], o[b], gq);
#pragma unroll
                    for (int c = 0; c < 6; c++) Gs[ob[b] + c * 6] = gq[c];
                }
            }
            __syncthreads();
            if (tw && side && kb == nb) {
                // the middle poses are the other side's: wait for their dp, enter it where this side's recurrence expects chat (their
                // running sums stay zero: `on` below never selects a middle column)
                if (tid == 0) while (__hip_atomic_load(B.fail + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != B.epoch) __builtin_amdgcn_s_sleep(2);
                __syncthreads();
                if (same_xcd()) asm volatile("" ::: "memory"); else __threadfence();
                for (int e = tid; e < hb * 6; e += BS_T) chat[6 * own + e] = __hip_atomic_load(dpo + 6 * gi(own + e / 6) + e % 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
            }
            if (B.trace && side == 0 && tid == 0) { const long long t_ = clock64(); B.trace[11] = t_ - trb0; trb0 = t_; }
            if (tid < 64) {
                // two steps per trip with the roles of the two register sets swapped: the rows / chat entries of the next step are
                // requested before this step's chain and nothing waits for them until they are used
                // Per lane (slot s_, row c_) the state moves by one step per request: jn = distance of the lane's slot from the step's slot
                // (0: the lane holds dp of that step), gp = row c_ of G_{k - jn, jn} in the staged blocks -- (k, jn) -> (k - 1, jn - 1) is
                // hb + 1 blocks back, and when the slot wraps (jn: 0 -> hb) the address stays --, cp = the step's chat entry.
                const int k_first = kb - 1;
                int jn = k_first % hb1 - s_; if (jn < 0) jn += hb1;
                int go = ((k_first - ka) * hb + jn - 1) * 36 + (act ? c_ : 0) * 6;      // offset in Gs (jn = 0: one block before the step's first; never read)
                int co = 6 * k_first + (act ? c_ : 0);                                   // offset in chat  (offsets, not pointers: loop-carried pointers lose their address space)
                auto load_next = [&](int k, double (&gg)[6], double &cc, bool &on, bool &mine) {   // for step k: row c_ of G_{k - jn, jn}, chat_k
                    mine = act && jn == 0;
                    on = act && jn > 0 && k - jn >= 0 && k - jn < kfac;
                    ld_rec<6>(Gs + (on ? go : 0), gg);
                    cc = chat[co];
                    go -= jn != 0 ? hb1 * 36 : 0; co -= 6;
                    jn = jn == 0 ? hb : jn - 1;
                };
                const bool same0 = tw && side == 0 && same_xcd();
                auto step = [&](int k, int ksl6, const double (&gg)[6], double cc, bool on, bool mine) {
                    const double v = mine ? base + cc : base;         // dp_k on the lanes of its slot
                    double dv[6];
#pragma unroll
                    for (int m = 0; m < 6; m++) {
                        const int lo = __builtin_amdgcn_readlane(__double2loint(v), ksl6 + m);
                        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), ksl6 + m);
                        dv[m] = __hiloint2double(hi, lo);
                    }
                    const double t = (fma(gg[1], dv[1], gg[0] * dv[0]) + fma(gg[3], dv[3], gg[2] * dv[2])) + fma(gg[5], dv[5], gg[4] * dv[4]);
                    if (mine) x[6 * k + c_] = v;
                    base = mine ? 0.0 : (on ? base - t : base);
                    if (tw && side == 0 && k >= own) {               // the middle: the other side waits for these
                        if (mine) __hip_atomic_store(dpo + 6 * k + c_, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (k == own) {
                            if (same0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else __threadfence();
                            if (lane == 0) __hip_atomic_store(B.fail + 2, B.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                };
                // three register sets in rotation: the rows / chat entry of step k - 2 are requested at the start of step k
                int ks6 = (k_first % hb1) * 6;                       // first lane of the step's slot
                auto dec6 = [&](int q) { return q == 0 ? hb * 6 : q - 6; };
                double gA[6], gB[6], gC[6], cA = 0.0, cB = 0.0, cC = 0.0; bool onA = false, onB = false, onC = false, mA = false, mB = false, mC = false;
#pragma unroll
                for (int m = 0; m < 6; m++) { gA[m] = 0.0; gB[m] = 0.0; gC[m] = 0.0; }
                load_next(kb - 1, gA, cA, onA, mA);
                if (kb - 2 >= ka) load_next(kb - 2, gB, cB, onB, mB);
                for (int k = kb - 1; k >= ka; k -= 3) {
                    const int ks1 = dec6(ks6), ks2 = dec6(ks1);
                    if (k - 2 >= ka) load_next(k - 2, gC, cC, onC, mC);
                    step(k, ks6, gA, cA, onA, mA);
                    if (k - 1 >= ka) {
                        if (k - 3 >= ka) load_next(k - 3, gA, cA, onA, mA);
                        step(k - 1, ks1, gB, cB, onB, mB);
                    }
                    if (k - 2 >= ka) {
                        if (k - 4 >= ka) load_next(k - 4, gB, cB, onB, mB);
                        step(k - 2, ks2, gC, cC, onC, mC);
                    }
                    ks6 = dec6(ks2);
                }
            }
            if (B.trace && side == 0 && tid == 0) { const long long t_ = clock64(); B.trace[12] = t_ - trb0; trb0 = t_; }
            __syncthreads();
            kb = ka;
        }
    } else {
    if (nb > 0) fetch_back(nb - 1);
    for (int k = nb - 1; k >= 0; k--) {
        const int np = nb - 1 - k < hb ? nb - 1 - k : hb;
        if (tid < np * 6) {
            const int di = tid / 6 + 1;
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 6; r++) t += lreg[r] * x[6 * (k + di) + r];
            part[tid] = t;                                   // [(di - 1) * 6 + c]
        }
        double li[6];
#pragma unroll
        for (int m = 0; m < 6; m++) li[m] = lic[m];
        if (k > 0) fetch_back(k - 1);
        bs_barrier();
        if (tid >= BS_T - 6) {
            const int c = tid - (BS_T - 6);
            double a = x[6 * k + c], pv[BS_MAXHB];
#pragma unroll
            for (int di = 0; di < BS_MAXHB; di++) pv[di] = di < np ? part[di * 6 + c] : 0.0;    // all reads in flight, then the ordered sum
#pragma unroll
            for (int di = 0; di < BS_MAXHB; di++) if (di < np) a -= pv[di];
            tv[c] = a;
        }
        bs_barrier();
        if (tid >= BS_T - 6) {
            const int c = tid - (BS_T - 6);
            double a = 0.0;
#pragma unroll
            for (int m = 0; m < 6; m++) a += (m >= c ? li[m] : 0.0) * tv[m];
            x[6 * k + c] = a;
        }
        bs_barrier();
    }
    }
    if (warm_acc == 1.2345e-300) x[0] = warm_acc;             // keeps the warm-up loads alive; never true in practice
    for (int a = tid; a < 6 * kfac2; a += BS_T) dpo[6 * gi(a / 6) + a % 6] = x[a];
    if (tid == 0) { if (!tw || side == 0) *B.fail = s_bad; if (s_bad) d.st->chol_fail = 1; }
    if (B.trace && side == 0) {
        if (tid == 0) { B.trace[0] = tr0; B.trace[1] = trD; B.trace[3] = clock64() - trT; B.trace[4] = trA; B.trace[5] = trB; }
        if (tid == 64) { B.trace[2] = trW; B.trace[6] = trC; B.trace[7] = trD; }
        if (tid == 192) { B.trace[8] = trW; B.trace[9] = trC; }
    }
}
__global__ __launch_bounds__(BS_T) void k_band_solve(BADev d, BandArgs B, int use_state) { band_solve_body(d, B, use_state); }

// ---- small windows that NO pose order makes banded (half-bandwidth > BS_MAXHB: every map point seen by almost every key-frame): the
//      damped reduced system of <= DS_MAXF free-span poses is factored DENSE by one workgroup, lower triangle of 6 x 6 blocks resident in
//      LDS (block (i, j), j <= i, at i (i + 1) / 2 + j; 134 KB at 30 poses), the right-hand side riding along as one more row:
//        per block column k:  wave 0 factors A_kk in place (wave-synchronous, six pivots);
//                             thread (i, r) forms row r of the panel block L_ik = A_ik L_kk^-T by forward substitution (and y_k likewise);
//                             thread (pair (i, j), r) updates row r of A_ij -= L_ik L_jk^T (the right-hand side: y_j -= L_jk y_k);
//        then L^T dp = y block row by block row, bottom up.
//      (P = 26, 25 free poses: 81 us -- factor wave 2.8 k cycles per column beside a trailing update of 4.2 k, bound by the LDS pipe; load 6 us,
//      back-substitution 9 us.)  One launch instead of the 7 + n / 32 of the tiled path (k_chol_prepare ... k_chol_backsolve: 164 us); the build of such a window goes
//      through the point groups like a banded one (window = the whole triangle).  Same damping rule as k_chol_prepare / k_band_solve.
#define DS_T 512
#define DS_MAXF 30
static size_t dense_lds_bytes(int F) { return ((size_t)F * (F + 1) / 2 * 36 + (size_t)6 * F * 2 + (size_t)36 * F + 64) * 8; }
__device__ __forceinline__ void ds_barrier() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
__global__ __launch_bounds__(DS_T) void k_dense_solve(BADev d, BandArgs B, int use_state)
{
    if (use_state && d.st->converged) return;
    extern __shared__ __attribute__((aligned(16))) double ds_sm[];
    __shared__ int s_bad;
    const int F = B.nb, n = d.n, ns = 6 * F, tid = threadIdx.x, nblk = F * (F + 1) / 2;
    double *A = ds_sm;                                   // [nblk][36]
    double *y = A + (size_t)nblk * 36;                   // [ns]: g -> y -> dp
    double *idg = y + ns;                                // [ns]: 1 / L_jj
    const double inv_delta = use_state ? 1.0 / d.st->delta : B.inv_delta_host;
    if (tid == 0) s_bad = 0;
    // ---- load: EVERY element is requested before the first one is stored (S was written by other kernels from all eight XCDs: first
    //      touches are trips to memory, and a load-store loop pays one after the other -- 25 round trips were half of the kernel).  Block
    //      (i, j) entry (r, c) is read through its symmetric twin S[6 j + c, 6 i + r] so that consecutive lanes read consecutive addresses
    constexpr int NL = (DS_MAXF * (DS_MAXF + 1) / 2 * 36 + DS_T - 1) / DS_T;
    {
        double v[NL];
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int e = tid + u * DS_T;
            v[u] = 0.0;
            if (e < nblk * 36) {
                const int blk = e / 36, q = e - 36 * blk, r = q / 6, c = q - 6 * r;
                int i = (int)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
                while (i * (i + 1) / 2 > blk) i--;
                while ((i + 1) * (i + 2) / 2 <= blk) i++;
                const int j = blk - i * (i + 1) / 2;
                v[u] = B.S[(size_t)(6 * j + c) + (size_t)(6 * i + r) * n];
            }
        }
        const double udv = tid < ns ? B.ud[tid] : 0.0, gv = tid < ns ? B.g[tid] : 0.0;
#pragma unroll
        for (int u = 0; u < NL; u++) { const int e = tid + u * DS_T; if (e < nblk * 36) A[e] = v[u]; }
        if (tid < ns) y[tid] = gv;
        ds_barrier();
        if (tid < ns) { const int k = tid / 6, r = tid - 6 * k; A[(size_t)(k * (k + 1) / 2 + k) * 36 + 7 * r] += fmin(fmax(udv, LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta; }
    }
    ds_barrier();
    // Cholesky of the diagonal block (k, k) by wave 0.  upd: the block first loses L_{k,k-1} L_{k,k-1}^T (the previous column's trailing update
    // for this one block -- the look-ahead: the wave factors block k while the other waves update the rest), lane q < 21 forms lower entry q.
    // Every lane then holds the whole triangle (42 v_readlane) and eliminates it division-free on scaled entries, k_band_solve's scheme:
    // m_ik <- m_ik p_j - m_ij m_kj (two dependent operations per pivot; every second pivot a power-of-two rescale), the six 1 / sqrt from
    // v_rsq_f64 + one Newton step, independent of each other: ~1.5 k cycles per block instead of ~5 k for the textbook loop on one wave
    // (a dependent Float64 operation of a lone wave costs 36 cycles).
    auto factor_diag = [&](int k, bool upd) {
#pragma clang fp contract(fast)
        double *D = A + (size_t)(k * (k + 1) / 2 + k) * 36;
        const int q = tid < 21 ? tid : 20;
        int r = 0; while ((r + 1) * (r + 2) / 2 <= q) r++;
        const int c = q - r * (r + 1) / 2;
        double e = D[6 * r + c];
        if (upd) {
            const double *Lp = A + (size_t)(k * (k + 1) / 2 + k - 1) * 36;
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 6; m++) acc += Lp[6 * r + m] * Lp[6 * c + m];
            e -= acc;
        }
        double M[21], ps[6], sj[6], Lr[21];
#pragma unroll
        for (int t = 0; t < 21; t++) M[t] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(e), t), __builtin_amdgcn_readlane(__double2loint(e), t));
        double sc = 1.0; bool bad = false;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double pj = M[j * (j + 1) / 2 + j];
            bad = bad || !(pj > 0.0 && pj < 1e300);
            pj = (pj > 0.0 && pj < 1e300) ? pj : 1.0;
            ps[j] = pj * sc; sj[j] = sc;
            Lr[j * (j + 1) / 2 + j] = pj;
#pragma unroll
            for (int i = j + 1; i < 6; i++) Lr[i * (i + 1) / 2 + j] = M[i * (i + 1) / 2 + j];      // (unscaled column: times rsqrt(p_j s_j) below)
            if ((j & 1) == 0) {
                const int ex = -__builtin_amdgcn_frexp_exp(pj);
                sc *= __builtin_amdgcn_frexp_mant(pj);
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int k2 = j + 1; k2 <= i; k2++)
                        M[i * (i + 1) / 2 + k2] = __builtin_amdgcn_ldexp(M[i * (i + 1) / 2 + k2] * pj - M[i * (i + 1) / 2 + j] * M[k2 * (k2 + 1) / 2 + j], ex);
            } else {
                sc *= pj;
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int k2 = j + 1; k2 <= i; k2++)
                        M[i * (i + 1) / 2 + k2] = M[i * (i + 1) / 2 + k2] * pj - M[i * (i + 1) / 2 + j] * M[k2 * (k2 + 1) / 2 + j];
            }
        }
        double out = 0.0, iq = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double y0 = __builtin_amdgcn_rsq(ps[j]);
            const double r0 = fma(-(ps[j] * y0), y0, 1.0), rd = fma(y0 * 0.5, r0, y0);
            if (tid == j) iq = sj[j] * rd;                        // 1 / L_jj = s_j rsqrt(p_j s_j)
#pragma unroll
            for (int i = j; i < 6; i++) if (q == i * (i + 1) / 2 + j) out = Lr[i * (i + 1) / 2 + j] * rd;
        }
        if (tid < 21) D[6 * r + c] = out;
        if (tid < 6) idg[6 * k + tid] = iq;
        if (bad && tid == 0) s_bad = 1;
    };
    if (tid < 64) factor_diag(0, false);
    ds_barrier();
    for (int k = 0; k < F; k++) {
        const double *D = A + (size_t)(k * (k + 1) / 2 + k) * 36;
        // (1) panel rows: x L_kk^T = a (forward substitution), item = (block row i > k, row r) or the right-hand side's block k
        const int npan = (F - 1 - k) * 6 + 1;
        for (int it = tid; it < npan; it += DS_T) {
            double *row = it < npan - 1 ? A + (size_t)((k + 1 + it / 6) * (k + 2 + it / 6) / 2 + k) * 36 + 6 * (it % 6) : y + 6 * k;
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double a = row[c];
#pragma unroll
                for (int m = 0; m < 6; m++) if (m < c) a -= x[m] * D[6 * c + m];
                x[c] = a * idg[6 * k + c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) row[c] = x[c];
        }
        ds_barrier();
        // (2) wave 0: the next diagonal block, updated and factored (look-ahead); waves 1-7: the trailing update A_ij -= L_ik L_jk^T of every
        //     other pair k < j <= i, item = (pair, row), and the right-hand side y_j -= L_jk y_k, item = block row j
        if (tid < 64) { if (k + 1 < F) factor_diag(k + 1, true); }
        else {
            // (measured: items of one whole block pair -- L_jk in registers for its six rows, 144 instead of 324 LDS accesses per block -- are
            //  slower: 126 k vs 104 k cycles per solve; past the first columns there are fewer pairs than lanes and a thread's six rows are a chain)
            const int m1 = F - 1 - k, npair = m1 * (m1 + 1) / 2, nupd = npair * 6 + m1;
            for (int it = tid - 64 + 6; it < nupd; it += DS_T - 64) {      // (items 0 .. 5 are the rows of pair (k + 1, k + 1): wave 0's)
                if (it < npair * 6) {
                    const int pr = it / 6, r = it - 6 * pr;
                    int a = (int)((sqrtf(8.0f * (float)pr + 1.0f) - 1.0f) * 0.5f);
                    while (a * (a + 1) / 2 > pr) a--;
                    while ((a + 1) * (a + 2) / 2 <= pr) a++;
                    const int b = pr - a * (a + 1) / 2, i = k + 1 + a, j = k + 1 + b;          // b <= a: j <= i
                    const double *Li = A + (size_t)(i * (i + 1) / 2 + k) * 36 + 6 * r, *Lj = A + (size_t)(j * (j + 1) / 2 + k) * 36;
                    double *T = A + (size_t)(i * (i + 1) / 2 + j) * 36 + 6 * r;
                    double li[6];
#pragma unroll
                    for (int m = 0; m < 6; m++) li[m] = Li[m];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double acc = 0.0;
#pragma unroll
                        for (int m = 0; m < 6; m++) acc += li[m] * Lj[6 * c + m];
                        T[c] -= acc;
                    }
                } else {
                    const int j = k + 1 + (it - npair * 6);
                    const double *Lj = A + (size_t)(j * (j + 1) / 2 + k) * 36;
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double acc = 0.0;
#pragma unroll
                        for (int m = 0; m < 6; m++) acc += Lj[6 * c + m] * y[6 * k + m];
                        y[6 * j + c] -= acc;
                    }
                }
            }
        }
        ds_barrier();
    }
    // ---- L^T dp = y, bottom up: dp_k = L_kk^-T (y_k - sum_{i > k} L_ik^T dp_i).  The diagonal blocks are inverted all at once first
    //      (thread = (block, column of the inverse): six forward substitutions each, in parallel), then wave 0 alone walks the block rows --
    //      lane c forms dp_k[c] from the inverse, the lanes share out the 6 k entries of y above it: no workgroup barrier in the chain
    double *Linv = idg + ns;                             // [F][36] L_kk^-1 (lower, row-major)
    for (int it = tid; it < 6 * F; it += DS_T) {
        const int k = it / 6, c = it - 6 * k;
        const double *D = A + (size_t)(k * (k + 1) / 2 + k) * 36;
        double x[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double a = r == c ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < 6; m++) if (m < r) a -= D[6 * r + m] * x[m];
            x[r] = r < c ? 0.0 : a * idg[6 * k + r];
        }
#pragma unroll
        for (int r = 0; r < 6; r++) Linv[36 * k + 6 * r + c] = x[r];
    }
    ds_barrier();
    if (tid < 64) {
        for (int k = F - 1; k >= 0; k--) {
            if (tid < 6) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) acc += Linv[36 * k + 6 * m + tid] * y[6 * k + m];      // (L^-T y)[c] = sum_m Linv[m][c] y[m]
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                y[6 * k + tid] = acc;
            } else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int it = tid; it < 6 * k; it += 64) {        // y_j[c] -= sum_m L_kj[m][c] dp_k[m], j < k
                const int j = it / 6, c = it - 6 * j;
                const double *Lk = A + (size_t)(k * (k + 1) / 2 + j) * 36;
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) acc += Lk[6 * m + c] * y[6 * k + m];
                y[it] -= acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    ds_barrier();
    double *const dpo = d.dp + 6 * B.p0;
    for (int a = tid; a < ns; a += DS_T) dpo[a] = y[a];
    if (tid == 0) { *B.fail = s_bad; if (s_bad) d.st->chol_fail = 1; }
}

__global__ __launch_bounds__(256) void k_backsub(BADev d, int use_state)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    __shared__ double sh[4];
    if (use_state && d.st->converged) return;
    const int kk = blockIdx.x * 256 + threadIdx.x, M = d.M;
    double mx = 0.0;
    if (kk < M) {
        const int j = d.pt_id[kk];
        double bl[3] = {d.bl[j], d.bl[(size_t)M + j], d.bl[(size_t)2 * M + j]};
        for (int i = d.pt_start[kk]; i < d.pt_start[kk + 1]; i++) {
            if (!d.hasp[i]) continue;
            const double *dp = d.dp + 6 * d.opose[i];
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) { a += d.Jp[(size_t)i * 12 + k] * dp[k]; b += d.Jp[(size_t)i * 12 + 6 + k] * dp[k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) bl[k] -= d.Jl[(size_t)i * 6 + k] * a + d.Jl[(size_t)i * 6 + 3 + k] * b;
        }
        double Vi[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Vi[k] = d.Vinv[(size_t)k * M + j];
        const double l0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
        const double l1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
        const double l2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
        d.dl[3 * j] = l0; d.dl[3 * j + 1] = l1; d.dl[3 * j + 2] = l2;
        pb.pts_t[3 * j] = pb.pts[3 * j] - l0; pb.pts_t[3 * j + 1] = pb.pts[3 * j + 1] - l1; pb.pts_t[3 * j + 2] = pb.pts[3 * j + 2] - l2;
        mx = fmax(fabs(l0), fmax(fabs(l1), fabs(l2)));
    }
    if (kk < d.n) { pb.pose_t[kk] = pb.pose[kk] - d.dp[kk]; mx = fmax(mx, fabs(d.dp[kk])); }
    const double t = block_max(mx, sh);
    if (threadIdx.x == 0) d.part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_trial(BADev d, int ignore_outliers, int use_state, int nb_pts)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    __shared__ double sh[4];
    if (use_state && d.st->converged) return;
    const int i = blockIdx.x * 256 + threadIdx.x, O = d.O;
    double st = 0.0, sp = 0.0;
    if (i < O) {
        const int p = d.opose[i], j = d.opoint[i];
        double r[2] = {0.0, 0.0};
        if (!(ignore_outliers && d.outl[i])) {
            const double X[3] = {pb.pts_t[3 * j], pb.pts_t[3 * j + 1], pb.pts_t[3 * j + 2]};
            double pose[6];
#pragma unroll
            for (int k = 0; k < 6; k++) pose[k] = pb.pose_t[6 * p + k];
            obs_eval(pose, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
        }
        double a = 0.0, b = 0.0;
        const double *dp = d.dp + 6 * p, *dl = d.dl + 3 * j;
#pragma unroll
        for (int k = 0; k < 6; k++) { a += d.Jp[(size_t)i * 12 + k] * dp[k]; b += d.Jp[(size_t)i * 12 + 6 + k] * dp[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { a += d.Jl[(size_t)i * 6 + k] * dl[k]; b += d.Jl[(size_t)i * 6 + 3 + k] * dl[k]; }
        a -= d.f[2 * (size_t)i]; b -= d.f[2 * (size_t)i + 1];
        st = r[0] * r[0] + r[1] * r[1];
        sp = a * a + b * b;
    }
    const double t1 = block_sum(st, sh);
    const double t2 = block_sum(sp, sh);
    if (threadIdx.x == 0) { d.part[nb_pts + 2 * blockIdx.x] = t1; d.part[nb_pts + 2 * blockIdx.x + 1] = t2; }
}

// k_backsub + k_trial on the point groups of k_schur_groups (one workgroup per group, dp in LDS): thread = observation forms
// Jl' (Jp dp), thread = point sums them in observation order, dl = V^-1 (bl - sum), trial point; thread = observation again: trial
// and predicted residual.  Partials: part[g] = max |dx|, part[ngrp + 2 g] = trial cost, part[ngrp + 2 g + 1] = predicted cost.
// s_dp [n], s_u [observations x 3], s_dl [points x 6: dl (3), trial point (3)], s_red [8], s_sct [n: sin / cos of every TRIAL pose's angles]: LDS of the
// caller (static arrays of the largest sizes in the single-window kernel; a batch carves them from dynamic LDS at its own sizes -- 36 KB of static
// arrays held k_update_groups_b to four workgroups per compute unit).  sct_ready: pb.sc_t holds the trial poses' sin / cos already (k_trial_poses_b).
// RECOMP (batches on the matrix-core build): the Jacobians and the residual at the committed parameters are formed HERE again instead of being stored by the
// build and read back (160 bytes per observation each way: the kernel was bound by those reads, 0.97 GB per launch of 128 x P20); s_sc = the committed poses'
// sin / cos.  Same function of the same arguments as in the build: the same bits.
template <int TT, bool RECOMP = false>
__device__ __forceinline__ void update_groups_body(const BADev &d, int ignore_outliers, int use_state, double *s_dp, double *s_u, double *s_dl, double *s_red, double *s_sct, bool sct_ready, double *s_sc = nullptr)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    if (use_state && d.st->converged) return;
    const int tid = threadIdx.x, M = d.M, O = d.O, n = d.n;
    const int4 G = d.grp[blockIdx.x];
    const int k0 = G.x, o0 = G.y, npts = G.z >> 16, nobs = G.w;
    for (int a = tid; a < n; a += TT) s_dp[a] = d.dp[a];
    if (sct_ready) { for (int a = tid; a < n; a += TT) s_sct[a] = pb.sc_t[a]; }
    if (RECOMP) { for (int a = tid; a < n; a += TT) s_sc[a] = pb.sc[a]; }
    lds_sync();
    if (!sct_ready)
        for (int q = tid; q < d.P; q += TT) {
            const double tp[3] = {pb.pose[6 * q] - s_dp[6 * q], pb.pose[6 * q + 1] - s_dp[6 * q + 1], pb.pose[6 * q + 2] - s_dp[6 * q + 2]};
            pose_sincos(tp, s_sct + 6 * q);
        }
    double mx = 0.0;
    if (blockIdx.x == 0)
        for (int a = tid; a < n; a += TT) { const double v = s_dp[a]; pb.pose_t[a] = pb.pose[a] - v; mx = fmax(mx, fabs(v)); }
    const int i = o0 + tid;
    int p = 0, pl = 0;
    double jp[12], jl[6], ff[2] = {0.0, 0.0}, a = 0.0, b = 0.0;
    bool active = false;
    if (tid < nobs) {
        p = d.opose[i]; pl = d.opk[i] - k0;
        active = !(ignore_outliers && d.outl[i]);
        if (RECOMP) {
#pragma unroll
            for (int k = 0; k < 12; k++) jp[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) jl[k] = 0.0;
            if (active) {
                const int j = d.opoint[i];
                const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                double sc[6], tr[3];
#pragma unroll
                for (int k = 0; k < 6; k++) sc[k] = s_sc[6 * p + k];
#pragma unroll
                for (int k = 0; k < 3; k++) tr[k] = pb.pose[6 * p + 3 + k];
                obs_eval_sc(sc, tr, X, d.pix[i], d.pix[O + i], d.cam, ff, jp, jl, nullptr);
                if (d.pconst[p]) {
#pragma unroll
                    for (int k = 0; k < 12; k++) jp[k] = 0.0;
                }
            }
        } else {
        if (d.hasp[i]) ld_rec<12>(d.Jp + (size_t)i * 12, jp);
        else {
#pragma unroll
            for (int k = 0; k < 12; k++) jp[k] = 0.0;                                                           // Jp = 0 unless the observation has a free pose (not stored then)
        }
        ld_rec<6>(d.Jl + (size_t)i * 6, jl); ld_rec<2>(d.f + 2 * (size_t)i, ff);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) { a += jp[k] * s_dp[6 * p + k]; b += jp[6 + k] * s_dp[6 * p + k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) s_u[tid * 3 + k] = jl[k] * a + jl[3 + k] * b;
    }
    lds_sync();
    if (tid < npts) {
        const int kk = k0 + tid, j = d.pt_id[kk];
        double bl[3] = {d.bl[j], d.bl[(size_t)M + j], d.bl[(size_t)2 * M + j]};
        const int t0 = d.pt_start[kk] - o0, t1 = d.pt_start[kk + 1] - o0;
        for (int t = t0; t < t1; t++) {
#pragma unroll
            for (int k = 0; k < 3; k++) bl[k] -= s_u[t * 3 + k];
        }
        double Vi[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Vi[k] = d.Vinv[(size_t)k * M + j];
        const double l0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
        const double l1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
        const double l2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
        const double X0 = pb.pts[3 * j] - l0, X1 = pb.pts[3 * j + 1] - l1, X2 = pb.pts[3 * j + 2] - l2;
        d.dl[3 * j] = l0; d.dl[3 * j + 1] = l1; d.dl[3 * j + 2] = l2;
        pb.pts_t[3 * j] = X0; pb.pts_t[3 * j + 1] = X1; pb.pts_t[3 * j + 2] = X2;
        s_dl[tid * 6] = l0; s_dl[tid * 6 + 1] = l1; s_dl[tid * 6 + 2] = l2;
        s_dl[tid * 6 + 3] = X0; s_dl[tid * 6 + 4] = X1; s_dl[tid * 6 + 5] = X2;
        mx = fmax(mx, fmax(fabs(l0), fmax(fabs(l1), fabs(l2))));
    }
    lds_sync();
    double st = 0.0, sp = 0.0;
    if (tid < nobs) {
        double r[2] = {0.0, 0.0};
        const double *dl = s_dl + pl * 6;
        if (active) {
            const double X[3] = {dl[3], dl[4], dl[5]};
            double sc[6], tr[3];
#pragma unroll
            for (int k = 0; k < 6; k++) sc[k] = s_sct[6 * p + k];
#pragma unroll
            for (int k = 0; k < 3; k++) tr[k] = pb.pose[6 * p + 3 + k] - s_dp[6 * p + 3 + k];
            obs_eval_sc(sc, tr, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { a += jl[k] * dl[k]; b += jl[3 + k] * dl[k]; }
        a -= ff[0]; b -= ff[1];
        st = r[0] * r[0] + r[1] * r[1];
        sp = a * a + b * b;
    }
    const double t1 = block_sum_lds(st, s_red);
    const double t2 = block_sum_lds(sp, s_red);
    const double t3 = block_max_lds(mx, s_red);
    if (tid == 0) { d.part[blockIdx.x] = t3; d.part[d.ngrp + 2 * blockIdx.x] = t1; d.part[d.ngrp + 2 * blockIdx.x + 1] = t2; }
}
__global__ __launch_bounds__(SG_T) void k_update_groups(BADev d, int ignore_outliers, int use_state)
{
    __shared__ double s_dp[SOLVE_MAX_N], s_u[SG_OB * 3], s_dl[SG_SB * 6], s_red[8], s_sct[SOLVE_MAX_N];
    update_groups_body<SG_T>(d, ignore_outliers, use_state, s_dp, s_u, s_dl, s_red, s_sct, false);
}

// Sums the partials (fixed order) and, in the single-GPU path, runs the
// LeastSquaresOptim accept/reject logic.  mode 0: ssr of the current residuals
// (after k_linearize); mode 1: trial/predicted/maxdx -> state (+ LM decision if lm).
// fixed-order strided sum / max of a partials array by one 256-thread workgroup
__device__ __forceinline__ double ctl_sum(const double *p, int n, int stride, double *sh)
{
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) t += p[(size_t)i * stride];
    return block_sum(t, sh);
}
__device__ __forceinline__ double ctl_max(const double *p, int n, double *sh)
{
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) t = fmax(t, p[i]);
    return block_max(t, sh);
}
// LeastSquaresOptim's accept / reject of a trial step (trust-region radius update, step-quality test): t = trial cost,
// p = predicted cost, mx = max |dx|
__device__ __forceinline__ void lm_decide(LMState *s, double t, double p, double mx)
{
    s->iters++;
    if (s->chol_fail) { s->converged = 1; s->accept = 0; return; }
    const double ssr = s->ssr;
    const double rho = (t - ssr) / (p - ssr);
    if (rho > LM_MIN_STEP_QUALITY) {
        const int x_conv = mx <= LM_XTOL;
        const int f_conv = fabs(ssr - t) / (fabs(ssr) + LM_FTOL) <= LM_FTOL;
        s->ssr = t;
        const double u = 2.0 * rho - 1.0;
        s->delta = fmin(s->delta / fmax(1.0 / 3.0, 1.0 - u * u * u), LM_MAX_DELTA);
        s->decrease_factor = 2.0;
        s->accept = 1;
        s->cur ^= 1;                                         // the trial parameters become the committed ones
        s->converged = x_conv || f_conv;
    } else {
        s->delta = fmax(s->delta / s->decrease_factor, LM_MIN_DELTA);
        s->decrease_factor *= 2.0;
        s->accept = 0;
        s->converged = mx <= LM_XTOL;
    }
}
__device__ __forceinline__ void control_body(const BADev &d, int mode, int nb_obs, int nb_pts, int lm, double *out4)
{
    __shared__ double sh[4];
    LMState *s = d.st;
    if (mode == 0) {
        const double t = ctl_sum(d.part, nb_obs, 1, sh);
        if (threadIdx.x == 0) { s->ssr = t; if (out4) out4[0] = t; }
        return;
    }
    const int paced = lm >> 1;                               // bit 1: the LM state lives on the device (device-paced paths): nothing runs after convergence
    lm &= 1;
    if ((lm || paced) && s->converged) return;               // (the sharded path used to overwrite trial_ssr / maxdx with this shard's stale LOCAL sums here)
    const double mx = ctl_max(d.part, nb_pts, sh);
    const double t = ctl_sum(d.part + nb_pts, nb_obs, 2, sh);
    const double p = ctl_sum(d.part + nb_pts + 1, nb_obs, 2, sh);
    if (threadIdx.x != 0) return;
    s->trial_ssr = t; s->pred_ssr = p; s->maxdx = mx;
    if (out4) { out4[0] = t; out4[1] = p; out4[2] = mx; out4[3] = (double)s->chol_fail; }
    if (!lm) return;
    lm_decide(s, t, p, mx);
}
__global__ __launch_bounds__(256) void k_control(BADev d, int mode, int nb_obs, int nb_pts, int lm, double *out4) { control_body(d, mode, nb_obs, nb_pts, lm, out4); }

// The sharded path: every rank's [trial_ssr, pred_ssr, max|dx|, chol_fail] gathered into g (nranks x 4).  Sums / maxima in
// rank order, then the same decision as the single-GPU path -- identical on every rank, taken on the device.
__global__ void k_control_gathered(BADev d, const double *g, int nranks)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LMState *s = d.st;
    if (s->converged) return;
    double t = 0.0, p = 0.0, mx = 0.0, cf = 0.0;
    for (int r = 0; r < nranks; r++) { t += g[4 * r]; p += g[4 * r + 1]; mx = fmax(mx, g[4 * r + 2]); cf = fmax(cf, g[4 * r + 3]); }
    s->trial_ssr = t; s->pred_ssr = p; s->maxdx = mx;
    if (cf != 0.0) s->chol_fail = 1;
    lm_decide(s, t, p, mx);
}
// start of an LM pass in the sharded path: the all-reduced cost of the current parameters comes from the reduce buffer
__global__ void k_lm_start(BADev d, const double *ssr_slot, int first_pass)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LMState *s = d.st;
    s->ssr = *ssr_slot;
    if (first_pass) { s->ssr_init = s->ssr; s->chol_fail = 0; }
    s->delta = LM_DELTA0; s->decrease_factor = 2.0; s->converged = 0; s->accept = 0; s->iters = 0;
}

// host-paced protocol (slam_ba_commit): the host has decided -- an accepted step swaps the two parameter buffers.  (The device-paced
// paths swap inside lm_decide: no launch at all.)
__global__ void k_commit(BADev d, int accept_host)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && accept_host) d.st->cur ^= 1;
}

__global__ void k_lm_reset(BADev d, int pass)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LMState *s = d.st;
    if (pass == 0) { s->ssr_init = s->ssr; s->chol_fail = 0; s->n_outliers = 0; }
    if (pass == 3) { s->ssr_pass1 = s->ssr; s->iters_pass1 = s->iters; return; }   // record the end of pass 1
    if (pass == 2) { s->ssr_final = s->ssr; s->iters_pass2 = s->iters; return; }   // record the end of pass 2
    s->delta = LM_DELTA0; s->decrease_factor = 2.0; s->converged = 0; s->accept = 0; s->iters = 0;
}

// _ba_detect_outliers!, bundle_adjustment.jl:90-111
__device__ __forceinline__ void outliers_body(const BADev &d, double repr_eps, double depth_eps)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    __shared__ double sh[4];
    const int i = blockIdx.x * 256 + threadIdx.x, O = d.O;
    double c = 0.0;
    if (i < O) {
        const int p = d.opose[i], j = d.opoint[i];
        const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
        double pose[6], r[2], z;
#pragma unroll
        for (int k = 0; k < 6; k++) pose[k] = pb.pose[6 * p + k];
        obs_eval(pose, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, &z);
        const bool out = z < depth_eps || (r[0] * r[0] + r[1] * r[1]) > repr_eps;
        d.outl[i] = out ? 1 : 0;
        c = out ? 1.0 : 0.0;
    }
    const double t = block_sum(c, sh);
    if (threadIdx.x == 0) d.part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void k_outliers(BADev d, double repr_eps, double depth_eps) { outliers_body(d, repr_eps, depth_eps); }
__device__ __forceinline__ void outlier_count_body(const BADev &d, int nb_obs)
{
    __shared__ double sh[4];
    const double t = ctl_sum(d.part, nb_obs, 1, sh);           // counts: exact in any order
    if (threadIdx.x == 0) d.st->n_outliers = (int)t;
}
__global__ __launch_bounds__(256) void k_outlier_count(BADev d, int nb_obs) { outlier_count_body(d, nb_obs); }

// ---- the same kernels for a BATCH of windows (slam_local_ba_batch): blockIdx.y = window.  A window's BADev / BandArgs sit in a device
// table that is read through the constant address space with a wave-uniform index -- scalar loads into SGPRs, exactly where a by-value
// kernel argument lives (a generic pointer into the table moved the plane pointers into VGPRs and reloaded them after every store) --
// and the grid's x extent is the largest window's: workgroups beyond a window's own count leave at once.  Every window runs its own
// device-side LM state; a converged window's workgroups early-out as in the single-window path.
struct BAWin { BADev d; BandArgs B; int nb_obs, nb_pts, n_red, pad; int ksplit, pad2; double *bwx; };   // ksplit / bwx: k_ba_window on TWO workgroups -- the first map point (sorted order) of the second one, their exchange area      // pad = 1: the window runs in k_ba_window (one workgroup, all iterations)
static_assert(sizeof(BAWin) % 8 == 0, "BAWin is copied as 64-bit words");
__device__ __forceinline__ BAWin ba_win(const BAWin *tab)
{
    typedef const __attribute__((address_space(4))) unsigned long long *cq_t;
    cq_t q = (cq_t)(const void *)(tab + blockIdx.y);
    unsigned long long raw[sizeof(BAWin) / 8];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(BAWin) / 8); k++) raw[k] = q[k];
    BAWin w;
    __builtin_memcpy(&w, raw, sizeof w);
    return w;
}
__global__ __launch_bounds__(256) void k_linearize_b(const BAWin *tab, int ignore_outliers, int respect_done)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    if ((int)blockIdx.x >= w.nb_obs) return;
    linearize_body<false>(w.d, ignore_outliers, respect_done);
}
// start of a pass: ssr of the current residuals (k_control mode 0) + the reset of the LM state (k_lm_reset), one launch
__global__ __launch_bounds__(256) void k_pass_start_b(const BAWin *tab, int pass)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    control_body(w.d, 0, w.nb_obs, w.nb_pts, 0, nullptr);
    {   const ParamBufs pb = param_bufs(w.d);                 // the committed poses' sin / cos for the pass's first build (later ones: the accepted trial's, k_trial_poses_b)
        for (int q = threadIdx.x; q < w.d.P; q += 256) pose_sincos(pb.pose + 6 * q, pb.sc + 6 * q); }
    __syncthreads();
    if (threadIdx.x != 0) return;
    LMState *s = w.d.st;
    if (pass == 0) { s->ssr_init = s->ssr; s->chol_fail = 0; s->n_outliers = 0; }
    s->delta = LM_DELTA0; s->decrease_factor = 2.0; s->converged = 0; s->accept = 0; s->iters = 0;
}
template <int TT> __global__ __launch_bounds__(TT) __attribute__((amdgpu_waves_per_eu(4))) void k_schur_groups_b(const BAWin *tab, int ignore_outliers)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    if ((int)blockIdx.x >= w.d.ngrp) return;
    schur_groups_body<TT>(w.d, 0.0, ignore_outliers, 1);
}
template <int TT> __global__ __launch_bounds__(TT) void k_schur_groups_m(const BAWin *tab, int ignore_outliers)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    if ((int)blockIdx.x >= w.d.ngrp) return;
    schur_groups_mfma_body<TT>(w.d, ignore_outliers);
}
__global__ __launch_bounds__(256) void k_schur_reduce_b(const BAWin *tab)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    if ((int)blockIdx.x >= w.n_red) return;
    schur_reduce_body(w.d, 1);
}
__global__ __launch_bounds__(BS_T) void k_band_solve_b(const BAWin *tab)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    band_solve_body(w.d, w.B, 1);
}
// the trial poses' sin / cos, once per window and iteration (one wave; behind k_band_solve_b, ahead of k_update_groups_b)
__global__ __launch_bounds__(64) void k_trial_poses_b(const BAWin *tab)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    const BADev &d = w.d;
    if (d.st->converged) return;
    const ParamBufs pb = param_bufs(d);
    for (int q = threadIdx.x; q < d.P; q += 64) {
        const double tp[3] = {pb.pose[6 * q] - d.dp[6 * q], pb.pose[6 * q + 1] - d.dp[6 * q + 1], pb.pose[6 * q + 2] - d.dp[6 * q + 2]};
        pose_sincos(tp, pb.sc_t + 6 * q);
    }
}
// dynamic LDS: [n] dp, [n] + [n] sin / cos of the trial / committed poses, [cap_ob x 3], [cap_sb x 6], [8] (ug_lds_bytes)
static size_t ug_lds_bytes(int n, int cap_ob, int cap_sb) { return ((size_t)3 * n + (size_t)cap_ob * 3 + (size_t)cap_sb * 6 + 8) * 8; }
template <int TT, bool RECOMP> __global__ __launch_bounds__(TT) void k_update_groups_b(const BAWin *tab, int ignore_outliers, int n_cap, int cap_ob, int cap_sb)
{
    extern __shared__ __attribute__((aligned(16))) double ug_lds[];
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    if ((int)blockIdx.x >= w.d.ngrp) return;
    double *s_dp = ug_lds, *s_sct = s_dp + n_cap, *s_sc = s_sct + n_cap, *s_u = s_sc + n_cap, *s_dl = s_u + (size_t)cap_ob * 3, *s_red = s_dl + (size_t)cap_sb * 6;
    update_groups_body<TT, RECOMP>(w.d, ignore_outliers, 1, s_dp, s_u, s_dl, s_red, s_sct, true, s_sc);
}
__global__ __launch_bounds__(256) void k_control_b(const BAWin *tab)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    control_body(w.d, 1, w.d.ngrp, w.d.ngrp, 1 | 2, nullptr);
}
// end of pass 1: record it, flag the outliers at theta_1 (bundle_adjustment.jl:45); the count follows in k_outlier_count_b
__global__ __launch_bounds__(256) void k_outliers_b(const BAWin *tab, double repr_eps, double depth_eps)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    if ((int)blockIdx.x >= w.nb_obs) return;
    outliers_body(w.d, repr_eps, depth_eps);
}
__global__ __launch_bounds__(256) void k_outlier_count_b(const BAWin *tab)
{
    const BAWin w = ba_win(tab);
    if (w.pad) return;                                       // the window is k_ba_window's
    LMState *s = w.d.st;
    if (threadIdx.x == 0) { s->ssr_pass1 = s->ssr; s->iters_pass1 = s->iters; }
    outlier_count_body(w.d, w.nb_obs);
}
// end of pass 2: record it and pack every window's result -- committed parameters (solver's pose order), LM state, outlier flags (sorted
// observation order) -- into one contiguous block for a single device -> host copy.  res: per window [LMState | theta 6P + 3M | outl O]
struct BARes { size_t off_state, off_theta, off_outl; };
__global__ __launch_bounds__(256) void k_results_b(const BAWin *tab, const BARes *rtab, char *res)
{
    const BAWin w = ba_win(tab);
    const BADev &d = w.d;
    LMState *s = d.st;
    const BARes r = rtab[blockIdx.y];
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    const int cur = s->cur;
    const double *pose = cur ? d.pose_t : d.pose, *pts = cur ? d.pts_t : d.pts;
    double *th = (double *)(res + r.off_theta);
    for (int i = tid; i < d.n; i += nth) th[i] = pose[i];
    for (int i = tid; i < 3 * d.M; i += nth) th[d.n + i] = pts[i];
    uint8_t *ol = (uint8_t *)(res + r.off_outl);
    for (int i = tid; i < d.O; i += nth) ol[i] = d.outl[i];
    if (tid == 0) {
        LMState h = *s;
        h.ssr_final = h.ssr; h.iters_pass2 = h.iters;
        *(LMState *)(res + r.off_state) = h;
    }
}


// ---- small windows: the WHOLE two-pass Levenberg-Marquardt of one window in ONE workgroup, one launch for the batch -------------------
// The reference's window is at most 5 free key-frames and their constant observers (estimator.jl:327-331): the reduced camera system is
// 30 x 30, a few hundred map points see a free pose at all, and the rest only move themselves.  Spread over the chip kernel by kernel
// (above) such a window costs 5 launches per iteration whose workgroups are mostly latency; here a 512-thread workgroup (or two: below) keeps the
// window to itself for all 5 + 10 iterations -- no launch boundaries, the LM state never leaves the compute unit:
//   A1 thread = observation (coalesced loads, pose data from LDS): residual + Jacobians, stored for the later phases
//   A2 thread = map point: V = sum Jl'Jl + D, V^-1, bl over its (contiguous) observations -- loads only, no evaluation
//   B  the observations of free poses (a host-built list), in chunks that fit LDS: thread = record -> W = Jp'Jl, gradient term;
//      then lane = block pair (a, b) of the 6 x 6 blocks, 32 subsets of 32 lanes walk the chunk's points; fixed-order fold
//      (deterministic, no atomics): S, g, diag U
//   S  dense damped Cholesky of the <= 30 x 30 system by ONE wave (wave-synchronous LDS, no workgroup barriers), L y = g, L' dp = y
//   C1 thread = map point: dl = V^-1 (bl - W' dp), trial point;  C2 thread = observation: trial and predicted residuals
//   D  LeastSquaresOptim's accept / reject (lm_decide), on the device as everywhere
// then the outlier flags between the passes.  128 such windows occupy 128 compute units at once (256 on two workgroups each).  Windows with more free poses, free
// poses that are not consecutive, > 128 poses or > BW_OMAX observations take the batch kernels above.
// (First version, thread = map point with a serial loop over its observations at 512 threads: 235 us per iteration -- two waves per
//  SIMD cannot hide the dependent loads and the Float64 latency of ten evaluations in a row; slower than the kernels it replaces.)
#ifdef BW_TRACE
#define BW_CLK(k) do { if (tid == 0) bw_clk[k] = clock64(); } while (0)
#else
#define BW_CLK(k)
#endif
#define BW_T 512
#define BW_FMAX 5
#define BW_PMAX 128
#define BW_OMAX 40000
#define BW_HC 352                   // free-pose observation records (W 18, Jp 12, gradient 6 doubles) per chunk (the reference's window: ~690 records = two chunks)
#define BW_PC 256                   // points per chunk
#define BW_WOB 168                  // phase A: observations per wave and trip (BW_WOB x 9 doubles x 8 waves = the record region)
#define BWX_DOUBLES (8 + 2 * 2 * 832)  // exchange area of a window on two workgroups: flags, then [half][buffer][32 lanes x 25 + scalars]
#define BW_FIXED_DBL(P) ((size_t)15 * (P) + 31 * 30 + 32 + 32 + 32 + 2 + (size_t)BW_PC * 10)
static size_t bw_lds_bytes(int P)
{
    size_t b = BW_FIXED_DBL(P) * 8 + (size_t)BW_PC * BW_FMAX * 2 + (size_t)P;
    b = (b + 15) & ~(size_t)15;
    return b + (size_t)BW_HC * 36 * 8 + 16;
}
__device__ __forceinline__ double bw_sum(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < BW_T / 64; w++) t += sh[w];
    return t;
}
__device__ __forceinline__ double bw_max(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < BW_T / 64; w++) t = fmax(t, sh[w]);
    return t;
}
__device__ __forceinline__ void bw_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// TWO WORKGROUPS PER WINDOW (two != 0; 128 windows then use all 256 compute units): workgroups b and b + 8 (same XCD) share window
// (b & 7) + 8 (b >> 4); half h takes the map points [0, ksplit) / [ksplit, M) and their observations through every phase, and the two
// exchange (a) the folded partials of the reduced system once per iteration -- both then assemble and solve it, identically --, (b) the
// three sums behind the step decision, (c) the cost at the start of a pass and the outlier count.  The LM state is a copy in LDS that
// both advance identically (half 0 writes it back at the end).  Exchanges go through a double-buffered area in global memory: values and
// a counting flag as agent-scope atomics (performed at the memory side), ordered by s_waitcnt vmcnt(0) -- no cache write-back or invalidate.
__global__ __launch_bounds__(BW_T) void k_ba_window(const BAWin *tab, const int *list, int ns, int two, int iters_fast, int iterations, double repr_eps, double depth_eps, long long xlimit)
{
    const int half = two ? (int)((blockIdx.x >> 3) & 1) : 0;
    const int widx = two ? (int)((blockIdx.x & 7) + 8 * (blockIdx.x >> 4)) : (int)blockIdx.x;
    if (widx >= ns) return;
    BAWin w;
    {   typedef const __attribute__((address_space(4))) unsigned long long *cq_t;
        cq_t q = (cq_t)(const void *)(tab + list[widx]);
        unsigned long long raw[sizeof(BAWin) / 8];
#pragma unroll
        for (int k = 0; k < (int)(sizeof(BAWin) / 8); k++) raw[k] = q[k];
        __builtin_memcpy(&w, raw, sizeof w); }
    const BADev &d = w.d;
    extern __shared__ __attribute__((aligned(16))) double bw_sm[];
    const int tid = threadIdx.x;
    const int P = d.P, M = d.M, O = d.O, p0 = w.B.p0, F = w.B.nb, n = 6 * F, nwin = F * (F + 1) / 2, NF = d.pfs[M];
    double *s_sc = bw_sm;                              // [P][6] sin / cos of the committed poses' angles
    double *s_sct = s_sc + 6 * P;                      // [P][6] of the trial poses
    double *s_tr = s_sct + 6 * P;                      // [P][3] committed translations
    double *s_A = s_tr + 3 * P;                        // [n + 1][n]: damped S (full, row-major), row n = right-hand side
    double *s_dp = s_A + 31 * 30;                      // [32]
    double *s_ud = s_dp + 32;                          // [32] diag U (damping)
    double *s_red = s_ud + 32;                         // [32]
    int *s_flag = (int *)(s_red + 32);                 // [4]
    double *s_pt = s_red + 34;                         // [BW_PC][10] V^-1 (6), bl (3)
    short *s_slot = (short *)(s_pt + BW_PC * 10);      // [BW_PC][BW_FMAX] record of point x for free pose a, or -1
    unsigned char *s_const = (unsigned char *)(s_slot + BW_PC * BW_FMAX);   // [P]
    double *s_W = bw_sm + (((BW_FIXED_DBL(P) * 8 + (size_t)BW_PC * BW_FMAX * 2 + (size_t)P + 15) & ~(size_t)15) >> 3);   // [BW_HC][18]
    double *s_Jp = s_W + BW_HC * 18;                   // [BW_HC][12]
    double *s_gr = s_Jp + BW_HC * 12;                  // [BW_HC][6]
    double *s_fold = s_W;                              // after the last chunk: [15][32 * 18 + n * 7] partials of the waves 1 .. 15
    // phase B: 32 subsets of 32 lanes; lane = (block pair (a <= b), upper / lower three rows of its 6 x 6 block) and / or (slot a2, row r2):
    // half a block per lane keeps the accumulators + one W record under the 128 registers of a 1024-thread workgroup
    const int sub = tid >> 5, wl = tid & 31, w2 = wl >> 1, rh = 3 * (wl & 1);
    int ba_ = 0, bb_ = 0;
    { int r = w2; while (ba_ < F && r >= F - ba_) { r -= F - ba_; ba_++; } bb_ = ba_ + r; }
    const bool live = w2 < nwin, xl = wl < n;
    const int a2 = xl ? wl / 6 : 0, r2 = wl - 6 * (wl / 6);
    __shared__ LMState s_lm;                           // the LM state lives HERE for the whole solve (both halves of a split window advance their copies identically)
    LMState *s = &s_lm;
    if (tid == 0) s_lm = *d.st;
    // this workgroup's map points [kLo, kHi) (sorted order) and observations [oLo, oHi)
    const int kLo = two && half ? w.ksplit : 0, kHi = two && !half ? w.ksplit : M;
    const int oLo = d.pt_start[kLo], oHi = d.pt_start[kHi];
    // exchange with the other half: own values -> area [half][e & 1], flag[half] = e; wait for flag[1 - half] >= e; the sums are own + other
    int xe = 0, xdead = 0;             // xdead: this half gave up waiting (lane 0 of wave 0 keeps it)
    int *xflag = (int *)w.bwx;
    auto xarea = [&](int h, int e) { return w.bwx + 8 + (size_t)(2 * h + (e & 1)) * 832; };
    auto xpost = [&](int e) {          // (called by the wave that wrote the values)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((tid & 63) == 0 && !xdead) __hip_atomic_store(xflag + half, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // The wait is BOUNDED (xlimit ticks of the 100 MHz wall clock): the launch is not cooperative, so nothing but the host's count of
    // compute units promises that the partner workgroup is resident.  A half that runs out of patience marks the window (xflag[2]), posts a
    // flag no later wait can miss (the partner never waits for it again) and goes on WITHOUT waiting -- it only ever exchanges doubles, every
    // loop bound is an iteration count, so the garbage it then computes ends by itself -- and half 0 reports chol_fail = 2: the host solves
    // the call again on one workgroup per window.  A missing partner is an error code, never a hung queue.
    auto xwait = [&](int e) {
        if ((tid & 63) == 0 && !xdead) {
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(xflag + (1 - half), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
                if ((long long)wall_clock64() - t0 > xlimit) {
                    xdead = 1;
                    __hip_atomic_store(xflag + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(xflag + half, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
    };
    // three scalars (two sums, one maximum) across the halves; every thread holds the workgroup's values on entry and the window's on return
    auto xscal = [&](double &a, double &b, double &c) {
        if (!two) return;
        const int e = ++xe;
        if (tid == 0) {
            double *o = xarea(half, e) + 800;
            __hip_atomic_store(o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(o + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(o + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 64) {
            xpost(e); xwait(e);
            if (tid == 0) {
                const double *q = xarea(1 - half, e) + 800;
                const double a1 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c1 = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_red[8] = half ? a1 + a : a + a1; s_red[9] = half ? b1 + b : b + b1; s_red[10] = fmax(c, c1);      // (half 0's value first on both sides: the same bits)
            }
        }
        __syncthreads();
        a = s_red[8]; b = s_red[9]; c = s_red[10];
        __syncthreads();
    };
    for (int p = tid; p < P; p += BW_T) s_const[p] = d.pconst[p];
    // phase A's chunks: npc consecutive map points (sorted order) per wave and trip, at most BW_WOB observations (d.sg_ob = most observations of one point)
    const int npc_a = max(1, min(64, BW_WOB / d.sg_ob));
#ifdef BW_TRACE
    long long bw_clk[12];
#endif
    // committed pose data -> LDS (sin / cos of the angles, translation)
    auto stage_poses = [&](const ParamBufs &pb) {
        for (int p = tid; p < P; p += BW_T) {
            pose_sincos(pb.pose + 6 * p, s_sc + 6 * p);
            s_tr[3 * p] = pb.pose[6 * p + 3]; s_tr[3 * p + 1] = pb.pose[6 * p + 4]; s_tr[3 * p + 2] = pb.pose[6 * p + 5];
        }
    };

    auto pbufs = [&]() { const bool sw = s->cur != 0; return ParamBufs{sw ? d.pose_t : d.pose, sw ? d.pts_t : d.pts, sw ? d.pose : d.pose_t, sw ? d.pts : d.pts_t, nullptr, nullptr}; };
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {
        const int ignore = pass, iters = pass ? iterations : iters_fast;
        // ---- cost at the committed parameters (LeastSquaresOptim evaluates f!(fcur, x) first)
        {
            const ParamBufs pb = pbufs();
            __syncthreads();
            stage_poses(pb);
            __syncthreads();
            double ss = 0.0;
            for (int i = oLo + tid; i < oHi; i += BW_T) {
                if (ignore && d.outl[i]) continue;
                const int p = d.opose[i], j = d.opoint[i];
                const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                double r[2];
                obs_eval_sc(s_sc + 6 * p, s_tr + 3 * p, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
                ss += r[0] * r[0] + r[1] * r[1];
            }
            double t = bw_sum(ss, s_red), tz1 = 0.0, tz2 = 0.0;
            xscal(t, tz1, tz2);
            if (tid == 0) {
                s->ssr = t;
                if (pass == 0) { s->ssr_init = t; s->chol_fail = 0; s->n_outliers = 0; }
                s->delta = LM_DELTA0; s->decrease_factor = 2.0; s->converged = 0; s->accept = 0; s->iters = 0;
            }
            __syncthreads();
        }
        for (int it = 1; it <= iters; it++) {
            if (s->converged) break;                           // (uniform: every thread reads the flag after a barrier)
            const ParamBufs pb = pbufs();
            const double inv_delta = 1.0 / s->delta;
            BW_CLK(0);
            stage_poses(pb);
            __syncthreads();
            BW_CLK(1);
            // ---- A: every WAVE takes chunks of npc_a consecutive map points (their observations are contiguous: coalesced loads, lane = observation):
            //      residual + Jacobians (stored for the later phases), the nine products Jl'Jl / Jl'f into the wave's own LDS block; then lane = map
            //      point of the chunk: V = sum + D, V^-1, bl in the observations' order.  Wave-synchronous -- no workgroup barrier inside the phase,
            //      the eight waves overlap each other's latencies.  (Versions before: thread = point summing from the stored records -- 64 cache
            //      lines per load instruction, 127 k cycles; workgroup-wide tiles with two barriers each -- 171 k.)  V^-1 / bl / dl: SORTED point order.
            {
                const int wvA = tid >> 6, ln = tid & 63;
                double *s_w9 = s_W + (size_t)wvA * BW_WOB * 9;
                for (int k0 = kLo + wvA * npc_a; k0 < kHi; k0 += (BW_T / 64) * npc_a) {
                    const int k1 = min(kHi, k0 + npc_a), o0 = d.pt_start[k0], nobs = d.pt_start[k1] - o0;
                    for (int t = ln; t < nobs; t += 64) {
                        const int i = o0 + t;
                        const int p = d.opose[i], j = d.opoint[i];
                        const bool active = !(ignore && d.outl[i]);
                        const bool hp = active && !s_const[p];
                        double r[2] = {0.0, 0.0}, Jp[12], Jl[6] = {0, 0, 0, 0, 0, 0};
                        if (active) {
                            const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                            obs_eval_sc(s_sc + 6 * p, s_tr + 3 * p, X, d.pix[i], d.pix[O + i], d.cam, r, hp ? Jp : nullptr, Jl, nullptr);
                        }
                        d.hasp[i] = hp ? 1 : 0;
                        st_rec<2>(d.f + 2 * (size_t)i, r);
                        st_rec<6>(d.Jl + (size_t)i * 6, Jl);
                        if (hp) st_rec<12>(d.Jp + (size_t)i * 12, Jp);
                        double *v = s_w9 + t * 9;
                        v[0] = Jl[0] * Jl[0] + Jl[3] * Jl[3]; v[1] = Jl[0] * Jl[1] + Jl[3] * Jl[4]; v[2] = Jl[0] * Jl[2] + Jl[3] * Jl[5];
                        v[3] = Jl[1] * Jl[1] + Jl[4] * Jl[4]; v[4] = Jl[1] * Jl[2] + Jl[4] * Jl[5]; v[5] = Jl[2] * Jl[2] + Jl[5] * Jl[5];
#pragma unroll
                        for (int c = 0; c < 3; c++) v[6 + c] = Jl[c] * r[0] + Jl[3 + c] * r[1];
                    }
                    bw_wave_sync();
                    if (ln < k1 - k0) {
                        const int k = k0 + ln;
                        double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                        const int t0 = d.pt_start[k] - o0, t1 = d.pt_start[k + 1] - o0;
                        for (int t = t0; t < t1; t++) {
#pragma unroll
                            for (int c = 0; c < 9; c++) V[c] += s_w9[t * 9 + c];
                        }
                        V[0] += fmin(fmax(V[0], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                        V[3] += fmin(fmax(V[3], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                        V[5] += fmin(fmax(V[5], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                        double Vi[6];
                        inv3_sym(V, Vi);
#pragma unroll
                        for (int c = 0; c < 6; c++) d.Vinv[(size_t)c * M + k] = Vi[c];
#pragma unroll
                        for (int c = 0; c < 3; c++) d.bl[(size_t)c * M + k] = V[6 + c];
                    }
                    bw_wave_sync();
                }
            }
            __syncthreads();
            BW_CLK(2);
            // ---- B: the reduced camera system from the observations of free poses (records fobs[0 .. NF)), chunk by chunk
            double acc[18], ex[7];
#pragma unroll
            for (int k = 0; k < 18; k++) acc[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 7; k++) ex[k] = 0.0;
            for (int k0 = kLo; k0 < kHi;) {
                // the chunk [k0, k1): <= BW_PC points, <= BW_HC records (pfs = running count of free-pose observations by sorted point)
                const int base = d.pfs[k0];
                int lo = k0 + 1, hi = min(kHi, k0 + BW_PC);
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (d.pfs[mid] - base <= BW_HC) lo = mid; else hi = mid - 1; }
                const int k1 = lo, npc = k1 - k0, nrec = d.pfs[k1] - base;
                if (nrec == 0) { k0 = k1; continue; }                    // no point of the chunk sees a free pose
                for (int x = tid; x < npc; x += BW_T) {
                    const int kk = k0 + x;
#pragma unroll
                    for (int c = 0; c < 6; c++) s_pt[x * 10 + c] = d.Vinv[(size_t)c * M + kk];
#pragma unroll
                    for (int c = 0; c < 3; c++) s_pt[x * 10 + 6 + c] = d.bl[(size_t)c * M + kk];
#pragma unroll
                    for (int a = 0; a < BW_FMAX; a++) s_slot[x * BW_FMAX + a] = -1;
                }
                __syncthreads();
                for (int rec = tid; rec < nrec; rec += BW_T) {           // thread = record
                    const int i = d.fobs[base + rec];
                    if (!d.hasp[i]) continue;                            // an ignored outlier: no slot
                    const int x = d.opk[i] - k0, p = d.opose[i];
                    double jp[12], jl[6], ff[2], Vi[6], bl[3];
                    ld_rec<12>(d.Jp + (size_t)i * 12, jp); ld_rec<6>(d.Jl + (size_t)i * 6, jl); ld_rec<2>(d.f + 2 * (size_t)i, ff);
#pragma unroll
                    for (int c = 0; c < 6; c++) Vi[c] = s_pt[x * 10 + c];
#pragma unroll
                    for (int c = 0; c < 3; c++) bl[c] = s_pt[x * 10 + 6 + c];
                    const double vb0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
                    const double vb1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
                    const double vb2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        const double w0 = jp[a] * jl[0] + jp[6 + a] * jl[3];
                        const double w1 = jp[a] * jl[1] + jp[6 + a] * jl[4];
                        const double w2 = jp[a] * jl[2] + jp[6 + a] * jl[5];
                        s_W[rec * 18 + 3 * a] = w0; s_W[rec * 18 + 3 * a + 1] = w1; s_W[rec * 18 + 3 * a + 2] = w2;
                        s_gr[rec * 6 + a] = (jp[a] * ff[0] + jp[6 + a] * ff[1]) - (w0 * vb0 + w1 * vb1 + w2 * vb2);
                    }
#pragma unroll
                    for (int c = 0; c < 12; c++) s_Jp[rec * 12 + c] = jp[c];
                    s_slot[x * BW_FMAX + (p - p0)] = (short)rec;
                }
                __syncthreads();
                if (live)
                    for (int x = sub; x < npc; x += BW_T / 32) {
                        const int ta = s_slot[x * BW_FMAX + ba_], tb = s_slot[x * BW_FMAX + bb_];
                        if (ta < 0 || tb < 0) continue;
                        double Vi[6], Wb[18];
                        ld_rec<6>(s_pt + x * 10, Vi); ld_rec<18>(s_W + tb * 18, Wb);
#pragma unroll
                        for (int rr = 0; rr < 3; rr++) {
                            const double *wa = s_W + ta * 18 + 3 * (rh + rr);
                            const double a0 = wa[0], a1 = wa[1], a2v = wa[2];
                            const double T0 = fma(a2v, Vi[2], fma(a1, Vi[1], a0 * Vi[0]));
                            const double T1 = fma(a2v, Vi[4], fma(a1, Vi[3], a0 * Vi[1]));
                            const double T2 = fma(a2v, Vi[5], fma(a1, Vi[4], a0 * Vi[2]));
#pragma unroll
                            for (int c = 0; c < 6; c++)
                                acc[6 * rr + c] = fma(-T2, Wb[3 * c + 2], fma(-T1, Wb[3 * c + 1], fma(-T0, Wb[3 * c], acc[6 * rr + c])));
                        }
                    }
                if (xl)
                    for (int x = sub; x < npc; x += BW_T / 32) {
                        const int ta = s_slot[x * BW_FMAX + a2];
                        if (ta < 0) continue;
                        double J[12];
                        ld_rec<12>(s_Jp + ta * 12, J);
                        const double j0 = r2 == 0 ? J[0] : r2 == 1 ? J[1] : r2 == 2 ? J[2] : r2 == 3 ? J[3] : r2 == 4 ? J[4] : J[5];
                        const double j1 = r2 == 0 ? J[6] : r2 == 1 ? J[7] : r2 == 2 ? J[8] : r2 == 3 ? J[9] : r2 == 4 ? J[10] : J[11];
#pragma unroll
                        for (int c = 0; c < 6; c++) ex[c] = fma(j1, J[6 + c], fma(j0, J[c], ex[c]));
                        ex[6] += s_gr[ta * 6 + r2];
                    }
                __syncthreads();
                k0 = k1;
            }
            BW_CLK(3);
            // fold the 32 subsets in a fixed order: the two subsets of a wave by a lane exchange, the 16 waves through LDS by wave 0
#pragma unroll
            for (int k = 0; k < 18; k++) acc[k] += __shfl_xor(acc[k], 32);
#pragma unroll
            for (int k = 0; k < 7; k++) ex[k] += __shfl_xor(ex[k], 32);
            const int fstride = 32 * 18 + n * 7, wv = tid >> 6;
            if (wv >= 1 && (tid & 63) < 32) {
                if (live) st_rec<18>(s_fold + (size_t)(wv - 1) * fstride + wl * 18, acc);
                if (xl) {
#pragma unroll
                    for (int k = 0; k < 7; k++) s_fold[(size_t)(wv - 1) * fstride + 32 * 18 + wl * 7 + k] = ex[k];
                }
            }
            __syncthreads();
            if (tid < 32) {
                if (live)
                    for (int q = 0; q < BW_T / 64 - 1; q++) {
                        double o[18];
                        ld_rec<18>(s_fold + (size_t)q * fstride + wl * 18, o);
#pragma unroll
                        for (int k = 0; k < 18; k++) acc[k] += o[k];
                    }
                if (xl)
                    for (int q = 0; q < BW_T / 64 - 1; q++) {
#pragma unroll
                        for (int k = 0; k < 7; k++) ex[k] += s_fold[(size_t)q * fstride + 32 * 18 + wl * 7 + k];
                    }
            }
            if (two) {                                         // ... and the other half's: own + other, lane by lane (wave 0; every lane of the 32 writes all its 25 values)
                const int e = ++xe;
                if (tid < 64) {
                    if (tid < 32) {
                        double *o = xarea(half, e) + wl * 25;
#pragma unroll
                        for (int k = 0; k < 18; k++) __hip_atomic_store(o + k, acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int k = 0; k < 7; k++) __hip_atomic_store(o + 18 + k, ex[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    xpost(e); xwait(e);
                    if (tid < 32) {
                        const double *q = xarea(1 - half, e) + wl * 25;
#pragma unroll
                        for (int k = 0; k < 18; k++) { const double v = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc[k] = half ? v + acc[k] : acc[k] + v; }
#pragma unroll
                        for (int k = 0; k < 7; k++) { const double v = __hip_atomic_load(q + 18 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ex[k] = half ? v + ex[k] : ex[k] + v; }
                    }
                }
            }
            __syncthreads();                                   // the fold buffer is read: the system goes where no partial lives (s_A)
            // ---- S: wave 0 alone assembles and solves (wave-synchronous LDS: a wave's DS instructions execute in order)
            if (tid < 64) {
                if (tid < 32 && xl) {
                    const double ud = r2 == 0 ? ex[0] : r2 == 1 ? ex[1] : r2 == 2 ? ex[2] : r2 == 3 ? ex[3] : r2 == 4 ? ex[4] : ex[5];
                    s_A[n * n + wl] = ex[6];                   // right-hand side row
                    s_ud[wl] = ud;
#pragma unroll
                    for (int c = 0; c < 6; c++) s_fold[a2 * 36 + r2 * 6 + c] = ex[c];      // Jp'Jp row r2 of slot a2 -> the diagonal block
                }
                bw_wave_sync();
                if (tid < 32 && live) {
                    if (ba_ == bb_) {
#pragma unroll
                        for (int k = 0; k < 18; k++) acc[k] += s_fold[ba_ * 36 + 6 * rh + k];
                    }
#pragma unroll
                    for (int rr = 0; rr < 3; rr++)
#pragma unroll
                        for (int c = 0; c < 6; c++) {
                            s_A[(6 * ba_ + rh + rr) * n + 6 * bb_ + c] = acc[6 * rr + c];
                            if (ba_ != bb_) s_A[(6 * bb_ + c) * n + 6 * ba_ + rh + rr] = acc[6 * rr + c];
                        }
                }
                bw_wave_sync();
                BW_CLK(4);
                // damped Cholesky A = L L': lane i keeps row i of the lower triangle in REGISTERS (lane n: the right-hand side row -- the forward
                // substitution comes for free); the entries of row jc a step needs are lane broadcasts (v_readlane), not LDS round trips
                // (a first version walked the rows in LDS: 79 k cycles per solve, every multiply-add behind an exposed LDS latency)
                if (tid < n) s_A[tid * n + tid] += fmin(fmax(s_ud[tid], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                bw_wave_sync();
                double a[6 * BW_FMAX];
#pragma unroll
                for (int k = 0; k < 6 * BW_FMAX; k++) a[k] = (tid <= n && k < n) ? s_A[tid * n + k] : 0.0;
                bool bad = false;
#pragma unroll
                for (int jc = 0; jc < 6 * BW_FMAX; jc++) {
                    if (jc < n) {
                        // (four partial sums: a lone wave pays 36 cycles for a DEPENDENT Float64 operation, 9.5 for an independent one; 1 / sqrt from
                        //  v_rsq_f64 + one Newton step, 4e-15 relative, as in k_band_solve -- the solve: 31 k -> 20 k cycles)
                        double sq[4] = {a[jc], 0.0, 0.0, 0.0};
#pragma unroll
                        for (int k = 0; k < jc; k++)
                            sq[k & 3] -= a[k] * __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[k]), jc), __builtin_amdgcn_readlane(__double2loint(a[k]), jc));
                        const double sum = (sq[0] + sq[1]) + (sq[2] + sq[3]);
                        const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sum), jc), __builtin_amdgcn_readlane(__double2loint(sum), jc));
                        const bool okp = piv > 0.0 && piv < 1e300;
                        if (!okp) bad = true;
                        const double pv = okp ? piv : 1.0;
                        const double y0 = __builtin_amdgcn_rsq(pv);
                        const double r0 = __builtin_fma(-(pv * y0), y0, 1.0), rd = __builtin_fma(y0 * 0.5, r0, y0);
                        a[jc] = tid == jc ? pv * rd : sum * rd;
                    }
                }
                // L (rows 0 .. n - 1) and y' = (L^-1 g)' (row n) back to LDS; then lane i takes COLUMN i of L and the back-substitution
                // L' dp = y runs in registers too
                if (tid <= n) {
#pragma unroll
                    for (int k = 0; k < 6 * BW_FMAX; k++) if (k < n) s_A[tid * n + k] = a[k];
                }
                bw_wave_sync();
                double y = tid < n ? s_A[n * n + tid] : 0.0;
                const double dg = 1.0 / (tid < n ? s_A[tid * n + tid] : 1.0);      // (one division per lane, all at once; the chain multiplies)
#pragma unroll
                for (int k = 0; k < 6 * BW_FMAX; k++) a[k] = (tid < n && k < n && k > tid) ? s_A[k * n + tid] : 0.0;      // a[k] = L[k][tid]
#pragma unroll
                for (int jc = 6 * BW_FMAX - 1; jc >= 0; jc--) {
                    if (jc < n) {
                        const double yj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), jc), __builtin_amdgcn_readlane(__double2loint(y), jc));
                        const double dj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dg), jc), __builtin_amdgcn_readlane(__double2loint(dg), jc));
                        const double xj = yj * dj;
                        if (tid == jc) y = xj;
                        if (tid < jc) y -= a[jc] * xj;
                    }
                }
                if (tid < n) s_dp[tid] = y;
                if (tid == 0) s_flag[0] = bad ? 1 : 0;
            }
            __syncthreads();
            if (s_flag[0]) { if (tid == 0) s->chol_fail = 1; }
            BW_CLK(5);
            // ---- C1: trial poses; thread = map point: dl, trial point
            double mx = 0.0;
            if (tid < n) { const int p = p0 + tid / 6, c = tid - 6 * (tid / 6); const double v = s_dp[tid]; pb.pose_t[6 * p + c] = pb.pose[6 * p + c] - v; mx = fabs(v); }
            for (int p = tid; p < P; p += BW_T) {
                if (s_const[p]) {
#pragma unroll
                    for (int c = 0; c < 6; c++) { pb.pose_t[6 * p + c] = pb.pose[6 * p + c]; s_sct[6 * p + c] = s_sc[6 * p + c]; }
                } else {
                    const int a = p - p0;
                    const double tp[3] = {pb.pose[6 * p] - s_dp[6 * a], pb.pose[6 * p + 1] - s_dp[6 * a + 1], pb.pose[6 * p + 2] - s_dp[6 * a + 2]};
                    pose_sincos(tp, s_sct + 6 * p);
                }
            }
            for (int k = kLo + tid; k < kHi; k += BW_T) {
                const int j = d.pt_id[k];
                double bl[3] = {d.bl[k], d.bl[(size_t)M + k], d.bl[(size_t)2 * M + k]};
                for (int rec = d.pfs[k]; rec < d.pfs[k + 1]; rec++) {      // the point's observations of free poses (host list): bl -= Jl' (Jp dp)
                    const int i = d.fobs[rec];
                    if (!d.hasp[i]) continue;
                    const int a = d.opose[i] - p0;
                    double jp[12], jl[6];
                    ld_rec<12>(d.Jp + (size_t)i * 12, jp); ld_rec<6>(d.Jl + (size_t)i * 6, jl);
                    double ua = 0.0, ub = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; c++) { ua += jp[c] * s_dp[6 * a + c]; ub += jp[6 + c] * s_dp[6 * a + c]; }
#pragma unroll
                    for (int c = 0; c < 3; c++) bl[c] -= jl[c] * ua + jl[3 + c] * ub;
                }
                double Vi[6];
#pragma unroll
                for (int c = 0; c < 6; c++) Vi[c] = d.Vinv[(size_t)c * M + k];
                const double l0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
                const double l1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
                const double l2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
                d.dl[3 * k] = l0; d.dl[3 * k + 1] = l1; d.dl[3 * k + 2] = l2;
                pb.pts_t[3 * j] = pb.pts[3 * j] - l0; pb.pts_t[3 * j + 1] = pb.pts[3 * j + 1] - l1; pb.pts_t[3 * j + 2] = pb.pts[3 * j + 2] - l2;
                mx = fmax(mx, fmax(fabs(l0), fmax(fabs(l1), fabs(l2))));
            }
            __syncthreads();
            // ---- C2: thread = observation: trial and predicted residuals
            double st = 0.0, sp = 0.0;
            for (int i = oLo + tid; i < oHi; i += BW_T) {
                const int p = d.opose[i], j = d.opoint[i];
                const bool active = !(ignore && d.outl[i]);
                double jl[6], ff[2], r[2] = {0.0, 0.0}, pa = 0.0, pbv = 0.0;
                ld_rec<6>(d.Jl + (size_t)i * 6, jl); ld_rec<2>(d.f + 2 * (size_t)i, ff);
                const int ks = d.opk[i];
                const double l0 = d.dl[3 * ks], l1 = d.dl[3 * ks + 1], l2 = d.dl[3 * ks + 2];
                const bool fr = !s_const[p];
                const int a = p - p0;
                if (d.hasp[i]) {
                    double jp[12];
                    ld_rec<12>(d.Jp + (size_t)i * 12, jp);
#pragma unroll
                    for (int c = 0; c < 6; c++) { pa += jp[c] * s_dp[6 * a + c]; pbv += jp[6 + c] * s_dp[6 * a + c]; }
                }
                if (active) {
                    const double Xt[3] = {pb.pts_t[3 * j], pb.pts_t[3 * j + 1], pb.pts_t[3 * j + 2]};
                    const double tr[3] = {s_tr[3 * p] - (fr ? s_dp[6 * a + 3] : 0.0), s_tr[3 * p + 1] - (fr ? s_dp[6 * a + 4] : 0.0), s_tr[3 * p + 2] - (fr ? s_dp[6 * a + 5] : 0.0)};
                    obs_eval_sc(s_sct + 6 * p, tr, Xt, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
                }
                pa += jl[0] * l0 + jl[1] * l1 + jl[2] * l2; pbv += jl[3] * l0 + jl[4] * l1 + jl[5] * l2;
                pa -= ff[0]; pbv -= ff[1];
                st += r[0] * r[0] + r[1] * r[1];
                sp += pa * pa + pbv * pbv;
            }
            BW_CLK(6);
            double tt = bw_sum(st, s_red), tp = bw_sum(sp, s_red), tm = bw_max(mx, s_red);
            xscal(tt, tp, tm);
            // ---- D
            if (tid == 0) { s->trial_ssr = tt; s->pred_ssr = tp; s->maxdx = tm; lm_decide(s, tt, tp, tm); }
            __syncthreads();
#ifdef BW_TRACE
            if (tid == 0 && (blockIdx.x == 5 || blockIdx.x == 13) && pass == 0 && it == 3) { bw_clk[7] = clock64();
                printf("k_ba_window (M %d, O %d, F %d, %d free-pose observations): sincos %lld | A %lld | B chunks %lld | fold %lld | solve %lld | C %lld | reduce + decide %lld cycles\n", M, O, F, NF,
                       bw_clk[1] - bw_clk[0], bw_clk[2] - bw_clk[1], bw_clk[3] - bw_clk[2], bw_clk[4] - bw_clk[3], bw_clk[5] - bw_clk[4], bw_clk[6] - bw_clk[5], bw_clk[7] - bw_clk[6]); }
#endif
        }
        if (tid == 0) { if (pass == 0) { s->ssr_pass1 = s->ssr; s->iters_pass1 = s->iters; } else { s->ssr_final = s->ssr; s->iters_pass2 = s->iters; } }
        if (pass == 0) {
            // ---- _ba_detect_outliers! at theta_1 (bundle_adjustment.jl:90-111)
            __syncthreads();
            const ParamBufs pb = pbufs();
            stage_poses(pb);
            __syncthreads();
            double cnt = 0.0;
            for (int i = oLo + tid; i < oHi; i += BW_T) {
                const int p = d.opose[i], j = d.opoint[i];
                const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                double r[2], z;
                obs_eval_sc(s_sc + 6 * p, s_tr + 3 * p, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, &z);
                const bool out = z < depth_eps || (r[0] * r[0] + r[1] * r[1]) > repr_eps;
                d.outl[i] = out ? 1 : 0;
                cnt += out ? 1.0 : 0.0;
            }
            double tc = bw_sum(cnt, s_red), tz1 = 0.0, tz2 = 0.0;
            xscal(tc, tz1, tz2);
            if (tid == 0) s->n_outliers = (int)tc;
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0 && half == 0) {
        if (two && (xdead || __hip_atomic_load(xflag + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) s_lm.chol_fail = 2;     // a half gave up waiting: nothing of this window is valid
        *d.st = s_lm;
    }
}

// ---------------------------------------------------------------------------------
static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

// dynamic LDS of k_band_solve for Ps block columns of half-bandwidth hb (n = 6 P: the damping / solution vectors span every pose)
static size_t band_lds_bytes(int n, int Ps, int hb)
{
    const size_t band_fixed = (3 * (size_t)n + 36 * (size_t)Ps) * 8;      // x, damp, chat, L_kk^-1 of every column
    size_t band_lds = band_fixed + ((size_t)(hb + 1) * (hb + 1) * BS_WS + (size_t)(hb + 1) * (6 + BS_WS + 6) + 8 + 36 + 56) * 8 + (size_t)BS_PF * BS_PT * 8;
    if (hb * 6 <= 58) {      // narrow bands: room to stage the G blocks of (up to) all back-substitution steps, at least one
        const size_t want = band_fixed + (size_t)Ps * hb * 36 * 8, least = band_fixed + (size_t)hb * 36 * 8;
        band_lds = std::max(band_lds, std::max(std::min(want, (size_t)150 * 1024), least));
    }
    return band_lds;
}

// A pose order in which the reduced camera system is block-banded, for windows that are not in the caller's order: a loop closure
// (the first and the last key-frames of the window share map points, map_manager.jl:300-449) turns the covisibility chain into a ring, and
// a ring of half-bandwidth h is a band of half-bandwidth ~2 h once it is folded (c, c - 1, c + 1, c - 2, ...).  Candidates, judged by the
// half-bandwidth of the FREE poses' covisibility graph (the constant poses go first: they have no block in the system and only widen the
// span between free ones): the caller's order, every fold of it, Cuthill-McKee from a pose of lowest degree and its reverse.  Returns
// false if none fits the banded solver.  Host work: O(observations) + O(poses x edges); only windows headed for the general path get here.
static bool ba_pose_order(int P, int M, int O, const uint8_t *theta_const, const int64_t *pose_ids, const int64_t *point_ids, std::vector<int> &order)
{
    std::vector<int> fidx(P, -1), fr;
    for (int p = 0; p < P; p++) if (!theta_const[p]) { fidx[p] = (int)fr.size(); fr.push_back(p); }
    const int F = (int)fr.size();
    if (F < 3) return false;
    // the free observers of every point (counting sort by point), then the graph: an edge per pair of free poses that share a point
    std::vector<int> st(M + 1, 0);
    for (int i = 0; i < O; i++) if (fidx[pose_ids[i] - 1] >= 0) st[point_ids[i]]++;
    for (int j = 0; j < M; j++) st[j + 1] += st[j];
    std::vector<int> lst(st[M]), fill(st.begin(), st.end() - 1);
    for (int i = 0; i < O; i++) { const int f = fidx[pose_ids[i] - 1]; if (f >= 0) lst[fill[point_ids[i] - 1]++] = f; }
    std::vector<uint8_t> adj((size_t)F * F, 0);
    for (int j = 0; j < M; j++) {
        const int a0 = st[j], a1 = st[j + 1], k = a1 - a0;
        if (j > 0 && st[j] - st[j - 1] == k && std::equal(lst.begin() + a0, lst.begin() + a1, lst.begin() + st[j - 1])) continue;   // same observers as the point before
        for (int a = a0; a < a1; a++)
            for (int b = a0; b < a1; b++) adj[(size_t)lst[a] * F + lst[b]] = 1;
    }
    std::vector<int2> edges; std::vector<int> deg(F, 0);
    for (int a = 0; a < F; a++)
        for (int b = a + 1; b < F; b++) if (adj[(size_t)a * F + b]) { edges.push_back(make_int2(a, b)); deg[a]++; deg[b]++; }
    std::vector<int> pos(F), cand(F), best;
    int best_bw = 1 << 30;
    auto judge = [&]() {                                       // cand[k] = free pose at position k
        for (int k = 0; k < F; k++) pos[cand[k]] = k;
        int bw = 0;
        for (const int2 &e : edges) bw = std::max(bw, std::abs(pos[e.x] - pos[e.y]));
        if (bw < best_bw) { best_bw = bw; best = cand; }
    };
    for (int k = 0; k < F; k++) cand[k] = k;
    judge();
    for (int c = 0; c < F; c++) {                              // folds: c, c - 1, c + 1, c - 2, ... around the ring
        for (int k = 0; k < F; k++) cand[k] = ((k & 1 ? c - (k + 1) / 2 : c + k / 2) % F + F) % F;
        judge();
    }
    {   // Cuthill-McKee (neighbours by ascending degree), every component from its pose of lowest degree; then reversed
        std::vector<uint8_t> seen(F, 0); std::vector<int> q; q.reserve(F);
        while ((int)q.size() < F) {
            int s0 = -1;
            for (int a = 0; a < F; a++) if (!seen[a] && (s0 < 0 || deg[a] < deg[s0])) s0 = a;
            seen[s0] = 1; size_t head = q.size(); q.push_back(s0);
            while (head < q.size()) {
                const int u = q[head++]; const size_t n0 = q.size();
                for (int v = 0; v < F; v++) if (!seen[v] && adj[(size_t)u * F + v]) { seen[v] = 1; q.push_back(v); }
                std::stable_sort(q.begin() + n0, q.end(), [&](int x, int y) { return deg[x] < deg[y]; });
            }
        }
        cand = q; judge();
        std::reverse(cand.begin(), cand.end()); judge();
    }
    const int hbq = std::min(std::max(best_bw, 1), F - 1);
    if (best_bw > BS_MAXHB || !sg_fold_fits(best_bw) || band_lds_bytes(6 * P, F, hbq) > 150 * 1024) return false;
    order.clear();
    for (int p = 0; p < P; p++) if (theta_const[p]) order.push_back(p);
    for (int k = 0; k < F; k++) order.push_back(fr[best[k]]);
    return true;
}

// ---- set-up of one window, in two host-only halves so that a batch of windows can be prepared by several threads:
//   ba_plan   the structure of the problem (map points sorted by first free observer, observation order, point groups or pair lists,
//             pose order) and the layout of its device memory in three regions -- uploaded arrays, zero-initialised state, work arrays;
//   ba_emit   binds the device pointers to the three region bases and writes the uploaded region into a host staging block.
// slam_ba_create / slam_local_ba give one window its own arena (the three regions back to back); slam_local_ba_batch lays the
// regions of all windows out region-major (one H2D copy, one memset for the whole batch).  Neither half makes a HIP call.
struct BAPlan {
    // inputs
    double fx = 0, fy = 0, cx = 0, cy = 0; int P = 0, M = 0, O = 0;
    const double *theta = nullptr; const uint8_t *theta_const_in = nullptr; const double *pixels_yx = nullptr;
    const int64_t *pose_ids = nullptr, *point_ids = nullptr;
    bool may_reorder = false, small_groups = false;
    bool window = false;         // result: the window fits k_ba_window (<= 5 consecutive free poses, ...): no point groups are built for it
    int nfree_obs = 0;           // result: observations of free poses
    // results
    slam_ba *ba = nullptr;
    int err = 0; char msg[160] = {0};
    std::vector<int> cnt, pfirst, new_of, pt_id, rank, start, fgrp;
    std::vector<uint8_t> const_perm;
    std::vector<int4> grp;
    std::vector<int2> pairs, blk_pq; std::vector<int> blk_start;
    std::vector<int> v_opose, v_opoint, v_opk; std::vector<double> v_pix; bool filled = false;
    const uint8_t *theta_const = nullptr;
    size_t npairs = 0; int nblk = 0, ngrp = 0, wstride = 0, hb = 0, sg_ob = SG_OB, sg_sb = SG_SB;
    int twice_pt = -1, twice_pose = -1;
    // layout: offsets inside the three regions
    size_t o_pose, o_pts, o_const, o_pix, o_opose, o_opoint, o_start, o_ptid, o_opk, o_ohp, o_pfs, o_fobs, o_grp, o_fgrp, o_pairs, o_bs, o_bpq, up_bytes = 0;
    int sg_hp = SG_OB;
    size_t o_st, o_cf, o_outl, o_bwx = 0, zero_bytes = 0;
    int ksplit = 0;
    size_t o_sc0, o_sc1, o_pose_t, o_pts_t, o_hasp, o_f, o_ft, o_Jp, o_Jl, o_Vinv, o_bl, o_T, o_W, o_red, o_Sw, o_dp, o_dl, o_li, o_lf, o_part, o_band, o_wpart, o_xchg, work_bytes = 0;
    ~BAPlan() { delete ba; }
    int lab(int64_t id) const { return new_of.empty() ? (int)id - 1 : new_of[id - 1]; }
    int fail(int code, const char *fmt, long long a = 0, long long b = 0, long long c = 0) { err = code; snprintf(msg, sizeof msg, fmt, a, b, c); return code; }
    // the per-observation arrays (sorted by point): ONE walk over the caller's observations -- the sorted position of observation i is the
    // next free one of its point.  The walk also finds a map point observed twice by one free pose: it has no place in a pose block.
    void fill_obs(int *opose, int *opoint, int *opk, double *pix, int *ohp = nullptr, int *pfs = nullptr, int *fobs = nullptr)
    {
        std::vector<int> fill(start.begin(), start.end() - 1), seen((size_t)P, -1);     // seen[p]: the last point (sorted position) free pose p observed
        for (int i = 0; i < O; i++) {
            const int j = (int)point_ids[i] - 1, k = rank[j], s = fill[k]++;
            ba->perm[s] = i;
            opose[s] = lab(pose_ids[i]); opoint[s] = j; opk[s] = k;
            pix[s] = pixels_yx[2 * i]; pix[(size_t)O + s] = pixels_yx[2 * i + 1];
        }
        for (int k = 0; k < M && twice_pt < 0; k++)
            for (int a = start[k]; a < start[k + 1]; a++) {
                const int p = opose[a];
                if (theta_const[p]) continue;
                if (seen[p] == k) { twice_pt = pt_id[k]; twice_pose = new_of.empty() ? p : ba->pose_order[p]; break; }
                seen[p] = k;
            }
        if (pfs) {                                              // running count of free-pose observations by sorted point
            int c = 0;
            for (int k = 0; k < M; k++) { pfs[k] = c; for (int a = start[k]; a < start[k + 1]; a++) if (!theta_const[opose[a]]) { if (fobs) fobs[c] = a; c++; } }
            pfs[M] = c;
        }
        if (ohp) {                                              // index of an observation among its group's observations of free poses
            int mx = 0;
            for (const int4 &G : grp) {
                int c = 0;
                for (int a = G.y; a < G.y + G.w; a++) ohp[a] = theta_const[opose[a]] ? -1 : c++;
                mx = std::max(mx, c);
            }
            if (small_groups) sg_hp = std::max(8, (mx + 1) & ~1);
        }
    }
};

static int ba_plan(BAPlan &pl)
{
    const int P = pl.P, M = pl.M, O = pl.O;
    const int64_t *pose_ids = pl.pose_ids, *point_ids = pl.point_ids;
    if (!(P > 0 && 6 * P <= SOLVE_MAX_N && M >= 0 && O >= 0 && pl.theta != nullptr && pl.theta_const_in != nullptr)) return pl.fail(SLAM_ERR_ARG, "slam_ba: bad arguments (P %lld, M %lld, O %lld)", P, M, O);
    if (!(O == 0 || (pl.pixels_yx != nullptr && pose_ids != nullptr && point_ids != nullptr))) return pl.fail(SLAM_ERR_ARG, "slam_ba: observations without arrays");
    pl.theta_const = pl.theta_const_in;
    const uint8_t *&theta_const = pl.theta_const;
    slam_ba *ba = pl.ba = new slam_ba();
    const int n = 6 * P;
    // --- host-side structure.  Map points sorted by (first free observing pose f, id); observations sorted by point in
    //     that order (stable).  hb = widest span of free observers of one point = block half-bandwidth of S.
    std::vector<int> &cnt = pl.cnt, &pfirst = pl.pfirst, plast(M), pany(M);
    cnt.assign(M, 0); pfirst.assign(M, 0);
    int hb = 0, bad_obs = -1, nfo = 0;
    auto spans = [&]() {                                     // (the first pass also checks the ids: one walk over the observations, not two)
        std::fill(cnt.begin(), cnt.end(), 0); std::fill(pfirst.begin(), pfirst.end(), P); std::fill(plast.begin(), plast.end(), -1); std::fill(pany.begin(), pany.end(), P);
        nfo = 0;
        for (int i = 0; i < O; i++) {
            if (pose_ids[i] < 1 || pose_ids[i] > P || point_ids[i] < 1 || point_ids[i] > M) { bad_obs = i; return; }
            const int j = (int)point_ids[i] - 1, p = pl.lab(pose_ids[i]);
            cnt[j]++; pany[j] = std::min(pany[j], p);
            if (!theta_const[p]) { pfirst[j] = std::min(pfirst[j], p); plast[j] = std::max(plast[j], p); nfo++; }
        }
        hb = 0;
        for (int j = 0; j < M; j++) {
            if (plast[j] >= 0) hb = std::max(hb, plast[j] - pfirst[j]);
            else pfirst[j] = pany[j] < P ? pany[j] : 0;          // no free observer: any window will do (it gets no slot)
        }
    };
    spans();
    if (bad_obs >= 0) return pl.fail(SLAM_ERR_ARG, "slam_ba: observation %lld has pose id %lld / point id %lld out of range", bad_obs, pose_ids[bad_obs], point_ids[bad_obs]);
    static const bool no_reorder = getenv("SLAMHIP_BA_NO_REORDER") != nullptr;      // (measurement knob)
    if (pl.may_reorder && !no_reorder && M > 0 && O > 0 && (hb > BS_MAXHB || !sg_fold_fits(hb)) && ba_pose_order(P, M, O, pl.theta_const_in, pose_ids, point_ids, ba->pose_order)) {
        // not banded in the caller's pose order, banded in another one: the solver works on relabelled poses, ba_download restores the order
        pl.new_of.resize(P); pl.const_perm.resize(P);
        for (int k = 0; k < P; k++) { pl.new_of[ba->pose_order[k]] = k; pl.const_perm[k] = pl.theta_const_in[ba->pose_order[k]]; }
        theta_const = pl.const_perm.data();
        spans();
    }
    ba->hb = hb; pl.hb = hb;
    {   int f0 = P, f1 = -1;
        for (int p = 0; p < P; p++) if (!theta_const[p]) { f0 = std::min(f0, p); f1 = std::max(f1, p); }
        if (f1 - f0 + 1 >= 2) { ba->p0 = f0; ba->pspan = f1 - f0 + 1; } else { ba->p0 = 0; ba->pspan = P; } }
    std::vector<int> &pt_id = pl.pt_id, &rank = pl.rank, &start = pl.start;
    pt_id.assign(M, 0); rank.assign(M, 0); start.assign(M + 1, 0);
    { std::vector<int> fb(P + 1, 0);
      for (int j = 0; j < M; j++) fb[pfirst[j] + 1]++;
      for (int p = 0; p < P; p++) fb[p + 1] += fb[p];
      for (int j = 0; j < M; j++) { const int k = fb[pfirst[j]]++; pt_id[k] = j; rank[j] = k; } }
    for (int k = 0; k < M; k++) start[k + 1] = start[k] + cnt[pt_id[k]];
    ba->perm.assign(O, 0);
    // --- point groups of k_schur_groups: same f, <= SG_SB points, <= SG_OB observations; evenly sized within one f
    static const bool no_groups = getenv("SLAMHIP_NO_GROUPS") != nullptr;
    // (windows no order makes banded, hb > BS_MAXHB: the point groups still build the system -- their window is the whole block triangle -- when
    //  the free span is small enough for the dense one-workgroup solver, k_dense_solve, and the group's LDS layout fits)
    static const bool no_dense = getenv("SLAMHIP_NO_DENSE") != nullptr;
    const bool dense_ok = !no_dense && hb > BS_MAXHB && ba->pspan >= 2 && ba->pspan <= DS_MAXF && P <= DS_MAXF + 8 && sg_lds_bytes(hb, P) <= 150 * 1024;
    bool grouped = !no_groups && (hb <= BS_MAXHB || dense_ok) && M > 0 && O > 0 && sg_fold_fits(hb);
    pl.nfree_obs = nfo;
    {   // a window one workgroup can keep to itself (k_ba_window, batches only): the point groups of the launch-per-phase kernels are not built
        static const bool no_bw = getenv("SLAMHIP_NO_BA_WINDOW") != nullptr;
        int nfree = 0; for (int p = 0; p < P; p++) nfree += theta_const[p] ? 0 : 1;
        int cmax = 0; for (int j = 0; j < M; j++) cmax = std::max(cmax, cnt[j]);
        pl.window = pl.small_groups && !no_bw && grouped && nfree >= 1 && nfree <= 5 && nfree == ba->pspan && P <= 128 && O <= 40000 && cmax <= 168;
        if (pl.window) pl.sg_ob = std::max(cmax, 1);               // k_ba_window's tiles are sized from it (no point groups are built for such a window)
        static const bool no_ends = getenv("SLAMHIP_BA_WINDOW_SORTED") != nullptr;      // (knob: keep the order by first free observer)
        if (pl.window && M > 1 && !no_ends) {
            // k_ba_window splits a window over two workgroups by map points: the points that see a free pose at all (the Schur phase's records --
            // a fifth of the reference's window, and next to each other in the order by first free observer) go to BOTH ends of the order,
            // alternately, the others between them: each half then holds half of the records and half of the observations (phase clocks per half:
            // A 65 k, Schur 43 k, C 47 k cycles; 1.64 -> 1.62 ms per 128 windows).  Nothing in that kernel depends on the order by first free
            // observer: it is the point groups' window structure, which such a window has not.
            std::vector<int> np_(M); int lo = 0, hi = M - 1, alt = 0, mid = 0;
            for (int k = 0; k < M; k++) { const int j = pt_id[k]; if (plast[j] >= 0) { if (alt++ & 1) np_[hi--] = j; else np_[lo++] = j; } }
            mid = lo;
            for (int k = 0; k < M; k++) { const int j = pt_id[k]; if (plast[j] < 0) np_[mid++] = j; }
            std::reverse(np_.begin() + hi + 1, np_.end());          // (the far end in ascending order of the old ranks, like the near one)
            for (int k = 0; k < M; k++) { pt_id[k] = np_[k]; rank[np_[k]] = k; }
            for (int k = 0; k < M; k++) start[k + 1] = start[k] + cnt[pt_id[k]];
        }
    }
    std::vector<int4> &grp = pl.grp; std::vector<int> &fgrp = pl.fgrp;
    fgrp.assign(P + 1, 0);
    int max_no = 0, max_np = 0;
    if (grouped && !pl.window) {
        static const int sg_points = [] { const char *v = getenv("SLAMHIP_SG_POINTS"); return v ? atoi(v) : 0; }();      // (measurement knob)
        // points per group: a small window in full groups occupies a few compute units and each workgroup walks 7 points per subset; with
        // 16-point groups the reference-shaped window (800 points: 18 -> 50 groups) builds in 0.75 instead of 0.83 ms per 15 iterations,
        // while anything that already fills the chip gets slower with more, smaller groups (more partials for k_schur_reduce, more than one
        // round of workgroups: P = 50 +12 % at 40 points per group) -- so: M / 96, between 16 and SG_SB
        // a batch of windows (small_groups) fills the chip whatever the group size: groups of <= 256 observations run as 256-thread
        // workgroups, three to a compute unit (128 reference-shaped windows: 7.5 ms with the single-window sizes, 3.7 ms so)
        int ob_cap = pl.small_groups ? 256 : SG_OB;
        int sb_eff = sg_points > 0 ? std::min(sg_points, SG_SB)
                   : pl.small_groups ? std::min(std::max(ob_cap / std::max(1, (O + M - 1) / M), 8), SG_SB)
                   : std::min(std::max((M + 95) / 96, 16), SG_SB);
        if (pl.small_groups) for (int j = 0; j < M; j++) if (cnt[j] > ob_cap) { ob_cap = SG_OB; sb_eff = std::min(sb_eff, SG_SB); break; }
        int k = 0;
        for (int f = 0; f < P && grouped; f++) {
            fgrp[f] = (int)grp.size();
            int ke = k;
            while (ke < M && pfirst[pt_id[ke]] == f) ke++;
            const int nf = ke - k, ng = (nf + sb_eff - 1) / sb_eff, tgt = ng ? (nf + ng - 1) / ng : 0;
            while (k < ke) {
                int k1 = k, no = 0;
                while (k1 < ke && k1 - k < tgt && no + cnt[pt_id[k1]] <= ob_cap) { no += cnt[pt_id[k1]]; k1++; }
                if (k1 == k) { grouped = false; break; }          // one point with more than SG_OB observations: pair lists
                grp.push_back(make_int4(k, start[k], f | ((k1 - k) << 16), no));
                max_no = std::max(max_no, no); max_np = std::max(max_np, k1 - k);
                k = k1;
            }
        }
        fgrp[P] = (int)grp.size();
    }
    ba->grouped = grouped;
    if (grouped && pl.small_groups && !pl.window) { pl.sg_ob = std::max(64, (max_no + 7) & ~7); pl.sg_sb = std::max(8, (max_np + 1) & ~1); }
    const int *opose = nullptr;                              // sorted observation -> pose, host copy (needed by the pair lists)
    if (!grouped) {
        pl.v_opose.resize(O); pl.v_opoint.resize(O); pl.v_opk.resize(O); pl.v_pix.resize(2 * (size_t)O);
        pl.fill_obs(pl.v_opose.data(), pl.v_opoint.data(), pl.v_opk.data(), pl.v_pix.data());
        pl.filled = true;
        if (pl.twice_pt >= 0) return pl.fail(SLAM_ERR_ARG, "slam_ba: map point %lld is observed twice by pose %lld", pl.twice_pt + 1, pl.twice_pose + 1);
        opose = pl.v_opose.data();
    }
    // --- pair lists sorted by upper pose block (p <= q), both poses free: only where the groups do not apply
    std::vector<int2> &pairs = pl.pairs, &blk_pq = pl.blk_pq; std::vector<int> &blk_start = pl.blk_start;
    size_t npairs = 0;
    if (!grouped) {
        std::vector<int> bcount((size_t)P * P + 1, 0);
        for (int j = 0; j < M; j++)
            for (int a = start[j]; a < start[j + 1]; a++) {
                if (theta_const[opose[a]]) continue;
                for (int b = start[j]; b < start[j + 1]; b++) {
                    if (theta_const[opose[b]]) continue;
                    const int p = opose[a], q = opose[b];
                    if (p > q || (p == q && a > b)) continue;   // upper blocks; within a diagonal block keep a <= b once
                    bcount[(size_t)p * P + q + 1]++; npairs++;
                }
            }
        std::vector<int> boff((size_t)P * P + 1, 0);
        for (size_t k = 0; k < (size_t)P * P; k++) boff[k + 1] = boff[k] + bcount[k + 1];
        pairs.resize(npairs);
        { std::vector<int> fill(boff.begin(), boff.end() - 1);
          for (int j = 0; j < M; j++)
              for (int a = start[j]; a < start[j + 1]; a++) {
                  if (theta_const[opose[a]]) continue;
                  for (int b = start[j]; b < start[j + 1]; b++) {
                      if (theta_const[opose[b]]) continue;
                      const int p = opose[a], q = opose[b];
                      if (p > q || (p == q && a > b)) continue;
                      pairs[fill[(size_t)p * P + q]++] = make_int2(a, b);
                  }
              } }
        for (int p = 0; p < P; p++)
            for (int q = p; q < P; q++) {
                const size_t k = (size_t)p * P + q;
                if (boff[k + 1] > boff[k]) { blk_start.push_back(boff[k]); blk_pq.push_back(make_int2(p, q)); }
            }
    }
    blk_start.push_back((int)npairs);
    pl.npairs = npairs;
    const int nblk = pl.nblk = (int)blk_pq.size();
    const int ngrp = pl.ngrp = (int)grp.size(), hbw = hb + 1;
    pl.wstride = grouped ? hbw * (hbw + 1) / 2 * 36 + hbw * 12 : 0;
    const int nbo = (O + 255) / 256, nbp = (std::max(M, n) + 255) / 256;
    ba->nblocks_obs = std::max(nbo, 1); ba->nblocks_pts = std::max(nbp, 1);
    // --- layout: uploaded arrays (one contiguous block: a single copy from the staging buffer), the zero-initialised ones (one memset), the rest
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    pl.o_pose = take(n * 8); pl.o_pts = take((size_t)3 * M * 8 + 8); pl.o_const = take(P); pl.o_pix = take((size_t)2 * O * 8 + 8);
    pl.o_opose = take((size_t)O * 4 + 4); pl.o_opoint = take((size_t)O * 4 + 4); pl.o_start = take((size_t)(M + 1) * 4);
    pl.o_ptid = take((size_t)M * 4 + 4); pl.o_opk = take((size_t)O * 4 + 4); pl.o_ohp = take(pl.window ? 8 : (size_t)O * 4 + 4); pl.o_pfs = take(pl.small_groups ? (size_t)(M + 1) * 4 : 8); pl.o_fobs = take(pl.small_groups ? (size_t)pl.nfree_obs * 4 + 4 : 8); pl.o_grp = take((size_t)ngrp * 16 + 16); pl.o_fgrp = take((size_t)(P + 1) * 4);      // (pfs / fobs: k_ba_window's lists, batches only)
    pl.o_pairs = take(npairs * 8 + 8); pl.o_bs = take((size_t)(nblk + 1) * 4); pl.o_bpq = take((size_t)nblk * 8 + 8);
    pl.up_bytes = off; off = 0;
    pl.o_st = take(sizeof(LMState)); pl.o_cf = take(64); pl.o_outl = take((size_t)O + 1);
    pl.o_dp = take(n * 8);                                       // dp of the constant poses outside the solve's span stays zero
    if (pl.window) {                                             // k_ba_window on two workgroups: flags + two double-buffered partials of 32 x 25 doubles each (zeroed: the flags count exchanges)
        pl.o_bwx = take(BWX_DOUBLES * 8);
        pl.ksplit = (int)(std::lower_bound(start.begin(), start.end(), (O + 1) / 2) - start.begin());
        pl.ksplit = std::min(std::max(pl.ksplit, 0), M);
    }
    pl.o_red = take(((size_t)n * n + 2 * n + 8) * 8);            // the private reduce buffer: only its band is ever rewritten
    pl.zero_bytes = off; off = 0;
    pl.o_sc0 = take(n * 8); pl.o_sc1 = take(n * 8);
    pl.o_pose_t = take(n * 8); pl.o_pts_t = take((size_t)3 * M * 8 + 8); pl.o_hasp = take((size_t)O + 1);
    pl.o_f = take((size_t)2 * O * 8 + 8); pl.o_ft = take(8);    // (trial residuals are not kept: every build re-evaluates d.f)
    pl.o_Jp = take((size_t)12 * O * 8 + 8); pl.o_Jl = take((size_t)6 * O * 8 + 8);
    pl.o_Vinv = take((size_t)6 * M * 8 + 8); pl.o_bl = take((size_t)3 * M * 8 + 8);
    pl.o_T = take(grouped ? 8 : (size_t)18 * O * 8 + 8); pl.o_W = take(grouped ? 8 : (size_t)18 * O * 8 + 8);   // T / W records: pair-list path only
    // (the tiled Cholesky's working matrices exist for every window: a window whose band does not fit k_band_solve's LDS takes that path)
    pl.o_Sw = take((size_t)(n + 1) * n * 8); pl.o_dl = take((size_t)3 * M * 8 + 8);
    pl.o_li = take((size_t)((n + CT - 1) / CT) * CT * CT * 8); pl.o_lf = take((size_t)(n + 1) * n * 8);
    pl.o_part = take(((size_t)std::max(ba->nblocks_pts, ngrp) + 2 * (size_t)std::max(ba->nblocks_obs, ngrp) + 8) * 8);
    pl.o_band = take((size_t)P * ((size_t)(BS_MAXHB + 1) * 36 + 8) * 8);
    pl.o_wpart = take((size_t)ngrp * pl.wstride * 8 + 8);
    pl.o_xchg = take(2048 * 8);
    pl.work_bytes = off;
    return SLAM_OK;
}

// bind the device pointers (region bases Aup / Azero / Awork) and write the uploaded region into `stage` (host memory, up_bytes)
static int ba_emit(BAPlan &pl, char *Aup, char *Azero, char *Awork, char *stage)
{
    slam_ba *ba = pl.ba;
    const int P = pl.P, M = pl.M, O = pl.O, n = 6 * P;
    BADev &d = ba->d;
    d.cam = {pl.fx, pl.fy, pl.cx, pl.cy}; d.P = P; d.M = M; d.O = O; d.n = n;
    d.sc0 = (double *)(Awork + pl.o_sc0); d.sc1 = (double *)(Awork + pl.o_sc1);
    d.pose = (double *)(Aup + pl.o_pose); d.pose_t = (double *)(Awork + pl.o_pose_t); d.pts = (double *)(Aup + pl.o_pts); d.pts_t = (double *)(Awork + pl.o_pts_t);
    d.pconst = (const uint8_t *)(Aup + pl.o_const); d.pix = (const double *)(Aup + pl.o_pix);
    d.opose = (const int *)(Aup + pl.o_opose); d.opoint = (const int *)(Aup + pl.o_opoint); d.pt_start = (const int *)(Aup + pl.o_start);
    d.outl = (uint8_t *)(Azero + pl.o_outl); d.hasp = (uint8_t *)(Awork + pl.o_hasp);
    d.f = (double *)(Awork + pl.o_f); d.ft = (double *)(Awork + pl.o_ft); d.Jp = (double *)(Awork + pl.o_Jp); d.Jl = (double *)(Awork + pl.o_Jl);
    d.Vinv = (double *)(Awork + pl.o_Vinv); d.bl = (double *)(Awork + pl.o_bl); d.T = (double *)(Awork + pl.o_T); d.Wm = (double *)(Awork + pl.o_W);
    d.pairs = (const int2 *)(Aup + pl.o_pairs); d.blk_start = (const int *)(Aup + pl.o_bs); d.blk_pq = (const int2 *)(Aup + pl.o_bpq); d.nblk = pl.nblk;
    ba->reduce = (double *)(Azero + pl.o_red);
    ba->zeroed = ba->reduce;                                   // (the set-up's memset of the zero region covers it)
    d.S = ba->reduce; d.g = ba->reduce + (size_t)n * n; d.udiag = d.g + n;
    d.Swork = (double *)(Awork + pl.o_Sw); d.dp = (double *)(Azero + pl.o_dp); d.dl = (double *)(Awork + pl.o_dl);
    d.part = (double *)(Awork + pl.o_part); d.st = (LMState *)(Azero + pl.o_st);
    ba->chol_flag = (int *)(Azero + pl.o_cf); ba->linv = (double *)(Awork + pl.o_li); ba->lfac = (double *)(Awork + pl.o_lf); ba->band = (double *)(Awork + pl.o_band);
    d.pt_id = (const int *)(Aup + pl.o_ptid); d.opk = (const int *)(Aup + pl.o_opk); d.grp = (const int4 *)(Aup + pl.o_grp); d.fgrp = (const int *)(Aup + pl.o_fgrp);
    d.ngrp = pl.ngrp; d.whb = pl.hb; d.wstride = pl.wstride; d.wpart = (double *)(Awork + pl.o_wpart);
    d.sg_ob = pl.sg_ob; d.sg_sb = pl.sg_sb; d.ohp = (const int *)(Aup + pl.o_ohp); d.pfs = (const int *)(Aup + pl.o_pfs); d.fobs = (const int *)(Aup + pl.o_fobs);
    ba->nparts = ba->grouped ? pl.ngrp : ba->nblocks_obs;
    ba->xchg = (double *)(Awork + pl.o_xchg);
#define UP(o, src, bytes) do { if ((bytes) > 0) memcpy(stage + (o), (src), (bytes)); } while (0)
    if (!pl.new_of.empty()) { double *dst = (double *)(stage + pl.o_pose); for (int k = 0; k < P; k++) memcpy(dst + 6 * k, pl.theta + 6 * ba->pose_order[k], 48); }
    else UP(pl.o_pose, pl.theta, (size_t)n * 8);
    UP(pl.o_pts, pl.theta + n, (size_t)3 * M * 8);
    UP(pl.o_const, pl.theta_const, (size_t)P); UP(pl.o_start, pl.start.data(), (size_t)(M + 1) * 4);
    if (pl.filled) { UP(pl.o_pix, pl.v_pix.data(), (size_t)2 * O * 8); UP(pl.o_opose, pl.v_opose.data(), (size_t)O * 4); UP(pl.o_opoint, pl.v_opoint.data(), (size_t)O * 4); UP(pl.o_opk, pl.v_opk.data(), (size_t)O * 4); }
    else {                                                   // (grouped: nothing on the host needs these arrays) written in place
        pl.fill_obs((int *)(stage + pl.o_opose), (int *)(stage + pl.o_opoint), (int *)(stage + pl.o_opk), (double *)(stage + pl.o_pix), pl.window ? nullptr : (int *)(stage + pl.o_ohp), pl.small_groups ? (int *)(stage + pl.o_pfs) : nullptr, pl.small_groups ? (int *)(stage + pl.o_fobs) : nullptr);
        if (pl.twice_pt >= 0) return pl.fail(SLAM_ERR_ARG, "slam_ba: map point %lld is observed twice by pose %lld", pl.twice_pt + 1, pl.twice_pose + 1);
    }
    UP(pl.o_pairs, pl.pairs.data(), pl.npairs * 8); UP(pl.o_bs, pl.blk_start.data(), (size_t)(pl.nblk + 1) * 4); UP(pl.o_bpq, pl.blk_pq.data(), (size_t)pl.nblk * 8);
    UP(pl.o_ptid, pl.pt_id.data(), (size_t)M * 4); UP(pl.o_grp, pl.grp.data(), (size_t)pl.ngrp * 16); UP(pl.o_fgrp, pl.fgrp.data(), (size_t)(P + 1) * 4);
#undef UP
    d.sg_hp = pl.sg_hp;
    return SLAM_OK;
}

static int ba_setup(slam_ctx *ctx, double fx, double fy, double cx, double cy, int P, int M, int O,
                    const double *theta, const uint8_t *theta_const_in, const double *pixels_yx,
                    const int64_t *pose_ids, const int64_t *point_ids, slam_ba **out, bool ctx_mem = false, bool may_reorder = false)
{
    BAPlan pl;
    pl.fx = fx; pl.fy = fy; pl.cx = cx; pl.cy = cy; pl.P = P; pl.M = M; pl.O = O; pl.theta = theta; pl.theta_const_in = theta_const_in;
    pl.pixels_yx = pixels_yx; pl.pose_ids = pose_ids; pl.point_ids = point_ids; pl.may_reorder = may_reorder;
    if (ba_plan(pl)) return slam_fail(ctx, pl.err, "%s", pl.msg);
    slam_ba *ba = pl.ba;
    ba->device = ctx->device;
    const size_t up_end = pl.up_bytes, zero_end = up_end + pl.zero_bytes, total = zero_end + pl.work_bytes;
    char *A = nullptr;
    if (ctx_mem) {                                             // slam_local_ba: the context's grow-only scratch, no hipMalloc / hipFree per call
        const int rcs = slam_scratch(ctx, total, (void **)&A);
        if (rcs) return rcs;
        ba->owns_arena = false;
    } else {
        hipError_t e = hipMalloc((void **)&A, total);
        if (e != hipSuccess) return slam_fail(ctx, SLAM_ERR_HIP, "slam_ba: hipMalloc(%zu): %s", total, hipGetErrorString(e));
    }
    ba->arena = A;
    struct Guard { slam_ba *b; ~Guard() { if (b && b->arena && b->owns_arena) (void)hipFree(b->arena); } } guard{ba};   // a failing upload frees the arena (the plan owns the object)
    hipStream_t st = ctx->stream;
    char *stage = nullptr;
    std::vector<char> pageable;
    if (ctx_mem) { const int rcs = slam_pinned(ctx, up_end, (void **)&stage); if (rcs) return rcs; }
    else { pageable.resize(up_end); stage = pageable.data(); }
    if (ba_emit(pl, A, A + up_end, A + zero_end, stage)) return slam_fail(ctx, pl.err, "%s", pl.msg);
    HIP_TRY(ctx, hipMemcpyAsync(A, stage, up_end, hipMemcpyHostToDevice, st));                  // pinned -> device: one DMA, nothing to wait for
    HIP_TRY(ctx, hipMemsetAsync(A + up_end, 0, zero_end - up_end, st));                         // LM state, flags, outlier marks, dp, the reduce buffer
    if (!ctx_mem) HIP_TRY(ctx, slam_stream_wait(st));        // the pageable staging block goes out of scope
    guard.b = nullptr;
    pl.ba = nullptr;                                           // ownership passes to the caller
    *out = ba;
    return SLAM_OK;
}


// linearise at the current parameters and build [S; g; udiag] into `red`
static int ba_enqueue_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double inv_delta, int use_state, double *red)
{
    BADev d = ba->d;
    const int n = d.n;
    d.S = red; d.g = red + (size_t)n * n; d.udiag = d.g + n;
    hipStream_t st = ctx->stream;
    // the grouped build rewrites every block inside THIS problem's band (d.whb), g and diag(U) each time: the rest of the buffer only
    // needs zeroing once -- as long as nobody else writes to it.  A caller-owned buffer (the sharded path all-reduces it in place, and a
    // peer's band may be wider than ours: those blocks would keep the previous iteration's SUM and be summed again) is zeroed every time.
    const bool private_red = red == ba->reduce;
    if (!ba->grouped || !private_red || ba->zeroed != red) { HIP_TRY(ctx, hipMemsetAsync(red, 0, ((size_t)n * n + 2 * n + 8) * 8, st)); ba->zeroed = private_red ? red : nullptr; }
    if (ba->grouped) {
        // the attribute belongs to the function object of the CURRENT device: once per device, result checked
        // (the reference's three tasks call the library concurrently, SLAM.jl:166: the flag is atomic; setting the attribute twice is harmless)
        static std::atomic<bool> attr_set[64];
        const int dv = ctx->device & 63;
        if (!attr_set[dv].load(std::memory_order_acquire)) { HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_schur_groups, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(sg_lds_bytes(BS_MAXHB, SOLVE_MAX_N / 6), (size_t)150 * 1024))); attr_set[dv].store(true, std::memory_order_release); }
        hipLaunchKernelGGL(k_schur_groups, dim3(d.ngrp), dim3(SG_T), sg_lds_bytes(d.whb, d.P, d.sg_ob, d.sg_sb, 512, d.sg_hp), st, d, inv_delta, ignore_outliers, use_state);
        if (!use_state) hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 0, d.ngrp, ba->nblocks_pts, 0, red + (size_t)n * n + 2 * n);
        const int nthr = d.P * (d.whb + 1) * 36 + d.P * 12;
        hipLaunchKernelGGL(k_schur_reduce, dim3((nthr + 255) / 256), dim3(256), 0, st, d, use_state);
        return SLAM_OK;
    }
    hipLaunchKernelGGL(k_linearize, dim3(ba->nblocks_obs), dim3(256), 0, st, d, ignore_outliers, use_state);
    if (!use_state) hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 0, ba->nblocks_obs, ba->nblocks_pts, 0, red + (size_t)n * n + 2 * n);
    if (d.M > 0) hipLaunchKernelGGL(k_points, dim3((d.M + 255) / 256), dim3(256), 0, st, d, inv_delta, use_state);
    if (d.O > 0) hipLaunchKernelGGL(k_obs_factors, dim3((d.O + 255) / 256), dim3(256), 0, st, d, use_state);
    if (d.nblk > 0) hipLaunchKernelGGL(k_blocks, dim3(((d.nblk + 7) / 8) * 8), dim3(256), 0, st, d, use_state);
    return SLAM_OK;
}

static int ba_enqueue_solve(slam_ctx *ctx, slam_ba *ba, const double *red, int ignore_outliers, double inv_delta, int use_state,
                            int lm, double *out4)
{
    BADev d = ba->d;
    const int n = d.n;
    hipStream_t st = ctx->stream;
    static const bool no_band = getenv("SLAMHIP_NO_BAND") != nullptr;
    const int Ps = ba->pspan > 0 ? ba->pspan : d.P, p0 = ba->pspan > 0 ? ba->p0 : 0;       // the poses the banded solve covers: first .. last free pose
    const int hb = std::min(std::max(ba->hb, 1), Ps - 1);      // >= 1: the factor wave reads block row k + 1 while row k + 1 + hb enters the ring
    const size_t band_lds = band_lds_bytes(n, Ps, hb);
    if (ba->grouped && hb > BS_MAXHB && Ps <= DS_MAXF) {       // not banded, small: dense one-workgroup solve (ba_plan admitted the groups for exactly this case)
        BandArgs B = {}; B.S = red + (size_t)6 * p0 * (n + 1); B.g = red + (size_t)n * n + 6 * p0; B.ud = red + (size_t)n * n + n + 6 * p0; B.Lg = nullptr; B.nb = Ps; B.hb = hb; B.p0 = p0;
        B.inv_delta_host = inv_delta; B.fail = ba->chol_flag;
        static std::atomic<bool> ds_attr[64];
        const int dv = ctx->device & 63;
        if (!ds_attr[dv].load(std::memory_order_acquire)) { HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dense_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dense_lds_bytes(DS_MAXF))); ds_attr[dv].store(true, std::memory_order_release); }
        hipLaunchKernelGGL(k_dense_solve, dim3(1), dim3(DS_T), dense_lds_bytes(Ps), st, d, B, use_state);
    } else
    if (!no_band && hb <= BS_MAXHB && band_lds <= 150 * 1024) {
        BandArgs B; B.S = red + (size_t)6 * p0 * (n + 1); B.g = red + (size_t)n * n + 6 * p0; B.ud = red + (size_t)n * n + n + 6 * p0; B.Lg = ba->band; B.nb = Ps; B.hb = hb; B.p0 = p0;
        B.inv_delta_host = inv_delta; B.fail = ba->chol_flag; B.lds_bytes = (int)band_lds;
        B.xchg = ba->xchg; B.epoch = ++ba->epoch;
        static const int twist_shift = [] { const char *v = getenv("SLAMHIP_TWIST_SHIFT"); return v ? atoi(v) : 0; }();    // (measurement knob: side 0 takes 2 x shift columns more than side 1; +1 paid while the hand-over cost 13 k cycles, with 7 k an even split is 1 % ahead)
        B.shift = twist_shift;
        static const bool no_twist = getenv("SLAMHIP_NO_TWIST") != nullptr;
        // (the two workgroups wait for each other: both must be resident, which a stream confined to one compute unit cannot promise)
        static const int twist_min = [] { const char *v = getenv("SLAMHIP_TWIST_MIN"); return v ? atoi(v) : 0; }();    // (measurement knob)
        const bool twist = !no_twist && hb * 6 <= 58 && Ps >= (twist_min > 0 ? std::max(twist_min, hb + 8) : std::max(2 * (hb + 1) - 1, hb + 8))      /* measured: pays from 19 free poses at hb = 9 (19: 92.1 -> 89.3 us per iteration, 18: equal) since the hand-overs stay in one L2 (24 before) */ && ctx->xwg_ok;
        static long long *trace_dev = nullptr; static int trace_n = 0;
        static const bool trace_on = getenv("SLAMHIP_BAND_TRACE") != nullptr;
        if (trace_on && !trace_dev) (void)hipHostMalloc((void **)&trace_dev, 1024);
        B.trace = trace_dev;
        if (trace_on && trace_n++ == 8) {      // side 0 of the ninth launch, shader cycles
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "band trace (cycles): factor wave: panel %lld factor %lld barrier %lld, then middle + back-substitution %lld | update wave 1: flag %lld update %lld barrier %lld | prefetch wave: flag %lld put/fetch %lld | backsub: chat %lld G %lld recurrence %lld\n",
                    trace_dev[5], trace_dev[4], trace_dev[1], trace_dev[3], trace_dev[2], trace_dev[6], trace_dev[7], trace_dev[8], trace_dev[9], trace_dev[10], trace_dev[11], trace_dev[12]);
            fprintf(stderr, "  side 0 timeline (cycles): set-up (damping, first window, tables, L2 warm-up) %lld; then, since the end of the set-up: column loop starts %lld, own columns done %lld, middle assembled + factored %lld, forward done %lld\n", trace_dev[26], trace_dev[16], trace_dev[17], trace_dev[18], trace_dev[19]);
            fprintf(stderr, "  set-up: pair table %lld, window requested %lld, tables %lld, warm-up requested %lld, damping in LDS %lld, window in the ring %lld\n", trace_dev[103], trace_dev[104], trace_dev[105], trace_dev[106], trace_dev[107], trace_dev[26]);
            fprintf(stderr, "  middle: entries prepared %lld, flag seen %lld, fence done %lld, assembled %lld\n", trace_dev[22], trace_dev[23], trace_dev[24], trace_dev[25]);
            fprintf(stderr, "  XCC_ID of side 0 / side 1: %lld / %lld; of the idle workgroups 1-7:", trace_dev[20] & 15, trace_dev[21] & 15);
            for (int w = 1; w < 8; w++) fprintf(stderr, " %lld", trace_dev[80 + w] & 15);
            fprintf(stderr, "\n");
            const long long t0 = trace_dev[32];      // step 10 per wave, relative to wave 0's start of the step: start, (wave 0: flag published), arrival at the barrier, release
            fprintf(stderr, "  factor wave, step 10: row loaded %lld, substituted %lld, flag %lld, D entry %lld, D in every lane %lld\n",
                    trace_dev[96] - t0, trace_dev[97] - t0, trace_dev[40] - t0, trace_dev[98] - t0, trace_dev[99] - t0);
            fprintf(stderr, "  wave 3: registers copied %lld, blocks in the ring %lld, next row requested %lld\n", trace_dev[72] - t0, trace_dev[73] - t0, trace_dev[74] - t0);
            for (int w = 0; w < 8; w++) fprintf(stderr, "  wave %d (simd %lld): start %lld%s arrive %lld release %lld\n", w, (trace_dev[64 + w] >> 4) & 3, trace_dev[32 + w] - t0,
                                                w == 0 ? (" flag " + std::to_string(trace_dev[40] - t0)).c_str() : "", trace_dev[48 + w] - t0, trace_dev[56 + w] - t0);
        }
        static std::atomic<bool> attr_set[64];
        const int dv = ctx->device & 63;
        if (!attr_set[dv].load(std::memory_order_acquire)) { HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_band_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr_set[dv].store(true, std::memory_order_release); }
        static const bool twist_spread = getenv("SLAMHIP_TWIST_SPREAD") != nullptr;     // (test knob: the two sides on different XCDs)
        hipLaunchKernelGGL(k_band_solve, dim3(twist ? (twist_spread ? 2 : 9) : 1), dim3(BS_T), band_lds, st, d, B, use_state);
    } else {
        CholArgs C; C.A = d.Swork; C.Lf = ba->lfac; C.n = n; C.ld = n + 1; C.fail = ba->chol_flag;
        const size_t tot = (size_t)(n + 1) * n;
        hipLaunchKernelGGL(k_chol_prepare, dim3((tot + 255) / 256), dim3(256), 0, st, d, red, red + (size_t)n * n, red + (size_t)n * n + n, inv_delta, use_state);
        hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(256), 0, st, d, C, ba->linv, use_state);
        const int nbc = (n + CT - 1) / CT, nbr = (n + 1 + CT - 1) / CT;
        for (int k = 0; k < nbc; k++) {
            int tiles = 0;
            for (int c = k; c < nbc; c++) tiles += nbr - c;
            if (tiles > 1) hipLaunchKernelGGL(k_chol_step, dim3(tiles), dim3(256), 0, st, d, C, ba->linv, k, nbr, use_state);
        }
        hipLaunchKernelGGL(k_chol_backsolve, dim3(1), dim3(256), 0, st, d, C, (const double *)ba->linv, use_state);
    }
    if (ba->grouped) {
        hipLaunchKernelGGL(k_update_groups, dim3(d.ngrp), dim3(SG_T), 0, st, d, ignore_outliers, use_state);
        hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 1, d.ngrp, d.ngrp, lm | (use_state ? 2 : 0), out4);
        return SLAM_OK;
    }
    hipLaunchKernelGGL(k_backsub, dim3(ba->nblocks_pts), dim3(256), 0, st, d, use_state);
    hipLaunchKernelGGL(k_trial, dim3(ba->nblocks_obs), dim3(256), 0, st, d, ignore_outliers, use_state, ba->nblocks_pts);
    hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 1, ba->nblocks_obs, ba->nblocks_pts, lm | (use_state ? 2 : 0), out4);
    return SLAM_OK;
}

static int ba_enqueue_commit(slam_ctx *ctx, slam_ba *ba, int accept, int use_state, int iter_tag)
{
    (void)iter_tag;
    if (use_state) return SLAM_OK;                           // device-paced: lm_decide has swapped the buffers already
    hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, ctx->stream, ba->d, accept);
    return SLAM_OK;
}

extern "C" {

int slam_ba_destroy(slam_ba *ba)
{
    if (!ba) return SLAM_OK;
    if (ba->owns_arena) {
        (void)hipSetDevice(ba->device);
        (void)hipDeviceSynchronize();
        if (ba->arena) (void)hipFree(ba->arena);
    }
    delete ba;
    return SLAM_OK;
}

int64_t slam_ba_reduce_len(int P) { const int64_t n = 6 * (int64_t)P; return n * n + 2 * n + 8; }

int slam_ba_create(slam_ctx *ctx, double fx, double fy, double cx, double cy, int P, int M_local, int O_local,
                   const double *theta, const uint8_t *theta_const, const double *pixels_yx,
                   const int64_t *pose_ids, const int64_t *point_ids_local, slam_ba **out)
{
    ARG_TRY(ctx, ctx != nullptr && out != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ba_setup(ctx, fx, fy, cx, cy, P, M_local, O_local, theta, theta_const, pixels_yx, pose_ids, point_ids_local, out);
}

int slam_ba_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double inv_delta, double *reduce_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_build(ctx, ba, ignore_outliers, inv_delta, 0, reduce_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_ba_solve(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, double inv_delta, double *trial_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr && trial_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ignore_outliers for the trial residual follows the flags: outliers are only ever set by slam_ba_flag_outliers
    int rc = ba_enqueue_solve(ctx, ba, reduce_dev, 1, inv_delta, 0, 0, trial_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_ba_commit(slam_ctx *ctx, slam_ba *ba, int accept)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ba_enqueue_commit(ctx, ba, accept, 0, 0);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

// ---- device-paced LM for the sharded path: every call returns after enqueueing on ctx's stream; the accept / reject decision
// is taken on the device from the gathered trial costs, so a pass needs no host synchronisation (slam.h has the protocol) ----
int slam_ba_lm_begin(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double *reduce_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_build(ctx, ba, ignore_outliers, 1.0 / LM_DELTA0, 0, reduce_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_start(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, int first_pass)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n = ba->d.n;
    hipLaunchKernelGGL(k_lm_start, dim3(1), dim3(1), 0, ctx->stream, ba->d, reduce_dev + (size_t)n * n + 2 * n, first_pass);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double *reduce_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_build(ctx, ba, ignore_outliers, 0.0, 1, reduce_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_solve(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, int ignore_outliers, double *trial_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr && trial_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_solve(ctx, ba, reduce_dev, ignore_outliers, 0.0, 1, 0, trial_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_step(slam_ctx *ctx, slam_ba *ba, const double *gathered_dev, int nranks, int iter_tag)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && gathered_dev != nullptr && nranks >= 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_control_gathered, dim3(1), dim3(1), 0, ctx->stream, ba->d, gathered_dev, nranks);
    ba_enqueue_commit(ctx, ba, 0, 1, iter_tag);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
// synchronises; out8 = {ssr, iters, converged, delta, chol_fail, ssr_init, trial_ssr, max|dx|}
int slam_ba_lm_state(slam_ctx *ctx, slam_ba *ba, double *out8)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && out8 != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LMState h;
    HIP_TRY(ctx, hipMemcpyAsync(&h, ba->d.st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    out8[0] = h.ssr; out8[1] = h.iters; out8[2] = h.converged; out8[3] = h.delta; out8[4] = h.chol_fail; out8[5] = h.ssr_init;
    out8[6] = h.trial_ssr; out8[7] = h.maxdx;
    return SLAM_OK;
}
// block half-bandwidth of this shard's reduced system (S_pq = 0 for |p - q| > hb); the all-reduced system has the maximum over
// the ranks, which the driver sets on every rank before the first solve
// host only: the pose order slam_local_ba solves in, and the block half-bandwidth of the reduced camera system in that order
int slam_ba_plan_order(int P, int M, int O, const uint8_t *theta_const, const int64_t *pose_ids, const int64_t *point_ids, int32_t *order_out, int *hb_out)
{
    if (P <= 0 || M < 0 || O < 0 || !theta_const || (O > 0 && (!pose_ids || !point_ids))) return SLAM_ERR_ARG;
    for (int i = 0; i < O; i++) if (pose_ids[i] < 1 || pose_ids[i] > P || point_ids[i] < 1 || point_ids[i] > M) return SLAM_ERR_ARG;
    std::vector<int> new_of(P), order(P);
    for (int p = 0; p < P; p++) new_of[p] = order[p] = p;
    auto halfband = [&]() {
        std::vector<int> lo(M, P), hi(M, -1);
        for (int i = 0; i < O; i++) {
            if (theta_const[pose_ids[i] - 1]) continue;
            const int j = (int)point_ids[i] - 1, q = new_of[pose_ids[i] - 1];
            lo[j] = std::min(lo[j], q); hi[j] = std::max(hi[j], q);
        }
        int hb = 0;
        for (int j = 0; j < M; j++) if (hi[j] >= 0) hb = std::max(hb, hi[j] - lo[j]);
        return hb;
    };
    int hb = halfband(), reordered = 0;
    static const bool no_reorder = getenv("SLAMHIP_BA_NO_REORDER") != nullptr;
    std::vector<int> cand;
    if (!no_reorder && M > 0 && O > 0 && (hb > BS_MAXHB || !sg_fold_fits(hb)) && ba_pose_order(P, M, O, theta_const, pose_ids, point_ids, cand)) {
        order = cand;
        for (int k = 0; k < P; k++) new_of[order[k]] = k;
        hb = halfband(); reordered = 1;
    }
    if (order_out) for (int k = 0; k < P; k++) order_out[k] = order[k];
    if (hb_out) *hb_out = hb;
    return reordered;
}

int slam_ba_halfband(const slam_ba *ba) { return ba ? ba->hb : SLAM_ERR_ARG; }
int slam_ba_set_halfband(slam_ba *ba, int hb) { if (!ba || hb < 0) return SLAM_ERR_ARG; ba->hb = hb; return SLAM_OK; }

int slam_ba_flag_outliers(slam_ctx *ctx, slam_ba *ba, double repr_eps, double depth_eps, int *n_out)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_outliers, dim3(ba->nblocks_obs), dim3(256), 0, ctx->stream, ba->d, repr_eps, depth_eps);
    hipLaunchKernelGGL(k_outlier_count, dim3(1), dim3(256), 0, ctx->stream, ba->d, ba->nblocks_obs);
    HIP_TRY(ctx, hipGetLastError());
    LMState h;
    HIP_TRY(ctx, hipMemcpyAsync(&h, ba->d.st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    if (n_out) *n_out = h.n_outliers;
    return SLAM_OK;
}

// cur_known: LMState::cur if the caller has the state on the host already, -1: ask the device (one more synchronisation)
static int ba_download(slam_ctx *ctx, slam_ba *ba, double *theta, uint8_t *outliers, int cur_known)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const BADev &d = ba->d;
    if (theta) {
        int cur = cur_known;                                 // which buffer pair holds the committed parameters (LMState::cur)
        if (cur < 0) {
            HIP_TRY(ctx, hipMemcpyAsync(&cur, &d.st->cur, sizeof cur, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, slam_stream_wait(ctx->stream));
        }
        if (ba->pose_order.empty()) HIP_TRY(ctx, hipMemcpyAsync(theta, cur ? d.pose_t : d.pose, (size_t)d.n * 8, hipMemcpyDeviceToHost, ctx->stream));
        else {                                               // the solver's pose order -> the caller's
            std::vector<double> tmp(d.n);
            HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), cur ? d.pose_t : d.pose, (size_t)d.n * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, slam_stream_wait(ctx->stream));
            for (int k = 0; k < d.P; k++) memcpy(theta + 6 * ba->pose_order[k], &tmp[6 * k], 48);
        }
        if (d.M > 0) HIP_TRY(ctx, hipMemcpyAsync(theta + d.n, cur ? d.pts_t : d.pts, (size_t)3 * d.M * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    std::vector<uint8_t> tmp;
    if (outliers && d.O > 0) { tmp.resize(d.O); HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), d.outl, (size_t)d.O, hipMemcpyDeviceToHost, ctx->stream)); }
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    if (outliers) for (int s = 0; s < d.O; s++) outliers[ba->perm[s]] = tmp[s];
    return SLAM_OK;
}
int slam_ba_download(slam_ctx *ctx, slam_ba *ba, double *theta, uint8_t *outliers) { return ba_download(ctx, ba, theta, outliers, -1); }

int slam_local_ba(slam_ctx *ctx, double fx, double fy, double cx, double cy, int P, int M, int O,
                  double *theta, const uint8_t *theta_const, const double *pixels_yx,
                  const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                  int iters_fast, int iterations, double repr_eps, double *stats)
{
    ARG_TRY(ctx, ctx != nullptr && outliers != nullptr && iters_fast >= 0 && iterations >= 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    slam_ba *ba = nullptr;
    static const bool host_times = getenv("SLAMHIP_BA_HOSTTIME") != nullptr;
    const auto tw0 = std::chrono::steady_clock::now();
    int rc = ba_setup(ctx, fx, fy, cx, cy, P, M, O, theta, theta_const, pixels_yx, pose_ids, point_ids, &ba, true, true);
    if (rc) return rc;
    const auto tw1 = std::chrono::steady_clock::now();
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    BADev d = ba->d;
    auto run_pass = [&](int ignore, int iters) {
        // f / ssr at the start of the pass (LeastSquaresOptim evaluates f!(fcur, x) first)
        hipLaunchKernelGGL(k_linearize, dim3(ba->nblocks_obs), dim3(256), 0, st, d, ignore, 0);
        hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 0, ba->nblocks_obs, ba->nblocks_pts, 0, (double *)nullptr);
        hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(1), 0, st, d, ignore ? 1 : 0);
        for (int it = 1; it <= iters; it++) {
            ba_enqueue_build(ctx, ba, ignore, 0.0, 1, ba->reduce);
            ba_enqueue_solve(ctx, ba, ba->reduce, ignore, 0.0, 1, 1, nullptr);
            ba_enqueue_commit(ctx, ba, 0, 1, it);
        }
    };
    run_pass(0, iters_fast);
    hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(1), 0, st, d, 3);
    // flag outliers at theta_1 (bundle_adjustment.jl:45)
    hipLaunchKernelGGL(k_outliers, dim3(ba->nblocks_obs), dim3(256), 0, st, d, repr_eps, 1e-6);
    hipLaunchKernelGGL(k_outlier_count, dim3(1), dim3(256), 0, st, d, ba->nblocks_obs);
    run_pass(1, iterations);
    hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(1), 0, st, d, 2);
    (void)hipEventRecord(e1, st);
    LMState h;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d.st, sizeof h, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = slam_stream_wait(st);
    float ms = 0;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) { slam_ba_destroy(ba); return slam_fail(ctx, SLAM_ERR_HIP, "slam_local_ba: %s", hipGetErrorString(e)); }
    // A failed factorisation leaves the caller's theta and outliers untouched (the reference's LSMR step cannot fail and
    // never leaves cache.theta half-updated): the state is only copied back from a run that completed.
    const auto tw2 = std::chrono::steady_clock::now();
    rc = h.chol_fail ? SLAM_OK : ba_download(ctx, ba, theta, outliers, h.cur);
    const auto tw3 = std::chrono::steady_clock::now();
    slam_ba_destroy(ba);
    if (host_times) {
        const auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "slam_local_ba host: setup %ld us, enqueue + wait %ld us (device %.0f us), download %ld us, destroy %ld us\n",
                us(tw0, tw1), us(tw1, tw2), ms * 1e3, us(tw2, tw3), us(tw3, std::chrono::steady_clock::now()));
    }
    if (rc) return rc;
    if (stats) {
        stats[0] = h.ssr_init; stats[1] = h.ssr_pass1; stats[2] = h.ssr_final; stats[3] = h.iters_pass1; stats[4] = h.iters_pass2;
        stats[5] = h.n_outliers; stats[6] = ms; stats[7] = h.chol_fail;
    }
    if (h.chol_fail) return slam_fail(ctx, SLAM_ERR_NUMERIC, "slam_local_ba: reduced camera system not positive definite (theta and outliers left unchanged)");
    return SLAM_OK;
}


} // extern "C"

// Worker threads for the host half of a batch (structure analysis, staging, result scatter of S windows): created once, parked on a
// condition variable between calls -- starting 15 threads per phase cost more than the work they did (128 windows: 2.3 ms of a 6.3 ms
// call).  Callers from several contexts take turns (run_mu).  Windows are handed out one at a time from an atomic counter.
namespace {
struct BAPool {
    std::vector<std::thread> th;
    std::mutex mu, run_mu;
    std::condition_variable cv, cv_done;
    const std::function<void(int)> *fn = nullptr;
    int n = 0, pending = 0; unsigned long gen = 0; bool stop = false;
    std::atomic<int> next{0};
    explicit BAPool(int workers)
    {
        for (int t = 0; t < workers; t++)
            th.emplace_back([this] {
                unsigned long seen = 0;
                for (;;) {
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; }
                    for (int z; (z = next.fetch_add(1)) < n;) (*fn)(z);
                    { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_one(); }
                }
            });
    }
    ~BAPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &x : th) x.join(); }
    void run(int count, const std::function<void(int)> &f)
    {
        std::lock_guard<std::mutex> turn(run_mu);
        if (th.empty() || count <= 1) { for (int z = 0; z < count; z++) f(z); return; }
        { std::lock_guard<std::mutex> lk(mu); fn = &f; n = count; next.store(0); pending = (int)th.size(); gen++; }
        cv.notify_all();
        for (int z; (z = next.fetch_add(1)) < count;) f(z);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};
std::atomic<long> n_xretry{0};       // calls that were solved again on one workgroup per window (slam_debug_ba_xretries)
BAPool &ba_pool()
{
    static const int env_threads = [] { const char *v = getenv("SLAMHIP_BA_THREADS"); return v ? atoi(v) : 0; }();
    const int hw = (int)std::thread::hardware_concurrency();
    static BAPool pool(std::max(0, (env_threads > 0 ? env_threads : std::min(std::max(hw / 4, 4), 32)) - 1));
    return pool;
}
}  // namespace

extern "C" {

// bundle_adjustment! for S windows at once (no reference counterpart, like the other *_batch entry points; the caller is the estimator
// task of S lock-stepped SlamManagers, estimator.jl:78-99 / :317-347): every kernel of slam_local_ba with the window on blockIdx.y, each
// window with its own device-side LM state; host set-up (structure analysis, staging) spread over threads; ONE host -> device copy, one
// memset, 5 launches per LM iteration for the whole batch, one device -> host copy.  Window z's results equal slam_local_ba's on its
// arrays.  Windows the banded group kernels do not cover (no banded pose order, a point with > 448 observations, no observations) are
// solved one by one through slam_local_ba afterwards.
int slam_local_ba_batch(slam_ctx *ctx, int S, const double *cams, const int32_t *Pn, const int32_t *Mn, const int32_t *On,
                        double *theta, const uint8_t *theta_const, const double *pixels_yx,
                        const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                        int iters_fast, int iterations, double repr_eps, double *stats, int32_t *status)
{
    ARG_TRY(ctx, ctx != nullptr && S >= 1 && S <= 65535 && cams != nullptr && Pn != nullptr && Mn != nullptr && On != nullptr);
    ARG_TRY(ctx, theta != nullptr && theta_const != nullptr && outliers != nullptr && iters_fast >= 0 && iterations >= 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    static const bool host_times = getenv("SLAMHIP_BA_HOSTTIME") != nullptr;
    const auto tw0 = std::chrono::steady_clock::now();
    std::vector<size_t> th_off(S + 1, 0), pc_off(S + 1, 0), ob_off(S + 1, 0);
    for (int z = 0; z < S; z++) {
        ARG_TRY(ctx, Pn[z] > 0 && Mn[z] >= 0 && On[z] >= 0);
        th_off[z + 1] = th_off[z] + 6 * (size_t)Pn[z] + 3 * (size_t)Mn[z]; pc_off[z + 1] = pc_off[z] + Pn[z]; ob_off[z + 1] = ob_off[z] + On[z];
    }
    ARG_TRY(ctx, ob_off[S] == 0 || (pixels_yx != nullptr && pose_ids != nullptr && point_ids != nullptr));
    std::vector<BAPlan> pl(S);
    for (int z = 0; z < S; z++) {
        BAPlan &q = pl[z];
        q.fx = cams[4 * z]; q.fy = cams[4 * z + 1]; q.cx = cams[4 * z + 2]; q.cy = cams[4 * z + 3];
        q.P = Pn[z]; q.M = Mn[z]; q.O = On[z]; q.theta = theta + th_off[z]; q.theta_const_in = theta_const + pc_off[z];
        q.pixels_yx = pixels_yx ? pixels_yx + 2 * ob_off[z] : nullptr; q.pose_ids = pose_ids ? pose_ids + ob_off[z] : nullptr; q.point_ids = point_ids ? point_ids + ob_off[z] : nullptr;
        q.may_reorder = true; q.small_groups = true;
    }
    BAPool &pool = ba_pool();
    const int nthr = (int)pool.th.size() + 1;
    auto parallel = [&](const std::function<void(int)> &fn) { pool.run(S, fn); };
    parallel([&](int z) { ba_plan(pl[z]); });
    const auto tw1 = std::chrono::steady_clock::now();
    std::vector<int> st_code(S, SLAM_OK);
    std::vector<int> batch;                                    // windows the batch kernels take
    std::vector<int> single;                                   // windows solved one by one afterwards
    for (int z = 0; z < S; z++) {
        BAPlan &q = pl[z];
        if (q.err) { st_code[z] = q.err; if (!status) return slam_fail(ctx, q.err, "slam_local_ba_batch: window %d: %s", z, q.msg); continue; }
        const slam_ba *b = q.ba;
        const int Ps = b->pspan > 0 ? b->pspan : q.P, hbq = std::min(std::max(b->hb, 1), Ps - 1);
        if (b->grouped && hbq >= 1 && hbq <= BS_MAXHB && band_lds_bytes(6 * q.P, Ps, hbq) <= 150 * 1024) batch.push_back(z); else single.push_back(z);
    }
    const int NB = (int)batch.size();
    float dev_ms = 0;
    if (NB > 0) {
        // region-major arena: [window table | result table | uploads of every window][zero regions][work regions][results]
        std::vector<size_t> up(NB + 1), ze(NB + 1), wk(NB + 1), rs(NB + 1);
        const size_t tab_bytes = al((size_t)NB * sizeof(BAWin)), rtab_bytes = al((size_t)NB * sizeof(BARes)) + al((size_t)NB * 4);   // (+ the list of k_ba_window's windows)
        up[0] = tab_bytes + rtab_bytes; ze[0] = 0; wk[0] = 0; rs[0] = 0;
        for (int k = 0; k < NB; k++) {
            const BAPlan &q = pl[batch[k]];
            up[k + 1] = up[k] + q.up_bytes; ze[k + 1] = ze[k] + q.zero_bytes; wk[k + 1] = wk[k] + q.work_bytes;
            rs[k + 1] = rs[k] + al(sizeof(LMState)) + al((6 * (size_t)q.P + 3 * (size_t)q.M) * 8 + 8) + al((size_t)q.O + 8);
        }
        const size_t up_total = up[NB], zero_base = up_total, work_base = zero_base + ze[NB], res_base = work_base + wk[NB], total = res_base + rs[NB];
        char *A = nullptr, *stage = nullptr;
        int rc = slam_scratch(ctx, total, (void **)&A);
        if (rc) return rc;
        rc = slam_pinned(ctx, up_total + rs[NB], (void **)&stage);
        if (rc) return rc;
        char *res_host = stage + up_total;
        BAWin *tab_h = (BAWin *)stage; BARes *rtab_h = (BARes *)(stage + tab_bytes);
        parallel([&](int zz) {
            if (zz >= NB) return;
            const int k = zz; BAPlan &q = pl[batch[k]];
            if (ba_emit(q, A + up[k], A + zero_base + ze[k], A + work_base + wk[k], stage + up[k])) return;
            slam_ba *b = q.ba; b->device = ctx->device; b->owns_arena = false; b->arena = A;
            BAWin &w = tab_h[k];
            memset(&w, 0, sizeof w);
            w.d = b->d;
            const int n = w.d.n, Ps = b->pspan > 0 ? b->pspan : q.P, p0 = b->pspan > 0 ? b->p0 : 0, hb = std::min(std::max(b->hb, 1), Ps - 1);
            const double *red = b->reduce;
            w.B.S = red + (size_t)6 * p0 * (n + 1); w.B.g = red + (size_t)n * n + 6 * p0; w.B.ud = red + (size_t)n * n + n + 6 * p0; w.B.Lg = b->band;
            w.B.nb = Ps; w.B.hb = hb; w.B.p0 = p0; w.B.inv_delta_host = 0.0; w.B.fail = b->chol_flag; w.B.trace = nullptr;
            w.B.lds_bytes = (int)band_lds_bytes(n, Ps, hb); w.B.xchg = b->xchg; w.B.epoch = 1; w.B.shift = 0;
            w.nb_obs = b->nblocks_obs; w.nb_pts = b->nblocks_pts; w.n_red = (w.d.P * (w.d.whb + 1) * 36 + w.d.P * 12 + 255) / 256;
            w.ksplit = q.ksplit; w.bwx = q.window ? (double *)(A + zero_base + ze[k] + q.o_bwx) : nullptr;
            if (q.window && q.M > 1) {
                // the split point of k_ba_window's two workgroups: an observation costs the evaluation phases ~24 cycles, an observation of a FREE
                // pose ~6.5 times that in the Schur phase (phase clocks, BW_TRACE) -- and those sit at one end of the sorted points: split by cost
                const int *pfs = (const int *)(stage + up[k] + q.o_pfs);
                const long total = 2L * q.O + 13L * pfs[q.M];
                int kk = 0;
                while (kk < q.M && 2 * (2L * q.start[kk] + 13L * pfs[kk]) < total) kk++;
                w.ksplit = std::min(std::max(kk, 1), q.M - 1);
            }
            BARes &r = rtab_h[k];
            r.off_state = res_base + rs[k]; r.off_theta = r.off_state + al(sizeof(LMState)); r.off_outl = r.off_theta + al((6 * (size_t)q.P + 3 * (size_t)q.M) * 8 + 8);
        });
        for (int k = 0; k < NB; k++) {
            BAPlan &q = pl[batch[k]];
            if (q.err) { st_code[batch[k]] = q.err; if (!status) return slam_fail(ctx, q.err, "slam_local_ba_batch: window %d: %s", batch[k], q.msg); }
        }
        // a window whose set-up failed in ba_emit (a point observed twice by one pose) stays in the table as an inert entry: no groups, no blocks
        // windows one workgroup can keep to itself (k_ba_window): <= 5 free poses, consecutive; the others take the launch-per-phase kernels
        static const bool no_bw = getenv("SLAMHIP_NO_BA_WINDOW") != nullptr;
        std::vector<int> small_list;
        size_t lds_bw = 0;
        for (int k = 0; k < NB && !no_bw; k++) {
            BAPlan &q = pl[batch[k]]; BAWin &w = tab_h[k];
            if (q.err) continue;
            if (q.window) {
                small_list.push_back(k); lds_bw = std::max(lds_bw, bw_lds_bytes(q.P));
                w.pad = 1;
            }
        }
        // (the list travels behind the result table in the same upload)
        const int NS_ = (int)small_list.size();
        bool all_small = NS_ == NB;
        for (int k = 0; k < NB; k++) if (!pl[batch[k]].err && tab_h[k].pad != 1) all_small = false;
        int *list_h = (int *)(stage + tab_bytes + rtab_bytes - al((size_t)NB * 4));
        for (int k = 0; k < NS_; k++) list_h[k] = small_list[k];
        int gx_obs = 1, gx_grp = 1, gx_red = 1, max_ob = 0, max_hb = 0; size_t lds_sg = 0, lds_band = 0;
        for (int k = 0; k < NB; k++) {
            BAPlan &q = pl[batch[k]]; BAWin &w = tab_h[k];
            if (q.err) { w.d.ngrp = 0; w.nb_obs = 0; w.n_red = 0; w.d.O = 0; w.d.M = 0; w.d.n = 0; w.B.nb = 0; continue; }
            if (w.pad) continue;
            gx_obs = std::max(gx_obs, w.nb_obs); gx_grp = std::max(gx_grp, w.d.ngrp); gx_red = std::max(gx_red, w.n_red);
            max_ob = std::max(max_ob, w.d.sg_ob); max_hb = std::max(max_hb, w.d.whb);
            lds_band = std::max(lds_band, (size_t)w.B.lds_bytes);
        }
        // small groups everywhere (the reference's window shape: 16 points x 10 observers): 256-thread workgroups, two to three per compute unit
        static const bool no_t256 = getenv("SLAMHIP_BA_BATCH_T512") != nullptr;
        const int TT = (!no_t256 && max_ob <= 256 && (max_hb + 1) * (max_hb + 2) / 2 <= 256) ? 256 : SG_T;
        for (int k = 0; k < NB; k++) if (!pl[batch[k]].err && !tab_h[k].pad) lds_sg = std::max(lds_sg, sg_lds_bytes(tab_h[k].d.whb, tab_h[k].d.P, tab_h[k].d.sg_ob, tab_h[k].d.sg_sb, TT, tab_h[k].d.sg_hp));
        // the Schur products on the matrix cores (k_schur_groups_m): 256-thread groups whose matrix Y (3 x points columns, 6 x window slots rows) fits
        // LDS beside two more workgroups; SLAMHIP_BA_NO_MFMA=1 keeps the vector kernel (A/B timing, and the parity reference of the tests)
        int ug_n = 6, ug_ob = 8, ug_sb = 8;                     // k_update_groups_b's LDS arrays at the batch's own sizes
        for (int k = 0; k < NB; k++) if (!pl[batch[k]].err && !tab_h[k].pad) { ug_n = std::max(ug_n, tab_h[k].d.n); ug_ob = std::max(ug_ob, tab_h[k].d.sg_ob); ug_sb = std::max(ug_sb, tab_h[k].d.sg_sb); }
        const size_t lds_ug = ug_lds_bytes(ug_n, ug_ob, ug_sb);
        static const bool no_mfma = getenv("SLAMHIP_BA_NO_MFMA") != nullptr;
        size_t lds_m = 0;
        for (int k = 0; k < NB; k++) if (!pl[batch[k]].err && !tab_h[k].pad) lds_m = std::max(lds_m, sgm_lds_bytes(tab_h[k].d.whb, tab_h[k].d.P, tab_h[k].d.sg_ob, tab_h[k].d.sg_sb, tab_h[k].d.sg_hp));
        const bool use_mfma = !no_mfma && TT == 256 && lds_m > 0 && lds_m <= 64 * 1024;
        const auto tw2 = std::chrono::steady_clock::now();
        hipStream_t st = ctx->stream;
        static std::atomic<bool> attr_set[64];
        const int dv = ctx->device & 63;
        if (!attr_set[dv].load(std::memory_order_acquire)) {
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_schur_groups_b<SG_T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg_lds_bytes(BS_MAXHB, SOLVE_MAX_N / 6)));
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_schur_groups_b<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg_lds_bytes(BS_MAXHB, SOLVE_MAX_N / 6)));
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_band_solve_b, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_schur_groups_m<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_update_groups_b<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_update_groups_b<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_update_groups_b<SG_T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            attr_set[dv].store(true, std::memory_order_release);
        }
        const BAWin *tab = (const BAWin *)A; const BARes *rtab = (const BARes *)(A + tab_bytes);
        auto run_pass = [&](int ignore, int iters) {
            hipLaunchKernelGGL(k_linearize_b, dim3(gx_obs, NB), dim3(256), 0, st, tab, ignore, 0);
            hipLaunchKernelGGL(k_pass_start_b, dim3(1, NB), dim3(256), 0, st, tab, ignore ? 1 : 0);
            for (int it = 1; it <= iters; it++) {
                if (use_mfma) hipLaunchKernelGGL(k_schur_groups_m<256>, dim3(gx_grp, NB), dim3(256), lds_m, st, tab, ignore);
                else if (TT == 256) hipLaunchKernelGGL(k_schur_groups_b<256>, dim3(gx_grp, NB), dim3(256), lds_sg, st, tab, ignore);
                else hipLaunchKernelGGL(k_schur_groups_b<SG_T>, dim3(gx_grp, NB), dim3(SG_T), lds_sg, st, tab, ignore);
                hipLaunchKernelGGL(k_schur_reduce_b, dim3(gx_red, NB), dim3(256), 0, st, tab);
                hipLaunchKernelGGL(k_band_solve_b, dim3(1, NB), dim3(BS_T), lds_band, st, tab);
                hipLaunchKernelGGL(k_trial_poses_b, dim3(1, NB), dim3(64), 0, st, tab);
                if (use_mfma) hipLaunchKernelGGL((k_update_groups_b<256, true>), dim3(gx_grp, NB), dim3(256), lds_ug, st, tab, ignore, ug_n, ug_ob, ug_sb);
                else if (TT == 256) hipLaunchKernelGGL((k_update_groups_b<256, false>), dim3(gx_grp, NB), dim3(256), lds_ug, st, tab, ignore, ug_n, ug_ob, ug_sb);
                else hipLaunchKernelGGL((k_update_groups_b<SG_T, false>), dim3(gx_grp, NB), dim3(SG_T), lds_ug, st, tab, ignore, ug_n, ug_ob, ug_sb);
                hipLaunchKernelGGL(k_control_b, dim3(1, NB), dim3(256), 0, st, tab);
            }
        };
        if (NS_ > 0) {                                             // (the flag is set on success only: a failed attribute call is tried again by the next call)
            static std::atomic<bool> bw_attr[64];
            if (!bw_attr[dv].load(std::memory_order_acquire)) {
                HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_ba_window, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bw_lds_bytes(BW_PMAX)));
                bw_attr[dv].store(true, std::memory_order_release);
            }
        }
        // two workgroups per window (k_ba_window) only while BOTH halves of EVERY window are resident at once -- they wait for each other: one
        // workgroup per compute unit (136-148 KB of LDS), so 2 x NS workgroups must fit the device's compute units (hipDeviceProp_t, not a
        // constant), the stream must not be CU-masked (a mask says nothing about how many of an XCD's units are left, and the halves b / b + 8
        // need two on the SAME XCD) and the architecture must be the one the memory-side hand-over was validated on (ctx->xwg_ok).  Other
        // processes' kernels can still hold LDS the count knows nothing about: the kernel's wait is bounded (xlimit) and a window whose halves
        // missed each other comes back with chol_fail = 2 -- the call is then solved again with one workgroup per window.
        // SLAMHIP_BA_WINDOW_ONE=1: always one; SLAMHIP_BA_XWAIT_US: the bound (default 500 000 us; 0 in the tests = give up at once).
        static const bool bw_one = getenv("SLAMHIP_BA_WINDOW_ONE") != nullptr;
        static const long long xlimit = [] { const char *v = getenv("SLAMHIP_BA_XWAIT_US"); return (v ? atoll(v) : 500000LL) * 100; }();
        const int two_grid = 16 * ((NS_ + 7) / 8);
        int two = (!bw_one && ctx->xwg_ok && two_grid <= ctx->dev_cus) ? 1 : 0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        for (int attempt = 0; e == hipSuccess && attempt < 2; attempt++) {
            e = hipMemcpyAsync(A, stage, up_total, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemsetAsync(A + zero_base, 0, ze[NB], st);
            if (e != hipSuccess) break;
            (void)hipEventRecord(e0, st);
            if (NS_ > 0) {
                const int *list_d = (const int *)(A + tab_bytes + rtab_bytes - al((size_t)NB * 4));
                hipLaunchKernelGGL(k_ba_window, dim3(two ? two_grid : NS_), dim3(BW_T), lds_bw, st, tab, list_d, NS_, two, iters_fast, iterations, repr_eps, 1e-6, xlimit);
            }
            if (!all_small) {
                run_pass(0, iters_fast);
                hipLaunchKernelGGL(k_outliers_b, dim3(gx_obs, NB), dim3(256), 0, st, tab, repr_eps, 1e-6);
                hipLaunchKernelGGL(k_outlier_count_b, dim3(1, NB), dim3(256), 0, st, tab);
                run_pass(1, iterations);
            }
            hipLaunchKernelGGL(k_results_b, dim3(8, NB), dim3(256), 0, st, tab, rtab, A);
            e = hipGetLastError();
            (void)hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipMemcpyAsync(res_host, A + res_base, rs[NB], hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = slam_stream_wait(st);
            if (e == hipSuccess) (void)hipEventElapsedTime(&dev_ms, e0, e1);
            if (e != hipSuccess || !two) break;
            bool missed = false;                                   // did the halves of some window miss each other?
            for (int k : small_list) if (((const LMState *)(res_host + (rtab_h[k].off_state - res_base)))->chol_fail == 2) { missed = true; break; }
            if (!missed) break;
            two = 0; n_xretry.fetch_add(1);
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (e != hipSuccess) return slam_fail(ctx, SLAM_ERR_HIP, "slam_local_ba_batch: %s", hipGetErrorString(e));
        const auto tw3 = std::chrono::steady_clock::now();
        // results -> the caller's arrays (its pose order, its observation order); a failed factorisation leaves a window's arrays untouched
        parallel([&](int zz) {
            if (zz >= NB) return;
            const int k = zz, z = batch[k]; BAPlan &q = pl[z];
            if (q.err) return;
            const BARes &r = rtab_h[k];
            const LMState &h = *(const LMState *)(res_host + (r.off_state - res_base));
            if (stats) {
                double *sv = stats + 8 * (size_t)z;
                sv[0] = h.ssr_init; sv[1] = h.ssr_pass1; sv[2] = h.ssr_final; sv[3] = h.iters_pass1; sv[4] = h.iters_pass2; sv[5] = h.n_outliers; sv[6] = dev_ms; sv[7] = h.chol_fail;
            }
            if (h.chol_fail) { st_code[z] = SLAM_ERR_NUMERIC; return; }
            const double *th = (const double *)(res_host + (r.off_theta - res_base));
            double *dst = theta + th_off[z];
            const int n = 6 * q.P;
            if (q.ba->pose_order.empty()) memcpy(dst, th, (size_t)n * 8);
            else for (int p = 0; p < q.P; p++) memcpy(dst + 6 * q.ba->pose_order[p], th + 6 * p, 48);
            memcpy(dst + n, th + n, (size_t)3 * q.M * 8);
            const uint8_t *ol = (const uint8_t *)(res_host + (r.off_outl - res_base));
            uint8_t *od = outliers + ob_off[z];
            for (int s2 = 0; s2 < q.O; s2++) od[q.ba->perm[s2]] = ol[s2];
        });
        if (host_times) {
            const auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
            fprintf(stderr, "slam_local_ba_batch host: %d windows (%d threads): plan %ld us, emit %ld us, enqueue + wait %ld us (device %.0f us), scatter %ld us; %zu B up, %zu B arena\n",
                    NB, nthr, us(tw0, tw1), us(tw1, tw2), us(tw2, tw3), dev_ms * 1e3, us(tw3, std::chrono::steady_clock::now()), up_total, total);
        }
    }
    for (int z : single) {
        if (st_code[z]) continue;
        const BAPlan &q = pl[z];
        double sv[8] = {0};
        const int rc1 = slam_local_ba(ctx, q.fx, q.fy, q.cx, q.cy, q.P, q.M, q.O, theta + th_off[z], theta_const + pc_off[z], q.pixels_yx, q.pose_ids, q.point_ids,
                                      outliers + ob_off[z], iters_fast, iterations, repr_eps, sv);
        if (stats) memcpy(stats + 8 * (size_t)z, sv, sizeof sv);
        st_code[z] = rc1;
        if (rc1 && rc1 != SLAM_ERR_NUMERIC && !status) return rc1;
    }
    int first = SLAM_OK;
    for (int z = 0; z < S; z++) { if (status) status[z] = st_code[z]; if (st_code[z] && !first) first = st_code[z]; }
    if (status) return SLAM_OK;                                // per-window codes are in status[]
    if (first == SLAM_ERR_NUMERIC) return slam_fail(ctx, SLAM_ERR_NUMERIC, "slam_local_ba_batch: a reduced camera system was not positive definite (that window's theta and outliers are left unchanged)");
    return first;
}


// slam_local_ba_batch in two halves: _begin hands the whole call (structure analysis, staging, upload, solve, download, scatter) to a thread of the
// library's own and returns; _end waits for it and returns its code.  The estimator task of a host (estimator.jl:78-99) thereby prepares key-frame
// k + 1's windows -- or does anything else -- while key-frame k's are planned and solved, without a thread of its own.  Between the two calls the
// context belongs to the job (one outstanding job per context; the arrays passed to _begin are read AND written by the job: they stay valid and
// untouched until _end returns).  A host that wants several batches in flight uses several contexts.
struct BABatchJob { std::thread th; int rc = SLAM_OK; bool active = false; };
static std::mutex g_job_mu;
static std::vector<std::pair<slam_ctx *, BABatchJob *>> g_jobs;
static BABatchJob *job_of(slam_ctx *ctx, bool create)
{
    std::lock_guard<std::mutex> lk(g_job_mu);
    for (auto &e : g_jobs) if (e.first == ctx) return e.second;
    if (!create) return nullptr;
    g_jobs.emplace_back(ctx, new BABatchJob());
    return g_jobs.back().second;
}
int slam_local_ba_batch_begin(slam_ctx *ctx, int S, const double *cams, const int32_t *Pn, const int32_t *Mn, const int32_t *On,
                              double *theta, const uint8_t *theta_const, const double *pixels_yx,
                              const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                              int iters_fast, int iterations, double repr_eps, double *stats, int32_t *status)
{
    ARG_TRY(ctx, ctx != nullptr);
    BABatchJob *j = job_of(ctx, true);
    if (j->active) return slam_fail(ctx, SLAM_ERR_ARG, "slam_local_ba_batch_begin: the context already has a batch in flight (call slam_local_ba_batch_end first)");
    j->active = true; j->rc = SLAM_OK;
    j->th = std::thread([=] { j->rc = slam_local_ba_batch(ctx, S, cams, Pn, Mn, On, theta, theta_const, pixels_yx, pose_ids, point_ids, outliers, iters_fast, iterations, repr_eps, stats, status); });
    return SLAM_OK;
}
int slam_local_ba_batch_end(slam_ctx *ctx)
{
    ARG_TRY(ctx, ctx != nullptr);
    BABatchJob *j = job_of(ctx, false);
    if (!j || !j->active) return slam_fail(ctx, SLAM_ERR_ARG, "slam_local_ba_batch_end: no batch in flight on this context");
    j->th.join(); j->active = false;
    return j->rc;                                              // (the message of a failure is the context's: slam_last_error)
}
// a context that goes away takes its job record along (called by slam_ctx_destroy; a job still in flight is waited for)
extern "C" void ba_forget_jobs(slam_ctx *ctx)
{
    BABatchJob *j = nullptr;
    {   std::lock_guard<std::mutex> lk(g_job_mu);
        for (size_t i = 0; i < g_jobs.size(); i++) if (g_jobs[i].first == ctx) { j = g_jobs[i].second; g_jobs.erase(g_jobs.begin() + (long)i); break; } }
    if (j) { if (j->active) j->th.join(); delete j; }
}

// how many slam_local_ba_batch calls of this process had to be solved again because the two workgroups of a window missed each other
long slam_debug_ba_xretries(void) { return n_xretry.load(); }

// host-only timing of the batch set-up (no HIP call, no device needed): plan + emit of S windows on `threads` threads (0: the library's parked
// worker pool, as slam_local_ba_batch uses it) into malloc'ed staging; out_us = {plan, emit}; returns the number of windows whose set-up failed.  Measurement aid for tuning the host side on any machine (scripts/probes/ba_host_time.py).
int slam_debug_ba_host_time(int S, const double *cams, const int32_t *Pn, const int32_t *Mn, const int32_t *On, const double *theta, const uint8_t *theta_const,
                            const double *pixels_yx, const int64_t *pose_ids, const int64_t *point_ids, int threads, double *out_us)
{
    std::vector<size_t> th_off(S + 1, 0), pc_off(S + 1, 0), ob_off(S + 1, 0);
    for (int z = 0; z < S; z++) { th_off[z + 1] = th_off[z] + 6 * (size_t)Pn[z] + 3 * (size_t)Mn[z]; pc_off[z + 1] = pc_off[z] + Pn[z]; ob_off[z + 1] = ob_off[z] + On[z]; }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<BAPlan> pl(S);
    for (int z = 0; z < S; z++) {
        BAPlan &q = pl[z];
        q.fx = cams[4 * z]; q.fy = cams[4 * z + 1]; q.cx = cams[4 * z + 2]; q.cy = cams[4 * z + 3];
        q.P = Pn[z]; q.M = Mn[z]; q.O = On[z]; q.theta = theta + th_off[z]; q.theta_const_in = theta_const + pc_off[z];
        q.pixels_yx = pixels_yx + 2 * ob_off[z]; q.pose_ids = pose_ids + ob_off[z]; q.point_ids = point_ids + ob_off[z];
        q.may_reorder = true; q.small_groups = true;
    }
    auto parallel = [&](auto fn) {
        if (threads == 0) { ba_pool().run(S, fn); return; }      // the parked worker pool of slam_local_ba_batch itself (callers from several threads take turns)
        if (threads <= 1) { for (int z = 0; z < S; z++) fn(z); return; }
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back([&, t] { for (int z = t; z < S; z += threads) fn(z); });
        for (int z = 0; z < S; z += threads) fn(z);
        for (auto &x : th) x.join();
    };
    parallel([&](int z) { ba_plan(pl[z]); });
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<size_t> up(S + 1, 0);
    for (int z = 0; z < S; z++) up[z + 1] = up[z] + pl[z].up_bytes;
    std::vector<char> stage(up[S] + 64);
    char *fake = (char *)(uintptr_t)0x100000000ull;
    parallel([&](int z) { if (!pl[z].err) ba_emit(pl[z], fake, fake, fake, stage.data() + up[z]); });
    const auto t2 = std::chrono::steady_clock::now();
    out_us[0] = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() * 1e-3;
    out_us[1] = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count() * 1e-3;
    int bad = 0;
    for (int z = 0; z < S; z++) bad += pl[z].err != 0;
    return bad;
}

} // extern "C"

// ---------------------------------------------------------------------------------
// pnp_bundle_adjustment (bundle_adjustment.jl:113-171): one pose, n points.  The
// whole two-pass LM (dense 6x6 normal equations, exact Cholesky step) runs inside
// ONE kernel / one workgroup: per iteration two block reductions and a
// single-thread 6x6 solve; no host round trips.

#define PNP_T 256
// (templates + forced inlining keep the partial sums in registers and the LDS / global pointers in their address spaces: the
//  generic version ran with 188 bytes of scratch per lane and FLAT accesses throughout)
typedef const __attribute__((address_space(1))) double *pnp_gcd;
typedef __attribute__((address_space(1))) uint8_t *pnp_gu8;
template <int cnt>
__device__ __forceinline__ void pnp_reduce(double *v, double *sh /* 4*cnt */, double *outv)
{
#pragma unroll
    for (int k = 0; k < cnt; k++) {
        double t = v[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
        v[k] = t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < cnt; k++) sh[(threadIdx.x >> 6) * cnt + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int k = 0; k < cnt; k++) { double t = 0.0; for (int w = 0; w < PNP_T / 64; w++) t += sh[w * cnt + k]; outv[k] = t; }
    __syncthreads();
}

__device__ __forceinline__ int pnp_lm(const PnPArgs &A, pnp_gcd gpx, pnp_gcd gpts, pnp_gu8 goutl, double *X /*shared 6*/, int ignore, int iterations, double *sh, double *red /*shared 40*/,
                      double *Xt /*shared 6*/, double *dxs /*shared 6*/, int *flags /*shared 4*/, double *ssr_out)
{
    const int tid = threadIdx.x, n = A.n;
    double v[28];
    // ssr at X
    v[0] = 0.0;
    for (int i = tid; i < n; i += PNP_T) {
        if (ignore && goutl[i]) continue;
        double r[2];
        obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, nullptr, nullptr, nullptr);
        v[0] += r[0] * r[0] + r[1] * r[1];
    }
    pnp_reduce<1>(v, sh, red);
    double ssr = red[0], delta = LM_DELTA0, decrease = 2.0;
    __shared__ double Hs[36], gs[6];
    int need_jac = 1, converged = 0, iter = 0;
    while (!converged && iter < iterations) {
        iter++;
        if (need_jac) {
#pragma unroll
            for (int k = 0; k < 27; k++) v[k] = 0.0;
            for (int i = tid; i < n; i += PNP_T) {
                if (ignore && goutl[i]) continue;
                double r[2], Jp[12], Jl[6];
                obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, Jp, Jl, nullptr);
                int c = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++) v[c++] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
#pragma unroll
                for (int a = 0; a < 6; a++) v[21 + a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
            }
            pnp_reduce<27>(v, sh, red);
            if (tid == 0) {
                int c = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++) { Hs[a + 6 * b] = red[c]; Hs[b + 6 * a] = red[c]; c++; }
#pragma unroll
                for (int a = 0; a < 6; a++) gs[a] = red[21 + a];
            }
            need_jac = 0;
            __syncthreads();
        }
        if (tid == 0) {
            // 6 x 6 damped normal equations, Cholesky + two triangular solves on one lane; every loop has constant bounds and is
            // unrolled, so H and x live in registers (with run-time bounds they sat in scratch: ~100 dependent scratch round trips
            // per LM iteration)
            double H[36], x[6];
#pragma unroll
            for (int k = 0; k < 36; k++) H[k] = Hs[k];
#pragma unroll
            for (int a = 0; a < 6; a++) { H[a + 6 * a] += fmin(fmax(Hs[a + 6 * a], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * (1 / delta); x[a] = gs[a]; }
            int fail = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                double dj = H[j + 6 * j];
#pragma unroll
                for (int k = 0; k < j; k++) dj -= H[j + 6 * k] * H[j + 6 * k];
                if (!(dj > 0)) fail = 1;
                dj = fail ? 1.0 : sqrt(dj); H[j + 6 * j] = dj;           // (after a failure the remaining values are not used)
#pragma unroll
                for (int i = j + 1; i < 6; i++) {
                    double sv = H[i + 6 * j];
#pragma unroll
                    for (int k = 0; k < j; k++) sv -= H[i + 6 * k] * H[j + 6 * k];
                    H[i + 6 * j] = sv / dj;
                }
            }
            if (!fail) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    double sv = x[i];
#pragma unroll
                    for (int k = 0; k < i; k++) sv -= H[i + 6 * k] * x[k];
                    x[i] = sv / H[i + 6 * i];
                }
#pragma unroll
                for (int i = 5; i >= 0; i--) {
                    double sv = x[i];
#pragma unroll
                    for (int k = i + 1; k < 6; k++) sv -= H[k + 6 * i] * x[k];
                    x[i] = sv / H[i + 6 * i];
                }
            }
#pragma unroll
            for (int a = 0; a < 6; a++) { dxs[a] = fail ? 0.0 : x[a]; Xt[a] = X[a] - dxs[a]; }
            flags[0] = fail;
        }
        __syncthreads();
        if (flags[0]) break;
        v[0] = 0.0; v[1] = 0.0;
        for (int i = tid; i < n; i += PNP_T) {
            if (ignore && goutl[i]) continue;   // zero residual and zero Jacobian row: contributes 0 to both sums
            double r[2], rt[2], Jp[12], Jl[6];
            obs_eval(Xt, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, rt, nullptr, nullptr, nullptr);
            obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, Jp, Jl, nullptr);
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) { a += Jp[k] * dxs[k]; b += Jp[6 + k] * dxs[k]; }
            a -= r[0]; b -= r[1];
            v[0] += rt[0] * rt[0] + rt[1] * rt[1];
            v[1] += a * a + b * b;
        }
        pnp_reduce<2>(v, sh, red);
        const double trial = red[0], pred = red[1];
        double mx = 0.0;
        for (int a = 0; a < 6; a++) mx = fmax(mx, fabs(dxs[a]));
        const double rho = (trial - ssr) / (pred - ssr);
        if (rho > LM_MIN_STEP_QUALITY) {
            const int x_conv = mx <= LM_XTOL;
            const int f_conv = fabs(ssr - trial) / (fabs(ssr) + LM_FTOL) <= LM_FTOL;
            ssr = trial;
            const double u = 2.0 * rho - 1.0;
            delta = fmin(delta / fmax(1.0 / 3.0, 1.0 - u * u * u), LM_MAX_DELTA);
            decrease = 2.0; need_jac = 1;
            converged = x_conv || f_conv;
            __syncthreads();
            if (tid < 6) X[tid] = Xt[tid];
        } else {
            delta = fmax(delta / decrease, LM_MIN_DELTA);
            decrease *= 2.0;
            converged = mx <= LM_XTOL;
        }
        __syncthreads();
    }
    *ssr_out = ssr;
    return iter;
}

__device__ __forceinline__ void pnp_body(const PnPArgs &A)
{
    const pnp_gcd gpx = (pnp_gcd)A.px, gpts = (pnp_gcd)A.pts; const pnp_gu8 goutl = (pnp_gu8)A.outl;
    __shared__ double X[6], Xt[6], dxs[6], sh[4 * 28], red[40];
    __shared__ int flags[4];
    const int tid = threadIdx.x, n = A.n;
    if (tid < 6) X[tid] = A.X0[tid];
    for (int i = tid; i < n; i += PNP_T) goutl[i] = 0;
    __syncthreads();
    double v[1] = {0.0};
    for (int i = tid; i < n; i += PNP_T) {
        double r[2];
        obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, nullptr, nullptr, nullptr);
        v[0] += r[0] * r[0] + r[1] * r[1];
    }
    pnp_reduce<1>(v, sh, red);
    const double err_init = red[0];
    double ssr1 = 0.0, ssr2 = 0.0;
    const int it1 = pnp_lm(A, gpx, gpts, goutl, X, 0, A.iters_fast, sh, red, Xt, dxs, flags, &ssr1);
    v[0] = 0.0;
    for (int i = tid; i < n; i += PNP_T) {
        double r[2], z;
        obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, nullptr, nullptr, &z);
        const bool o = z < A.depth_eps || (r[0] * r[0] + r[1] * r[1]) > A.repr_eps;
        goutl[i] = o ? 1 : 0;
        v[0] += o ? 1.0 : 0.0;
    }
    __threadfence_block();
    pnp_reduce<1>(v, sh, red);
    const int no = (int)red[0];
    int identity = 0, it2 = 0;
    if (n - no < 5) { identity = 1; ssr2 = ssr1; }
    else it2 = pnp_lm(A, gpx, gpts, goutl, X, 1, A.iterations, sh, red, Xt, dxs, flags, &ssr2);
    __syncthreads();
    if (tid == 0) {
        for (int a = 0; a < 6; a++) A.result[a] = X[a];
        A.result[6] = err_init; A.result[7] = ssr2; A.result[8] = no; A.result[9] = identity; A.result[10] = it1; A.result[11] = it2;
    }
}

__global__ __launch_bounds__(PNP_T) void k_pnp(PnPArgs A) { pnp_body(A); }
// S independent problems, one workgroup each (slam_pnp_ba_batch); the argument blocks live in mapped host memory
__global__ __launch_bounds__(PNP_T) void k_pnp_batch(const PnPArgs *args)
{
    pnp_body(args[blockIdx.x]);                                  // read in place (wave-uniform scalar loads): no LDS copy behind a generic reference
}

int pnp_launch_device(slam_ctx *ctx, int S, const PnPArgs *args_dev)
{
    ProfScope span(ctx, "pnp_ba");
    hipLaunchKernelGGL(k_pnp_batch, dim3(S), dim3(PNP_T), 0, ctx->stream, args_dev);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}

// RotZYX(pose[1:3,1:3]) -> angles (Rotations.jl), pose column-major: R[i][j] = pose[i + 4j]
static void pnp_pose_to_x(const double *pose_cw, double *X0)
{
    const double R11 = pose_cw[0], R21 = pose_cw[1], R31 = pose_cw[2], R12 = pose_cw[4], R22 = pose_cw[5], R13 = pose_cw[8], R23 = pose_cw[9];
    const double t1 = std::atan2(R21, R11), s1 = std::sin(t1), c1 = std::cos(t1);
    X0[0] = t1; X0[1] = std::atan2(-R31, std::sqrt(R11 * R11 + R21 * R21)); X0[2] = std::atan2(R13 * s1 - R23 * c1, R22 * c1 - R12 * s1);
    X0[3] = pose_cw[12]; X0[4] = pose_cw[13]; X0[5] = pose_cw[14];
}
static void pnp_x_to_pose(const double *res, double *out_pose)
{
    for (int k = 0; k < 16; k++) out_pose[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (res[9] == 0.0) {
        const double s1 = std::sin(res[0]), c1 = std::cos(res[0]), s2 = std::sin(res[1]), c2 = std::cos(res[1]), s3 = std::sin(res[2]), c3 = std::cos(res[2]);
        const double R[9] = {c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3, s1 * c2, s1 * s2 * s3 + c1 * c3, s1 * s2 * c3 - c1 * s3, -s2, c2 * s3, c2 * c3};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out_pose[i + 4 * j] = R[3 * i + j];
        out_pose[12] = res[3]; out_pose[13] = res[4]; out_pose[14] = res[5];
    }
}

// S single-pose refinements in one launch: problem z owns points [offsets[z], offsets[z+1]); cams S x 4 (fx, fy,
// cx, cy), poses_cw / out_poses S x 16 column-major
extern "C" int slam_pnp_ba_batch(slam_ctx *ctx, int S, const int32_t *offsets, const double *cams, const double *poses_cw,
                                 const double *pixels_yx, const double *points_xyz, int iters_fast, int iterations,
                                 double depth_eps, double repr_eps, double *out_poses, double *err_init, double *err_final,
                                 uint8_t *outliers, int *n_outliers)
{
    ARG_TRY(ctx, ctx != nullptr && S >= 0);
    if (S == 0) return SLAM_OK;
    ARG_TRY(ctx, offsets && cams && poses_cw && out_poses && offsets[0] == 0);
    for (int z = 0; z < S; z++) ARG_TRY(ctx, offsets[z + 1] >= offsets[z]);
    const int ntot = offsets[S];
    ARG_TRY(ctx, ntot == 0 || (pixels_yx && points_xyz && outliers));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pxb = al((size_t)ntot * 16 + 8), ptb = al((size_t)ntot * 24 + 8), ob = al((size_t)ntot + 8), rb = al((size_t)S * 128);
    char *s;
    int rc = slam_scratch(ctx, pxb + ptb + ob + rb, (void **)&s);
    if (rc) return rc;
    double *d_px = (double *)s, *d_pts = (double *)(s + pxb); uint8_t *d_o = (uint8_t *)(s + pxb + ptb); double *d_res = (double *)(s + pxb + ptb + ob);
    char *h, *d;
    rc = slam_pinned(ctx, (size_t)S * sizeof(PnPArgs) + (size_t)S * 128, (void **)&h);
    if (rc) return rc;
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&d, h, 0));
    PnPArgs *args = (PnPArgs *)h;
    for (int z = 0; z < S; z++) {
        PnPArgs &A = args[z];
        A.cam = {cams[4 * z], cams[4 * z + 1], cams[4 * z + 2], cams[4 * z + 3]};
        A.n = offsets[z + 1] - offsets[z]; A.iters_fast = iters_fast; A.iterations = iterations;
        A.depth_eps = depth_eps; A.repr_eps = repr_eps;
        pnp_pose_to_x(poses_cw + 16 * z, A.X0);
        A.px = d_px + 2 * (size_t)offsets[z]; A.pts = d_pts + 3 * (size_t)offsets[z]; A.outl = d_o + offsets[z]; A.result = d_res + 16 * z;
    }
    if (ntot > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_px, pixels_yx, (size_t)ntot * 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_pts, points_xyz, (size_t)ntot * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    { ProfScope span(ctx, "pnp_ba");
      hipLaunchKernelGGL(k_pnp_batch, dim3(S), dim3(PNP_T), 0, ctx->stream, (const PnPArgs *)d); }
    HIP_TRY(ctx, hipGetLastError());
    double *res = (double *)(h + (size_t)S * sizeof(PnPArgs));
    HIP_TRY(ctx, hipMemcpyAsync(res, d_res, (size_t)S * 128, hipMemcpyDeviceToHost, ctx->stream));
    if (ntot > 0) HIP_TRY(ctx, hipMemcpyAsync(outliers, d_o, (size_t)ntot, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    for (int z = 0; z < S; z++) {
        const double *r = res + 16 * z;
        if (err_init) err_init[z] = r[6];
        if (err_final) err_final[z] = r[7];
        if (n_outliers) n_outliers[z] = (int)r[8];
        pnp_x_to_pose(r, out_poses + 16 * z);
    }
    return SLAM_OK;
}

extern "C" int slam_pnp_ba(slam_ctx *ctx, double fx, double fy, double cx, double cy,
                           const double pose_cw[16], const double *pixels_yx, const double *points_xyz, int n,
                           int iters_fast, int iterations, double depth_eps, double repr_eps,
                           double out_pose[16], double *err_init, double *err_final, uint8_t *outliers, int *n_outliers)
{
    ARG_TRY(ctx, ctx != nullptr && pose_cw != nullptr && out_pose != nullptr && n >= 0);
    ARG_TRY(ctx, n == 0 || (pixels_yx != nullptr && points_xyz != nullptr && outliers != nullptr));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PnPArgs A;
    A.cam = {fx, fy, cx, cy}; A.n = n; A.iters_fast = iters_fast; A.iterations = iterations;
    A.depth_eps = depth_eps; A.repr_eps = repr_eps;
    pnp_pose_to_x(pose_cw, A.X0);
    const size_t pxb = al((size_t)n * 16 + 8), ptb = al((size_t)n * 24 + 8), ob = al((size_t)n + 8);
    char *s;
    int rc = slam_scratch(ctx, pxb + ptb + ob + 256, (void **)&s);
    if (rc) return rc;
    double *d_px = (double *)s, *d_pts = (double *)(s + pxb); uint8_t *d_o = (uint8_t *)(s + pxb + ptb); double *d_res = (double *)(s + pxb + ptb + ob);
    if (n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_px, pixels_yx, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_pts, points_xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    A.px = d_px; A.pts = d_pts; A.outl = d_o; A.result = d_res;
    hipLaunchKernelGGL(k_pnp, dim3(1), dim3(PNP_T), 0, ctx->stream, A);
    HIP_TRY(ctx, hipGetLastError());
    double res[12];
    HIP_TRY(ctx, hipMemcpyAsync(res, d_res, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
    if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(outliers, d_o, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    if (err_init) *err_init = res[6];
    if (err_final) *err_final = res[7];
    if (n_outliers) *n_outliers = (int)res[8];
    pnp_x_to_pose(res, out_pose);
    return SLAM_OK;
}
