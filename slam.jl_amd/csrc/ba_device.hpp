// ba_device.hpp -- what the four translation units of the local bundle adjustment share: the device-side state (LMState, BADev, BAWin), the
// residual / Jacobian evaluation, the bodies of the grouped build, the banded and dense solves, the update and the LM control (every kernel of
// ba_single.hip and ba_batch.hip is a thin wrapper around one of them), their LDS-size helpers, and the constants of k_ba_window that the host
// planner needs.  (Round 6: ba.hip, 4 900 lines, split into ba_single.hip / ba_batch.hip / ba_window.hip / ba_host.hip + this header.)
#pragma once
#include "common.hpp"
#include <atomic>
#include <algorithm>
#include <type_traits>
#include <chrono>
#include <thread>
#include <functional>
#include <memory>
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


// BADevT<PlainP> = BADev: what the host fills and every kernel receives.  BADevT<GlobP> = BADevG: the same bytes with every pointer typed as GLOBAL memory -- a
// kernel that reads the structure out of the window table (batches) sees generic pointers otherwise, and every access through one is a FLAT instruction, which
// also counts against the LDS counter (an LDS wait then waits for the outstanding global loads).
template <class T> struct PlainP { typedef T *type; };
template <class T> struct GlobP { typedef __attribute__((address_space(1))) T *type; };
template <template <class> class Q> struct BADevT {
#define BP(T) typename Q<T>::type
    Cam cam;
    int P, M, O, n;              // n = 6P
    BP(double) pose, pose_t, pts, pts_t;
    BP(const uint8_t) pconst;
    BP(const double) pix;         // SoA: py[O], px[O]
    BP(const int) opose, opoint, pt_start;
    BP(uint8_t) outl, hasp;
    BP(double) f, ft;            // AoS O x 2
    BP(double) Jp, Jl;           // AoS O x 12, O x 6 (a lane reads its observation's block contiguously)
    BP(double) Vinv, bl;         // SoA 6 x M, 3 x M
    BP(double) T, Wm;            // AoS O x 18 each
    BP(const int2) pairs; BP(const int) blk_start; BP(const int2) blk_pq; int nblk;
    // pt_start / pt_id: observations are sorted by map point, the map points by (first free observing pose, id); opk = the
    // sorted position of an observation's point.  grp / fgrp / wpart: the point groups of k_schur_groups (below).
    BP(const int) pt_id, opk;
    BP(const int4) grp; BP(const int) fgrp; int ngrp, whb, wstride;
    BP(const int) fobs;           // the observations of free poses, grouped by map point in sorted order (pfs[M] entries)
    BP(const int) pfs;            // pfs[k]: observations of free poses of the map points (sorted order) before point k, M + 1 entries (k_ba_window's chunks)
    BP(const int) ohp; int sg_hp;   // ohp[i]: index of observation i among its group's observations of FREE poses (or -1): the phase 2-3 records (W, Jp, gradient:
                                 // 36 doubles) exist for those only -- the reference's window is 80 % observations of constant poses; sg_hp: room for that many
    int sg_ob, sg_sb;            // k_schur_groups' LDS layout: room for sg_ob observations / sg_sb points per group (SG_OB / SG_SB; a batch of small
                                 // windows sizes it to its largest group, so that several workgroups share a compute unit)
    BP(double) wpart;
    BP(double) S, g, udiag;     // reduce buffer views
    BP(double) Swork, dp, dl;
    BP(double) sc0, sc1;         // [P][6] each: sin / cos of the angles of the poses in d.pose / in d.pose_t (batches: formed once per window and iteration by
                                 // k_pass_start_b / k_trial_poses_b and swapped with the parameter buffers; every point group used to form them for itself)
    BP(double) part;              // reduction partials
    BP(LMState) st;
#undef BP
};
typedef BADevT<PlainP> BADev;
typedef BADevT<GlobP> BADevG;
static_assert(sizeof(BADev) == sizeof(BADevG), "BADevG re-types BADev's pointers, nothing else");
// the committed parameters and the trial ones: d.pose / d.pts hold the committed set while st->cur == 0, d.pose_t / d.pts_t while it is 1
template <template <class> class Q> struct ParamBufsT { typename Q<double>::type pose, pts, pose_t, pts_t, sc, sc_t; };
typedef ParamBufsT<PlainP> ParamBufs; typedef ParamBufsT<GlobP> ParamBufsG;     // sc / sc_t: sin / cos of the committed / trial poses (batches)
__device__ __forceinline__ ParamBufs param_bufs(const BADev &d)
{
    const bool sw = d.st->cur != 0;
    return ParamBufs{sw ? d.pose_t : d.pose, sw ? d.pts_t : d.pts, sw ? d.pose : d.pose_t, sw ? d.pts : d.pts_t, sw ? d.sc1 : d.sc0, sw ? d.sc0 : d.sc1};
}
__device__ __forceinline__ ParamBufsG param_bufs(const BADevG &d)
{
    const bool sw = d.st->cur != 0;
    return ParamBufsG{sw ? d.pose_t : d.pose, sw ? d.pts_t : d.pts, sw ? d.pose : d.pose_t, sw ? d.pts : d.pts_t, sw ? d.sc1 : d.sc0, sw ? d.sc0 : d.sc1};
}
// a group record through a pointer typed as global memory (the vector type's copy constructor wants a generic reference: component by component)
__device__ __forceinline__ int4 ld_grp(__attribute__((address_space(1))) const int4 *g, int k)
{
    __attribute__((address_space(1))) const int *q = (__attribute__((address_space(1))) const int *)(g + k);
    return make_int4(q[0], q[1], q[2], q[3]);
}
// the same window with its pointers typed as global memory (see BADevT)
__device__ __forceinline__ BADevG ba_global(const BADev &d) { BADevG g; __builtin_memcpy(&g, &d, sizeof g); return g; }


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
// (the same through pointers typed as global memory: BADevG)
typedef double sg_d2 __attribute__((ext_vector_type(2)));
template <int N> __device__ __forceinline__ void ld_rec(__attribute__((address_space(1))) const double *p, double *v)
{
    static_assert(N % 2 == 0, "even record length");
    __attribute__((address_space(1))) const sg_d2 *q = (__attribute__((address_space(1))) const sg_d2 *)p;
#pragma unroll
    for (int k = 0; k < N / 2; k++) { const sg_d2 t = q[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
}
template <int N> __device__ __forceinline__ void st_rec(__attribute__((address_space(1))) double *p, const double *v)
{
    static_assert(N % 2 == 0, "even record length");
    __attribute__((address_space(1))) sg_d2 *q = (__attribute__((address_space(1))) sg_d2 *)p;
#pragma unroll
    for (int k = 0; k < N / 2; k++) { sg_d2 t; t.x = v[2 * k]; t.y = v[2 * k + 1]; q[k] = t; }
}

template <bool STORE = true>          // STORE = false (batches): the cost only -- the grouped build evaluates every observation again and keeps what it needs
__device__ __forceinline__ void linearize_body(const BADev &d0, int ignore_outliers, int respect_done)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store, not flat accesses, where the caller read the window out of the batch's table)
    const ParamBufsG pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
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

__device__ __forceinline__ void inv3_sym(const double V[6], double I[6])
{
    const double a = V[0], b = V[1], c = V[2], dd = V[3], e = V[4], f = V[5];
    const double A = dd * f - e * e, B = c * e - b * f, C = b * e - c * dd;
    const double det = a * A + b * B + c * C, id = 1.0 / det;
    I[0] = A * id; I[1] = B * id; I[2] = C * id;
    I[3] = (a * f - c * c) * id; I[4] = (b * c - a * e) * id; I[5] = (a * dd - b * b) * id;
}


// per observation: W = Jp'Jl (6x3), T = W V^-1 of its point

// One wave per non-zero upper block (p <= q) of the reduced camera system.

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
#define SG_CLK_DECL long long sg_clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const long long sg_t0 = clock64()
#define SG_CLK(k) sg_clk[k] = clock64() - sg_t0
#define SG_NOW() clock64()
#define SG_ADD(k, t) sg_clk[k] += clock64() - (t)
#define SG_DUMP() do { if (threadIdx.x == 0 && blockIdx.x == (gridDim.y > 1 ? 10 : 100) && blockIdx.y == (gridDim.y > 1 ? 5 : 0) && d.st->iters == 3) printf("schur group: npts %d nobs %d | init %lld ph0 %lld bar %lld ph1 %lld ph2 %lld ph3 %lld bar %lld fold %lld tail %lld cycles\n", npts, nobs, sg_clk[0], sg_clk[1] - sg_clk[0], sg_clk[2] - sg_clk[1], sg_clk[3] - sg_clk[2], sg_clk[4] - sg_clk[3], sg_clk[8] - sg_clk[4], sg_clk[9] - sg_clk[8], sg_clk[5] - sg_clk[9], sg_clk[6] - sg_clk[5]); } while (0)
#define SGM_DUMP() do { if (threadIdx.x == 0 && blockIdx.x == 10 && blockIdx.y == 5 && d.st->iters == 3) printf("schur group (mfma): npts %d nobs %d | init %lld ph0 %lld bar %lld ph1 %lld ph2a %lld ph2x %lld ph2b %lld mfma+out %lld (products %lld, output %lld) cycles\n", npts, nobs, sg_clk[0], sg_clk[1] - sg_clk[0], sg_clk[2] - sg_clk[1], sg_clk[3] - sg_clk[2], sg_clk[4] - sg_clk[3], sg_clk[5] - sg_clk[4], sg_clk[6] - sg_clk[5], sg_clk[7] - sg_clk[6], sg_clk[8], sg_clk[9]); } while (0)
#else
#define SG_CLK_DECL
#define SG_CLK(k)
#define SG_NOW() 0
#define SG_ADD(k, t) (void)(t)
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
__device__ __forceinline__ void schur_groups_body(const BADev &d0, double inv_delta_host, int ignore_outliers, int use_state)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store, not flat accesses, where the caller read the window out of the batch's table)
    const ParamBufsG pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    extern __shared__ __attribute__((aligned(16))) double sg_lds[];
    SG_CLK_DECL;
    if (use_state && d.st->converged) return;
    const int tid = threadIdx.x, M = d.M, O = d.O;
    const int4 G = ld_grp(d.grp, blockIdx.x);               // first point, first observation, f | points << 16, observations
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
    for (int p = tid; p < d.P; p += TT) { const double ang[3] = {pb.pose[6 * p], pb.pose[6 * p + 1], pb.pose[6 * p + 2]}; pose_sincos(ang, s_sc + 6 * p); }
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
    __attribute__((address_space(1))) double *out = d.wpart + (size_t)blockIdx.x * d.wstride;
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
    return (sgm_r_doubles(whb, ob, sb, hp) + (size_t)sb * 16 + 8 + (size_t)(whb + 1) * 36 + (size_t)P * 6 + (size_t)sgm_rp(whb)) * 8 + (size_t)sb * (whb + 1) * 2 + 16;
}
template <int TT>
__device__ __forceinline__ void schur_groups_mfma_body(const BADev &d0, int ignore_outliers)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store, not flat accesses, where the caller read the window out of the batch's table)
    extern __shared__ __attribute__((aligned(16))) double sg_lds[];
    SG_CLK_DECL;
    if (d.st->converged) return;
    const ParamBufsG pb = param_bufs(d);
    // (the window's arrays through pointers typed as global memory: a pointer that a batch kernel reads out of the window table is a generic pointer to the
    //  compiler, every access a FLAT instruction -- which also counts against the LDS counter, so that an LDS wait waits for the outstanding global loads)
#define SGM_G(T, name, src) __attribute__((address_space(1))) T *name = (__attribute__((address_space(1))) T *)(src)
    SGM_G(const int, g_opose, d.opose); SGM_G(const int, g_opoint, d.opoint); SGM_G(const int, g_opk, d.opk); SGM_G(const int, g_ohp, d.ohp);
    SGM_G(const uint8_t, g_outl, d.outl); SGM_G(const uint8_t, g_pconst, d.pconst); SGM_G(const double, g_pix, d.pix);
    SGM_G(const int, g_pt_id, d.pt_id); SGM_G(const int, g_pt_start, d.pt_start); SGM_G(const int, g_grp, d.grp); SGM_G(const LMState, g_st, d.st);
    SGM_G(const double, g_sc, pb.sc); SGM_G(const double, g_pts, pb.pts); SGM_G(const double, g_pose, pb.pose);
    SGM_G(double, g_Vinv, d.Vinv); SGM_G(double, g_bl, d.bl);
#undef SGM_G
    const int tid = threadIdx.x, M = d.M, O = d.O, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (the wave index in a scalar register: the tile loop branches on it)
    constexpr int NW = TT / 64;
    const int4 G = make_int4(g_grp[4 * blockIdx.x], g_grp[4 * blockIdx.x + 1], g_grp[4 * blockIdx.x + 2], g_grp[4 * blockIdx.x + 3]);                       // first point, first observation, f | points << 16, observations
    const int k0 = __builtin_amdgcn_readfirstlane(G.x), o0 = __builtin_amdgcn_readfirstlane(G.y), f = __builtin_amdgcn_readfirstlane(G.z & 0xffff),
              npts = __builtin_amdgcn_readfirstlane(G.z >> 16), nobs = __builtin_amdgcn_readfirstlane(G.w);      // (uniform by construction; told to the compiler: scalar loop counters)
    const int hbw = d.whb + 1, nwin = hbw * (hbw + 1) / 2, RP = sgm_rp(d.whb), nrow = 6 * hbw;
    const int OBc = d.sg_ob, SBc = d.sg_sb, HPc = d.sg_hp;
    double *s_R = sg_lds;
    double *s_pt = s_R + sgm_r_doubles(d.whb, OBc, SBc, HPc);   // [SBc][16]  V^-1 (6), bl (3), L (6: l00 l10 l20 l11 l21 l22)
    double *s_dg = s_pt + SBc * 16 + 8;                          // [hbw][36]  Jp'Jp per window slot
    double *s_sc = s_dg + hbw * 36;                              // [P][6]     sin / cos of every pose's angles
    unsigned *s_tr = (unsigned *)(s_sc + 6 * d.P), *s_tc = s_tr + RP;   // [RP] each: where row / column i of Y Y' goes in the group's output (phase 3)
    short *s_slot = (short *)(s_tr + 2 * RP);                    // [SBc][hbw] record of point x in window slot y, or -1
    // the observation's scalars are requested BEFORE the set-up work below (their latency hides behind the sin / cos of the poses)
    int i = 0, p = 0, j = 0, pl = 0, hpi = -1; bool active = false, hp = false; double py = 0.0, px = 0.0;
    if (tid < nobs) {
        i = o0 + tid; p = g_opose[i]; j = g_opoint[i]; pl = g_opk[i] - k0; hpi = g_ohp[i];
        active = !(ignore_outliers && g_outl[i]); hp = active && !g_pconst[p];
        py = g_pix[i]; px = g_pix[O + i];
    }
    // ... and so is what phase 1's lanes (one per map point) read from global memory: behind the barriers below these loads were a memory round trip of their own
    int p1_jj = 0, p1_t0 = 0, p1_t1 = 0;
    if (tid < npts) { p1_jj = g_pt_id[k0 + tid]; p1_t0 = g_pt_start[k0 + tid] - o0; p1_t1 = g_pt_start[k0 + tid + 1] - o0; }
    const double inv_delta = 1.0 / g_st->delta;
    for (int a = tid; a < 6 * d.P; a += TT) s_sc[a] = g_sc[a];      // (formed once per window: k_pass_start_b / k_trial_poses_b)
    for (int x = tid; x < npts * hbw; x += TT) s_slot[x] = -1;
    // element (row, col) of Y Y' is entry (rr, cc) of window block (a, b) = (row / 6, col / 6), stored at (a hbw - a (a - 1) / 2 + b - a) 36 + rr 6 + cc:
    // slot << 26 | the row's / the column's share of that offset, once per group instead of once per accumulator element
    for (int x = tid; x < RP; x += TT) {
        const int a = x / 6, rr = x - 6 * a;
        s_tr[x] = a < hbw ? ((unsigned)a << 26) | (unsigned)((a * hbw - a * (a - 1) / 2 - a) * 36 + rr * 6) : 0xfc000000u;      // (a row past the window: slot 63 is above every column's)
        s_tc[x] = a < hbw ? ((unsigned)a << 26) | (unsigned)(a * 36 + rr) : 0u;                                                  // (a column past the window: slot 0 with the offset bits 0 -- told apart from block 0's column 0 by x)
    }
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
            const double X[3] = {g_pts[3 * j], g_pts[3 * j + 1], g_pts[3 * j + 2]};
            double sc[6], tr[3];
#pragma unroll
            for (int k = 0; k < 6; k++) sc[k] = s_sc[6 * p + k];
#pragma unroll
            for (int k = 0; k < 3; k++) tr[k] = g_pose[6 * p + 3 + k];
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
        const int jj = p1_jj, t0 = p1_t0, t1 = p1_t1;
        double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
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
        for (int c = 0; c < 6; c++) { g_Vinv[(size_t)c * M + jj] = Vi[c]; s_pt[tid * 16 + c] = Vi[c]; }
#pragma unroll
        for (int c = 0; c < 3; c++) { g_bl[(size_t)c * M + jj] = V[6 + c]; s_pt[tid * 16 + 6 + c] = V[6 + c]; }
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
    __attribute__((address_space(1))) double *out = (__attribute__((address_space(1))) double *)d.wpart + (size_t)blockIdx.x * d.wstride;
    {
        const int TPW = (nrow + NW - 1) / NW, xs = TPW <= 16 ? 2 : TPW <= 32 ? 1 : 0, XQ = 1 << xs;      // (a shift, not a run-time division)
        const int task = wv * TPW + (lane >> xs), xq = lane & (XQ - 1);
        const bool xl = (lane >> xs) < TPW && task < nrow;
        const int a2 = xl ? task / 6 : 0, rr = task - 6 * (task / 6);
        double ex[7];
#pragma unroll
        for (int k = 0; k < 7; k++) ex[k] = 0.0;
        if (xl)
            for (int x = xq; x < npts; x += XQ) {
                const int ta = s_slot[x * hbw + a2];
                if (ta < 0) continue;
                double J[12];
                const double *rec = s_R + ta * 18;
                ld_rec<12>(rec, J);
                const double j0 = rec[rr], j1 = rec[6 + rr];      // (two more LDS reads instead of two five-deep select chains over J)
#pragma unroll
                for (int c = 0; c < 6; c++) ex[c] = fma(j1, J[6 + c], fma(j0, J[c], ex[c]));
                ex[6] += rec[12 + rr];
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
    // the padding rows nrow .. RP (fewer than 16: a 16-lane group per column) and the padding columns K3 .. K4 (fewer than 4), without a run-time division
    if ((tid & 15) < RP - nrow)
        for (int k = tid >> 4; k < K4; k += TT / 16) s_R[((size_t)(k >> 1) * RP + nrow + (tid & 15)) * 2 + (k & 1)] = 0.0;
    for (int k = K3; k < K4; k++)
        for (int row = tid; row < nrow; row += TT) s_R[((size_t)(k >> 1) * RP + row) * 2 + (k & 1)] = 0.0;
    lds_sync();
    SG_CLK(6);
    // ---- phase 3: -(Y Y') on the matrix cores: the upper-triangular 16 x 16 tiles dealt to the waves, three tiles (three independent
    //      accumulator chains) at a time
    {
        const int NT = RP / 16, ntiles = NT * (NT + 1) / 2, q = lane >> 4, c16 = lane & 15, nks = K4 >> 2;
        const size_t lane_off = ((size_t)(q >> 1) * RP + c16) * 2 + (q & 1), kstep = (size_t)4 * RP;
        auto emit = [&](int I, int J, const sgm_d4 &acc) {
            const int col = J * 16 + c16;
            const unsigned tc = s_tc[col];
            const int b = (int)(tc >> 26), oc = (int)(tc & 0x3ffffffu);
            if (col >= nrow) return;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned tr = s_tr[I * 16 + 4 * r + q];
                const int a = (int)(tr >> 26);
                if (a > b) continue;                                 // (below the block diagonal, or a row past the window)
                const int o = (int)(tr & 0x3ffffffu) + oc;           // (w 36 + rr 6) + (b 36 + cc) - ... see the table: the row's share carries -a 36
                if (a < b) out[o] = -acc[r];
                else {
                    const int rr6 = (int)(tr & 0x3ffffffu) - (a * hbw - a * (a - 1) / 2 - a) * 36, cc = oc - b * 36;
                    const double v = s_dg[a * 36 + rr6 + cc] - acc[r];
                    out[o] = v;
                    if (I < J) out[o - rr6 - cc + cc * 6 + rr6 / 6] = v;      // a diagonal block cut by a tile boundary: its mirror half lies in a tile below the diagonal, which nobody computes
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
            const long long sg_tm = SG_NOW();
            // (an instruction occupies the matrix pipe for 64 cycles whether or not it depends on the one before: a wave's last round of tiles -- one or two
            //  of them -- must not run dummy chains beside them)
            if (ok[2])
                for (int ks = 0; ks < nks; ks++) {
                    const size_t o = (size_t)ks * kstep;
                    const double a0 = pa0[o], b0 = pb0[o], a1 = pa1[o], b1 = pb1[o], a2m = pa2[o], b2 = pb2[o];
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2m, b2, c2, 0, 0, 0);
                }
            else if (ok[1])
                for (int ks = 0; ks < nks; ks++) {
                    const size_t o = (size_t)ks * kstep;
                    const double a0 = pa0[o], b0 = pb0[o], a1 = pa1[o], b1 = pb1[o];
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
                }
            else
                for (int ks = 0; ks < nks; ks++) {
                    const size_t o = (size_t)ks * kstep;
                    const double a0 = pa0[o], b0 = pb0[o];
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
                }
            SG_ADD(8, sg_tm);
            const long long sg_te = SG_NOW();
            emit(I[0], J[0], c0);
            if (ok[1]) emit(I[1], J[1], c1);
            if (ok[2]) emit(I[2], J[2], c2);
            SG_ADD(9, sg_te);
        }
    }
    SG_CLK(7);
    SGM_DUMP();
}

// S, g, diag(U) from the window partials: thread = (band block (p, p + dq), entry) / (pose, gradient or diagonal entry)
__device__ __forceinline__ void schur_reduce_body(const BADev &d0, int use_state)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store instead of flat accesses where the caller read the window out of the batch's table)
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

// ---- damped solve of the reduced camera system ----------------------------------
// Tiled right-looking Cholesky over 32x32 tiles, one launch per tile column, every
// tile of the trailing matrix on its own workgroup.  The right-hand side rides
// along as row n of the (n+1) x n working matrix, so the forward substitution
// L y = g falls out of the factorisation (y' = last row of L); a single blocked
// back-substitution kernel finishes L' dp = y.
#define CT 32
struct CholArgs { double *A; double *Lf; int n, ld; int *fail; };   // A: working matrix (updated in place); Lf: finished factor tiles

// copy S -> work (lower triangle + rhs row), add the LM damping to the diagonal

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



// L' dp = y (y = row n of the factor), blocked from the last tile column upwards;
// the diagonal solves are mat-vecs with the stored tile inverses.

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
template <template <class> class Q> struct BandArgsT { typename Q<const double>::type S, g, ud; typename Q<double>::type Lg; int nb, hb; double inv_delta_host; typename Q<int>::type fail; typename Q<long long>::type trace; int lds_bytes; typename Q<double>::type xchg; int epoch; int shift; int p0; };
typedef BandArgsT<PlainP> BandArgs; typedef BandArgsT<GlobP> BandArgsG;
static_assert(sizeof(BandArgs) == sizeof(BandArgsG), "BandArgsG re-types BandArgs' pointers, nothing else");
__device__ __forceinline__ BandArgsG band_global(const BandArgs &b) { BandArgsG g; __builtin_memcpy(&g, &b, sizeof g); return g; }
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
__device__ __forceinline__ void band_solve_body(const BADev &d0, const BandArgs &B0, int use_state)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store, not flat accesses, where the caller read the window out of the batch's table)
    const BandArgsG B = band_global(B0);
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
    __attribute__((address_space(1))) double *const Lg = B.Lg + (size_t)side * nbT * lgs;
    __attribute__((address_space(1))) double *const dpo = d.dp + 6 * B.p0;                   // dp of the solve's first pose (the span of the free poses)
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
            __attribute__((address_space(1))) const double *src = e[q].x >= 0 ? B.S + (e[q].y + di * strideS) : B.g + (e[q].y + di * strideG);
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
        auto Lgk = Lg + (size_t)k * lgs;
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
        __attribute__((address_space(1))) const double *Lgk = B.Lg + (size_t)k * lgs;
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
__device__ __forceinline__ void update_groups_body(const BADev &d0, int ignore_outliers, int use_state, double *s_dp, double *s_u, double *s_dl, double *s_red, double *s_sct, bool sct_ready, double *s_sc = nullptr)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store instead of flat accesses where the caller read the window out of the batch's table)
    const ParamBufsG pb = param_bufs(d);                     // committed / trial parameters (LMState::cur)
    if (use_state && d.st->converged) return;
    const int tid = threadIdx.x, M = d.M, O = d.O, n = d.n;
    const int4 G = ld_grp(d.grp, blockIdx.x);
    const int k0 = G.x, o0 = G.y, npts = G.z >> 16, nobs = G.w;
    // everything a lane reads from global memory at an address it already knows is requested HERE, in front of the barriers: behind them the observation's
    // scalars and the point's index / range were memory round trips of their own (~1 us each on a loaded device), three of them per group
    const int i = o0 + tid;
    int p = 0, pl = 0, jo = 0; bool active = false; double py = 0.0, px = 0.0;
    if (tid < nobs) {
        p = d.opose[i]; pl = d.opk[i] - k0; jo = d.opoint[i];
        active = !(ignore_outliers && d.outl[i]);
        py = d.pix[i]; px = d.pix[O + i];
    }
    int p2_j = 0, p2_t0 = 0, p2_t1 = 0;
    if (tid < npts) { p2_j = d.pt_id[k0 + tid]; p2_t0 = d.pt_start[k0 + tid] - o0; p2_t1 = d.pt_start[k0 + tid + 1] - o0; }
    for (int a = tid; a < n; a += TT) s_dp[a] = d.dp[a];
    if (sct_ready) { for (int a = tid; a < n; a += TT) s_sct[a] = pb.sc_t[a]; }
    if (RECOMP) { for (int a = tid; a < n; a += TT) s_sc[a] = pb.sc[a]; }
    lds_sync();
    // (the point's own records -- their address needs its index -- travel while the observations are evaluated)
    double p2_bl[3] = {0.0, 0.0, 0.0}, p2_Vi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, p2_X[3] = {0.0, 0.0, 0.0};
    if (tid < npts) {
#pragma unroll
        for (int k = 0; k < 3; k++) { p2_bl[k] = d.bl[(size_t)k * M + p2_j]; p2_X[k] = pb.pts[3 * p2_j + k]; }
#pragma unroll
        for (int k = 0; k < 6; k++) p2_Vi[k] = d.Vinv[(size_t)k * M + p2_j];
    }
    if (!sct_ready)
        for (int q = tid; q < d.P; q += TT) {
            const double tp[3] = {pb.pose[6 * q] - s_dp[6 * q], pb.pose[6 * q + 1] - s_dp[6 * q + 1], pb.pose[6 * q + 2] - s_dp[6 * q + 2]};
            pose_sincos(tp, s_sct + 6 * q);
        }
    double mx = 0.0;
    if (blockIdx.x == 0)
        for (int a = tid; a < n; a += TT) { const double v = s_dp[a]; pb.pose_t[a] = pb.pose[a] - v; mx = fmax(mx, fabs(v)); }
    double jp[12], jl[6], ff[2] = {0.0, 0.0}, a = 0.0, b = 0.0;
    if (tid < nobs) {
        if (RECOMP) {
#pragma unroll
            for (int k = 0; k < 12; k++) jp[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) jl[k] = 0.0;
            if (active) {
                const int j = jo;
                const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                double sc[6], tr[3];
#pragma unroll
                for (int k = 0; k < 6; k++) sc[k] = s_sc[6 * p + k];
#pragma unroll
                for (int k = 0; k < 3; k++) tr[k] = pb.pose[6 * p + 3 + k];
                obs_eval_sc(sc, tr, X, py, px, d.cam, ff, jp, jl, nullptr);
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
        const int j = p2_j;
        double bl[3] = {p2_bl[0], p2_bl[1], p2_bl[2]};
        const int t0 = p2_t0, t1 = p2_t1;
        for (int t = t0; t < t1; t++) {
#pragma unroll
            for (int k = 0; k < 3; k++) bl[k] -= s_u[t * 3 + k];
        }
        const double *Vi = p2_Vi;
        const double l0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
        const double l1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
        const double l2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
        const double X0 = p2_X[0] - l0, X1 = p2_X[1] - l1, X2 = p2_X[2] - l2;
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
            obs_eval_sc(sc, tr, X, py, px, d.cam, r, nullptr, nullptr, nullptr);
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

// Sums the partials (fixed order) and, in the single-GPU path, runs the
// LeastSquaresOptim accept/reject logic.  mode 0: ssr of the current residuals
// (after k_linearize); mode 1: trial/predicted/maxdx -> state (+ LM decision if lm).
// fixed-order strided sum / max of a partials array by one 256-thread workgroup
__device__ __forceinline__ double ctl_sum(__attribute__((address_space(1))) const double *p, int n, int stride, double *sh)
{
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) t += p[(size_t)i * stride];
    return block_sum(t, sh);
}
__device__ __forceinline__ double ctl_max(__attribute__((address_space(1))) const double *p, int n, double *sh)
{
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) t = fmax(t, p[i]);
    return block_max(t, sh);
}
// LeastSquaresOptim's accept / reject of a trial step (trust-region radius update, step-quality test): t = trial cost,
// p = predicted cost, mx = max |dx|
template <class SP>      // LMState * (k_ba_window's copy in LDS) or a pointer typed as global memory
__device__ __forceinline__ void lm_decide(SP s, double t, double p, double mx)
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
__device__ __forceinline__ void control_body(const BADev &d0, int mode, int nb_obs, int nb_pts, int lm, double *out4)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store, not flat accesses, where the caller read the window out of the batch's table)
    __shared__ double sh[4];
    __attribute__((address_space(1))) LMState *s = d.st;
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

// The sharded path: every rank's [trial_ssr, pred_ssr, max|dx|, chol_fail] gathered into g (nranks x 4).  Sums / maxima in
// rank order, then the same decision as the single-GPU path -- identical on every rank, taken on the device.
// start of an LM pass in the sharded path: the all-reduced cost of the current parameters comes from the reduce buffer

// host-paced protocol (slam_ba_commit): the host has decided -- an accepted step swaps the two parameter buffers.  (The device-paced
// paths swap inside lm_decide: no launch at all.)


// _ba_detect_outliers!, bundle_adjustment.jl:90-111
__device__ __forceinline__ void outliers_body(const BADev &d0, double repr_eps, double depth_eps)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store, not flat accesses, where the caller read the window out of the batch's table)
    const ParamBufsG pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
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
__device__ __forceinline__ void outlier_count_body(const BADev &d0, int nb_obs)
{
    const BADevG d = ba_global(d0);                          // (global_load / global_store, not flat accesses, where the caller read the window out of the batch's table)
    __shared__ double sh[4];
    const double t = ctl_sum(d.part, nb_obs, 1, sh);           // counts: exact in any order
    if (threadIdx.x == 0) d.st->n_outliers = (int)t;
}

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
// k_ba_window is defined in ba_window.hip and launched by ba_batch.hip
__global__ void k_ba_window(const BAWin *tab, const int *list, int ns, int two, int iters_fast, int iterations, double repr_eps, double depth_eps, long long xlimit);
#include "ba_host.hpp"
