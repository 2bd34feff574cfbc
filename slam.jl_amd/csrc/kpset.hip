// kpset.hip -- device-resident keypoint lists for S lock-stepped streams.
//
// In the reference the keypoints of a frame live in a Dict (src/frame.jl) and optical_flow_matching!
// (src/map_manager.jl:451-564) copies them into arrays, tracks, and applies updates / removals one by one; the
// round-1 batch seams kept that list on the HOST and crossed PCIe with it at every call.  Here the list stays in HBM:
// tracking (slam_kpset_flow_match), the removal of lost keypoints, map culling (slam_kpset_remove), the avoidance list
// and the merge of fresh keypoints in key-frame detection (slam_kpset_detect), stereo matching
// (slam_kpset_stereo_match) and triangulation (slam_kpset_triangulate) all read and write the same device arrays; the
// host sees one small copy of the per-stream counts per step (slam_kpset_counts).  Compaction is stable (a stream's
// keypoints keep their order) and is done with wave ballots + prefix counts, one workgroup per stream.
#include "common.hpp"
#include <algorithm>
#include "tri_device.hpp"
#include <cmath>

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// live slots of all streams back to back + their number.  One small workgroup per stream (it sums the counts before its own: S
// loads): a single 1024-thread workgroup had to wait for sixteen free wave slots on one CU while the pyramid kernels fill the chip
// (58 us on average in the pipeline for 12 us of work).
__global__ __launch_bounds__(256) void k_kpset_worklist(const int *count, int S, int cap, int *work, int *ntot)
{
    __shared__ int s_off;
    const int s = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {
        int c = 0, t = 0;
        for (int i = tid; i < S; i += 64) { const int v = count[i]; t += v; c += i < s ? v : 0; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { c += __shfl_xor(c, m); t += __shfl_xor(t, m); }
        if (tid == 0) { s_off = c; if (s == 0) ntot[0] = t; }
    }
    __syncthreads();
    const int off = s_off, n = count[s];
    for (int j = tid; j < n; j += 256) work[off + j] = s * cap + j;
}

int kpset_build_worklist(slam_ctx *ctx, slam_kpset *ks)
{
    hipLaunchKernelGGL(k_kpset_worklist, dim3(ks->S), dim3(256), 0, ctx->stream, (const int *)ks->count, ks->S, ks->cap, ks->work, ks->ntot);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}

// Stable in-place compaction of every stream's list, one 256-thread workgroup per stream.
// mode 0 (after a temporal match): keep st != 0; a tracked keypoint (st == 1) takes its new position from oyx.
// mode 1 (removal by flags): keep flags[slot] == 0.
// Chunks of 256 slots are read (all fields into registers), ranked with a wave ballot + the popcount of the lower lanes,
// and written back after a barrier: destinations never lie to the right of their sources, so in place is safe.
struct KpsetView {
    double *yx, *oyx, *syx, *xyz, *kyx, *fyx; int64_t *id; uint8_t *is3d, *stereo, *st, *haskf; int *fkf, *kfcount; int *count; int cap;
};
__global__ __launch_bounds__(256) void k_kpset_compact(KpsetView K, int mode, const uint8_t *flags)
{
    __shared__ int s_w[4], s_base;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t b = (size_t)s * K.cap;
    const int n = K.count[s];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int j = c0 + tid;
        bool keep = false;
        double y = 0, x = 0, sy = 0, sx = 0, X0 = 0, X1 = 0, X2 = 0, ky = 0, kx = 0, fy0 = 0, fx0 = 0; int64_t id = 0; uint8_t f3 = 0, fs = 0, fk = 0; int fkid = 0;
        if (j < n) {
            const size_t q = b + j;
            if (mode == 0) { const uint8_t t = K.st[q]; keep = t != 0; if (t == 1) { y = K.oyx[2 * q]; x = K.oyx[2 * q + 1]; } else { y = K.yx[2 * q]; x = K.yx[2 * q + 1]; } }
            else { keep = flags[q] == 0; y = K.yx[2 * q]; x = K.yx[2 * q + 1]; }
            if (keep) { sy = K.syx[2 * q]; sx = K.syx[2 * q + 1]; X0 = K.xyz[3 * q]; X1 = K.xyz[3 * q + 1]; X2 = K.xyz[3 * q + 2]; id = K.id[q]; f3 = K.is3d[q]; fs = K.stereo[q];
                        ky = K.kyx[2 * q]; kx = K.kyx[2 * q + 1]; fk = K.haskf[q]; fy0 = K.fyx[2 * q]; fx0 = K.fyx[2 * q + 1]; fkid = K.fkf[q]; }
        }
        const unsigned long long m = __ballot(keep);
        const int rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wv] = __builtin_popcountll(m);
        __syncthreads();                                        // all reads of the chunk done; wave totals visible
        int wbase = 0;
        for (int i = 0; i < wv; i++) wbase += s_w[i];
        const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (keep) {
            const size_t q = b + s_base + wbase + rank;
            K.yx[2 * q] = y; K.yx[2 * q + 1] = x; K.syx[2 * q] = sy; K.syx[2 * q + 1] = sx;
            K.xyz[3 * q] = X0; K.xyz[3 * q + 1] = X1; K.xyz[3 * q + 2] = X2; K.id[q] = id; K.is3d[q] = f3; K.stereo[q] = fs;
            K.kyx[2 * q] = ky; K.kyx[2 * q + 1] = kx; K.haskf[q] = fk; K.fyx[2 * q] = fy0; K.fyx[2 * q + 1] = fx0; K.fkf[q] = fkid;
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) K.count[s] = s_base;
}

static KpsetView view_of(slam_kpset *ks)
{
    KpsetView K; K.yx = ks->yx; K.oyx = ks->oyx; K.syx = ks->syx; K.xyz = ks->xyz; K.kyx = ks->kyx; K.id = ks->id; K.is3d = ks->is3d; K.stereo = ks->stereo;
    K.haskf = ks->haskf; K.fyx = ks->fyx; K.fkf = ks->fkf; K.kfcount = ks->kfcount;
    K.st = ks->st; K.count = ks->count; K.cap = ks->cap;
    return K;
}

int kpset_stage_params(slam_ctx *ctx, slam_kpset *ks, const double *host, size_t n, const double **dev_out)
{
    const size_t slot_d = (size_t)ks->S * 32;
    ARG_TRY(ctx, n <= slot_d);
    const int sl = ks->par_slot; ks->par_slot = (sl + 1) & 7;
    HIP_TRY(ctx, hipEventSynchronize(ks->par_ev[sl]));           // the copy that last used this slot (8 calls ago) has long completed
    double *h = ks->par_host + (size_t)sl * slot_d, *dv = ks->par + (size_t)sl * slot_d;
    memcpy(h, host, n * 8);
    HIP_TRY(ctx, hipMemcpyAsync(dv, h, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ks->par_ev[sl], ctx->stream));
    *dev_out = dv;
    return SLAM_OK;
}

int kpset_compact(slam_ctx *ctx, slam_kpset *ks, int mode, const uint8_t *flags_dev)
{
    hipLaunchKernelGGL(k_kpset_compact, dim3(ks->S), dim3(256), 0, ctx->stream, view_of(ks), mode, flags_dev);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}

// Array-level body of triangulate_stereo! (src/mapper.jl:142-183) on the set: every 2-D keypoint with a stereo match is
// triangulated (the DLT of slam_triangulate, same gates); success -> map point Twc[s] * X, is3d = 1; failure -> the stereo
// observation is dropped (remove_stereo_keypoint!).
struct KTriArgs {
    double P1[16], P2[16], T21[16], cam1[4], cam2[4];
    const double *Twc;                 // [S][16] column-major camera-1 -> world, device
    double max_error, min_depth;
};
__device__ __forceinline__ void k_kpset_triangulate_slot(const KpsetView &K, const KTriArgs &T, const int *work, int i)
{
    const size_t q = (size_t)work[i];
    if (!K.stereo[q] || K.is3d[q]) return;
    const int s = (int)(q / K.cap);
    const double x1 = K.yx[2 * q + 1], y1 = K.yx[2 * q], x2 = K.syx[2 * q + 1], y2 = K.syx[2 * q];
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
    bool ok = !(L2 < T.min_depth);
    double R[3];
    for (int r = 0; r < 3; r++) R[r] = ((T.T21[r] * L0 + T.T21[r + 4] * L1) + T.T21[r + 8] * L2) + T.T21[r + 12] * L3;
    if (ok && R[2] < T.min_depth) ok = false;
    if (ok) {
        const double iz = 1.0 / L2;
        const double py = T.cam1[1] * L1 * iz + T.cam1[3], px = T.cam1[0] * L0 * iz + T.cam1[2];
        const double dy = y1 - py, dx = x1 - px;
        if (sqrt(dy * dy + dx * dx) > T.max_error) ok = false;
    }
    if (ok) {
        const double iz = 1.0 / R[2];
        const double py = T.cam2[1] * R[1] * iz + T.cam2[3], px = T.cam2[0] * R[0] * iz + T.cam2[2];
        const double dy = y2 - py, dx = x2 - px;
        if (sqrt(dy * dy + dx * dx) > T.max_error) ok = false;
    }
    if (ok) {
        const double *W = T.Twc + 16 * (size_t)s;                // project_camera_to_world (frame.jl): Twc * X
        for (int r = 0; r < 3; r++) K.xyz[3 * q + r] = ((W[r] * L0 + W[r + 4] * L1) + W[r + 8] * L2) + W[r + 12] * L3;
        K.is3d[q] = 1;
    } else K.stereo[q] = 0;
}
// (the grid is sized from the host's bound of the list lengths, which is only a hint: the loop covers every live slot whatever it was)
__global__ __launch_bounds__(64) void k_kpset_triangulate(KpsetView K, KTriArgs T, const int *work, const int *ntot)
{
    const int n = ntot[0];
    for (int i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) k_kpset_triangulate_slot(K, T, work, i);
}

// Array-level body of triangulate_temporal! (src/mapper.jl:185-262) on the set: every 2-D keypoint whose first observer (the
// key-frame that detected it, observers[1] of its map point) is an earlier key-frame is triangulated from that observation and the
// current one.  The caller supplies, per stream and observer key-frame (slot kf % nkf of `tab`, 64 doubles), what mapper.jl:226-231
// computes once per observer: P2 = K * rel_pose_inv, rel_pose_inv, rel_pose = observer.cw * frame.wc, and observer.wc -- all
// column-major 4 x 4 -- so their arithmetic stays the host's (Manifolds' inv(SE3, .)).  Gates as slam_triangulate's temporal mode:
// a failed gate removes the observation only when the rotation-compensated parallax exceeds min_parallax (20 px), otherwise the
// point is accepted as it is (:244-258).
struct KTempArgs {
    const double *par;        // S x 32: [16..19] fx fy cx cy, [20..23] k1 k2 p1 p2
    const double *tab;        // S x nkf x 64
    const int *kf_cur, *kf_lo; int nkf;
    double max_error, min_depth, min_parallax;
    uint8_t *flags;
};
__device__ __forceinline__ void kp_undistort(const double *par, double y, double x, double &uy, double &ux)
{
    const double fx = par[16], fy = par[17], cx = par[18], cy = par[19], k1 = par[20], k2 = par[21], p1 = par[22], p2 = par[23];
    const double ny = (y - cy) / fy, nx = (x - cx) / fx;
    const double s0 = ny * ny, s1 = nx * nx, r2 = s0 + s1;
    const double rd = (1.0 + k1 * r2) + k2 * (r2 * r2);
    const double pp = ny * nx;
    const double dtx = 2 * p1 * pp + p2 * (r2 + 2 * s0), dty = p1 * (r2 + 2 * s1) + 2 * p2 * pp;
    uy = (rd * ny + dty) * fy + cy; ux = (rd * nx + dtx) * fx + cx;
}
__device__ __forceinline__ void k_kpset_tri_temporal_slot(const KpsetView &K, const KTempArgs &T, const int *work, int i)
{
    const size_t q = (size_t)work[i];
    if (K.is3d[q] || !K.haskf[q]) return;                        // get_2d_keypoints; keypoints no key-frame has observed yet
    const int s = (int)(q / K.cap), kf = K.fkf[q];
    if (kf == T.kf_cur[s] || kf < T.kf_lo[s]) return;            // :216 the frame itself is the first observer; observer no longer in the table
    const double *par = T.par + 32 * (size_t)s;
    const double *E = T.tab + ((size_t)s * T.nkf + (kf % T.nkf)) * 64;
    const double *P2 = E, *T21 = E + 16, *REL = E + 32, *WOB = E + 48;
    const double fx = par[16], fy = par[17], cx = par[18], cy = par[19];
    double y1, x1, y2, x2;
    kp_undistort(par, K.fyx[2 * q], K.fyx[2 * q + 1], y1, x1);  // obup
    kp_undistort(par, K.yx[2 * q], K.yx[2 * q + 1], y2, x2);    // kpup
    // parallax = |obup - project(camera, R(rel_pose) * kp.position)|, :236-237
    const double bx = (x2 - cx) / fx, by = (y2 - cy) / fy;
    const double rx = (REL[0] * bx + REL[4] * by) + REL[8] * 1.0, ry = (REL[1] * bx + REL[5] * by) + REL[9] * 1.0, rz = (REL[2] * bx + REL[6] * by) + REL[10] * 1.0;
    const double qy = fy * ry / rz + cy, qx = fx * rx / rz + cx;
    const double pdy = y1 - qy, pdx = x1 - qx;
    const bool gated = sqrt(pdy * pdy + pdx * pdx) > T.min_parallax;
    // P1 = K * I
    const double P1[16] = {fx, 0, 0, 0, 0, fy, 0, 0, cx, cy, 1, 0, 0, 0, 0, 1};
    double A[16], S[16], v[4];
    for (int j = 0; j < 4; j++) {
        A[0 + j] = x1 * P1[2 + 4 * j] - P1[0 + 4 * j];
        A[4 + j] = y1 * P1[2 + 4 * j] - P1[1 + 4 * j];
        A[8 + j] = x2 * P2[2 + 4 * j] - P2[0 + 4 * j];
        A[12 + j] = y2 * P2[2 + 4 * j] - P2[1 + 4 * j];
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
    bool ok = !(L2 < T.min_depth && gated);
    double R[3];
    for (int r = 0; r < 3; r++) R[r] = ((T21[r] * L0 + T21[r + 4] * L1) + T21[r + 8] * L2) + T21[r + 12] * L3;
    if (ok && R[2] < T.min_depth && gated) ok = false;
    if (ok) {
        const double iz = 1.0 / L2;
        const double py = fy * L1 * iz + cy, px = fx * L0 * iz + cx;
        const double dy = y1 - py, dx = x1 - px;
        if (sqrt(dy * dy + dx * dx) > T.max_error && gated) ok = false;
    }
    if (ok) {
        const double iz = 1.0 / R[2];
        const double py = fy * R[1] * iz + cy, px = fx * R[0] * iz + cx;
        const double dy = y2 - py, dx = x2 - px;
        if (sqrt(dy * dy + dx * dx) > T.max_error && gated) ok = false;
    }
    if (ok) {
        for (int r = 0; r < 3; r++) K.xyz[3 * q + r] = ((WOB[r] * L0 + WOB[r + 4] * L1) + WOB[r + 8] * L2) + WOB[r + 12] * L3;   // project_camera_to_world(observer_kf, .)
        K.is3d[q] = 1;
    } else T.flags[q] = 1;                                        // remove_mappoint_obs!(map_manager, id, frame.kfid)
}
// (the grid is sized from the host's bound of the list lengths, which is only a hint: the loop covers every live slot whatever it was)
__global__ __launch_bounds__(64) void k_kpset_tri_temporal(KpsetView K, KTempArgs T, const int *work, const int *ntot)
{
    const int n = ntot[0];
    for (int i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) k_kpset_tri_temporal_slot(K, T, work, i);
}

extern "C" {

int slam_kpset_destroy(slam_kpset *ks);

int slam_kpset_create(slam_ctx *ctx, int S, int cap, slam_kpset **out)
{
    ARG_TRY(ctx, ctx != nullptr && out != nullptr && S >= 1 && S <= 128 && cap >= 1 && (size_t)S * cap < (1u << 30));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)S * cap;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += al256(b); return o; };
    const size_t o_yx = take(n * 16), o_oyx = take(n * 16), o_syx = take(n * 16), o_xyz = take(n * 24), o_id = take(n * 8);
    const size_t o_3d = take(n), o_st = take(n), o_ss = take(n), o_cnt = take((size_t)S * 4), o_work = take(n * 4), o_nt = take(64);
    const size_t o_nid = take((size_t)S * 8), o_par = take((size_t)8 * S * 32 * 8), o_kyx = take(n * 16), o_hk = take(n), o_fyx = take(n * 16), o_fkf = take(n * 4), o_kfc = take((size_t)S * 4);
    slam_kpset *ks = new slam_kpset();
    ks->device = ctx->device; ks->S = S; ks->cap = cap;
    hipError_t e = hipMalloc((void **)&ks->base, off);
    if (e == hipSuccess) e = hipMemsetAsync(ks->base, 0, off, ctx->stream);
    if (e == hipSuccess) e = slam_stream_wait(ctx->stream);
    if (e != hipSuccess) { if (ks->base) (void)hipFree(ks->base); delete ks; return slam_fail(ctx, SLAM_ERR_HIP, "slam_kpset_create: %s", hipGetErrorString(e)); }
    char *B = ks->base;
    ks->yx = (double *)(B + o_yx); ks->oyx = (double *)(B + o_oyx); ks->syx = (double *)(B + o_syx); ks->xyz = (double *)(B + o_xyz);
    ks->id = (int64_t *)(B + o_id); ks->is3d = (uint8_t *)(B + o_3d); ks->stereo = (uint8_t *)(B + o_st); ks->st = (uint8_t *)(B + o_ss);
    ks->count = (int *)(B + o_cnt); ks->work = (int *)(B + o_work); ks->ntot = (int *)(B + o_nt); ks->next_id = (int64_t *)(B + o_nid);
    ks->par = (double *)(B + o_par); ks->kyx = (double *)(B + o_kyx); ks->haskf = (uint8_t *)(B + o_hk);
    ks->fyx = (double *)(B + o_fyx); ks->fkf = (int *)(B + o_fkf); ks->kfcount = (int *)(B + o_kfc);
    e = hipHostMalloc((void **)&ks->par_host, (size_t)8 * S * 32 * 8);
    for (int i = 0; i < 8 && e == hipSuccess; i++) { e = hipEventCreateWithFlags(&ks->par_ev[i], hipEventDisableTiming); if (e == hipSuccess) e = hipEventRecord(ks->par_ev[i], ctx->stream); }
    if (e != hipSuccess) { slam_kpset_destroy(ks); return slam_fail(ctx, SLAM_ERR_HIP, "slam_kpset_create: %s", hipGetErrorString(e)); }
    *out = ks;
    return SLAM_OK;
}

int slam_kpset_destroy(slam_kpset *ks)
{
    if (!ks) return SLAM_OK;
    (void)hipSetDevice(ks->device);
    (void)hipDeviceSynchronize();
    if (ks->base) (void)hipFree(ks->base);
    if (ks->par_host) (void)hipHostFree(ks->par_host);
    for (int i = 0; i < 8; i++) if (ks->par_ev[i]) (void)hipEventDestroy(ks->par_ev[i]);
    delete ks;
    return SLAM_OK;
}

int slam_kpset_streams(const slam_kpset *ks) { return ks ? ks->S : SLAM_ERR_ARG; }
int slam_kpset_capacity(const slam_kpset *ks) { return ks ? ks->cap : SLAM_ERR_ARG; }

// replace stream s's list (initialisation, tests); ids == NULL: 0 .. n-1, the stream's id counter moves past them
int slam_kpset_upload(slam_ctx *ctx, slam_kpset *ks, int s, const double *yx, const uint8_t *is3d, const double *xyz, const int64_t *ids, int n)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && s >= 0 && s < ks->S && n >= 0 && n <= ks->cap && (n == 0 || (yx != nullptr && is3d != nullptr)));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t b = (size_t)s * ks->cap;
    std::vector<int64_t> idv((size_t)n);
    int64_t mx = -1;
    for (int i = 0; i < n; i++) { idv[i] = ids ? ids[i] : i; mx = idv[i] > mx ? idv[i] : mx; }
    const int64_t next = mx + 1;
    if (n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(ks->yx + 2 * b, yx, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ks->is3d + b, is3d, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        if (xyz) HIP_TRY(ctx, hipMemcpyAsync(ks->xyz + 3 * b, xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
        else HIP_TRY(ctx, hipMemsetAsync(ks->xyz + 3 * b, 0, (size_t)n * 24, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ks->id + b, idv.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ks->stereo + b, 0, (size_t)n, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ks->haskf + b, 0, (size_t)n, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ks->count + s, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ks->next_id + s, &next, 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

// read stream s's list back (every output may be NULL); cap_out = capacity of the caller's arrays in keypoints
int slam_kpset_download(slam_ctx *ctx, slam_kpset *ks, int s, double *yx, uint8_t *is3d, double *xyz, int64_t *ids,
                        double *stereo_yx, uint8_t *has_stereo, int cap_out, int *n_out)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && s >= 0 && s < ks->S && n_out != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, ks->count + s, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    *n_out = n;
    if (n > cap_out) return slam_fail(ctx, SLAM_ERR_CAPACITY, "slam_kpset_download: %d keypoints but cap = %d", n, cap_out);
    const size_t b = (size_t)s * ks->cap;
    if (n > 0) {
        if (yx) HIP_TRY(ctx, hipMemcpyAsync(yx, ks->yx + 2 * b, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
        if (is3d) HIP_TRY(ctx, hipMemcpyAsync(is3d, ks->is3d + b, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        if (xyz) HIP_TRY(ctx, hipMemcpyAsync(xyz, ks->xyz + 3 * b, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
        if (ids) HIP_TRY(ctx, hipMemcpyAsync(ids, ks->id + b, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (stereo_yx) HIP_TRY(ctx, hipMemcpyAsync(stereo_yx, ks->syx + 2 * b, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
        if (has_stereo) HIP_TRY(ctx, hipMemcpyAsync(has_stereo, ks->stereo + b, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    }
    return SLAM_OK;
}

// create_keyframe! (map_manager.jl:60-96) as far as the lists are concerned: the current frame becomes the previous key-frame
// of every keypoint it holds (frames_map[kfid] is a copy of the frame, new keypoints included: call it after slam_kpset_detect)
__global__ __launch_bounds__(256) void k_kpset_keyframe(KpsetView K)
{
    const int s = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= K.count[s]) return;
    const size_t q = (size_t)s * K.cap + j;
    const double y = K.yx[2 * q], x = K.yx[2 * q + 1];
    if (!K.haskf[q]) { K.fyx[2 * q] = y; K.fyx[2 * q + 1] = x; K.fkf[q] = K.kfcount[s]; }     // detected by this key-frame: its first observer
    K.kyx[2 * q] = y; K.kyx[2 * q + 1] = x; K.haskf[q] = 1;
}
__global__ void k_kpset_kf_advance(int *kfcount, int S) { const int s = blockIdx.x * 64 + threadIdx.x; if (s < S) kfcount[s] += 1; }   // (one 64-thread block until round 5: streams 64.. never advanced)
int slam_kpset_keyframe(slam_ctx *ctx, slam_kpset *ks)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_kpset_keyframe, dim3((ks->cap + 255) / 256, ks->S), dim3(256), 0, ctx->stream, view_of(ks));
    hipLaunchKernelGGL(k_kpset_kf_advance, dim3((ks->S + 63) / 64), dim3(64), 0, ctx->stream, ks->kfcount, ks->S);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
// the key-frame observations of stream s's list, host <-> device (restoring state, tests): kyx n x 2 (y, x), has_kf n flags
int slam_kpset_upload_first(slam_ctx *ctx, slam_kpset *ks, int s, const double *first_yx, const int32_t *first_kf, int n, int kf_count)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && s >= 0 && s < ks->S && n >= 0 && n <= ks->cap && (n == 0 || (first_yx != nullptr && first_kf != nullptr)));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t b = (size_t)s * ks->cap;
    if (n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(ks->fyx + 2 * b, first_yx, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ks->fkf + b, first_kf, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ks->kfcount + s, &kf_count, 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}
int slam_kpset_download_first(slam_ctx *ctx, slam_kpset *ks, int s, double *first_yx, int32_t *first_kf, int cap_out, int *n_out, int *kf_count)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && s >= 0 && s < ks->S && n_out != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int n = 0, kc = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, ks->count + s, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&kc, ks->kfcount + s, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    *n_out = n; if (kf_count) *kf_count = kc;
    if (n > cap_out) return slam_fail(ctx, SLAM_ERR_CAPACITY, "slam_kpset_download_first: %d keypoints but cap = %d", n, cap_out);
    const size_t b = (size_t)s * ks->cap;
    if (n > 0) {
        if (first_yx) HIP_TRY(ctx, hipMemcpyAsync(first_yx, ks->fyx + 2 * b, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
        if (first_kf) HIP_TRY(ctx, hipMemcpyAsync(first_kf, ks->fkf + b, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    }
    return SLAM_OK;
}
int slam_kpset_upload_keyframe(slam_ctx *ctx, slam_kpset *ks, int s, const double *kyx, const uint8_t *has_kf, int n)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && s >= 0 && s < ks->S && n >= 0 && n <= ks->cap && (n == 0 || (kyx != nullptr && has_kf != nullptr)));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t b = (size_t)s * ks->cap;
    if (n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(ks->kyx + 2 * b, kyx, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ks->haskf + b, has_kf, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    }
    return SLAM_OK;
}
int slam_kpset_download_keyframe(slam_ctx *ctx, slam_kpset *ks, int s, double *kyx, uint8_t *has_kf, int cap_out, int *n_out)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && s >= 0 && s < ks->S && n_out != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, ks->count + s, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    *n_out = n;
    if (n > cap_out) return slam_fail(ctx, SLAM_ERR_CAPACITY, "slam_kpset_download_keyframe: %d keypoints but cap = %d", n, cap_out);
    const size_t b = (size_t)s * ks->cap;
    if (n > 0) {
        if (kyx) HIP_TRY(ctx, hipMemcpyAsync(kyx, ks->kyx + 2 * b, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
        if (has_kf) HIP_TRY(ctx, hipMemcpyAsync(has_kf, ks->haskf + b, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    }
    return SLAM_OK;
}

// the one small device -> host copy of a step: the S list lengths (synchronises ctx's stream)
int slam_kpset_counts(slam_ctx *ctx, slam_kpset *ks, int32_t *counts)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && counts != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *h;
    int rc = slam_pinned(ctx, std::max<size_t>(256, (size_t)ks->S * 4), &h);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h, ks->count, (size_t)ks->S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    memcpy(counts, h, (size_t)ks->S * 4);
    return SLAM_OK;
}

// remove the keypoints whose flag is set (flags_dev: S x cap bytes in HBM, slot order): map culling, outliers of the pose
// estimators, ... -- stable compaction on the device, returns after enqueueing
int slam_kpset_remove(slam_ctx *ctx, slam_kpset *ks, const uint8_t *flags_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && flags_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return kpset_compact(ctx, ks, 1, flags_dev);
}

int slam_kpset_triangulate(slam_ctx *ctx, slam_kpset *ks, const double *P1, const double *P2, const double *T21,
                           const double *cam1, const double *cam2, const double *Twc, double max_error, double min_depth, int n_bound)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && P1 && P2 && T21 && cam1 && cam2 && Twc);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    KTriArgs T;
    memcpy(T.P1, P1, sizeof T.P1); memcpy(T.P2, P2, sizeof T.P2); memcpy(T.T21, T21, sizeof T.T21);
    memcpy(T.cam1, cam1, sizeof T.cam1); memcpy(T.cam2, cam2, sizeof T.cam2);
    T.max_error = max_error; T.min_depth = min_depth;
    int rc = kpset_stage_params(ctx, ks, Twc, (size_t)ks->S * 16, &T.Twc);
    if (rc) return rc;
    rc = kpset_build_worklist(ctx, ks);
    if (rc) return rc;
    const int nb = n_bound > 0 && n_bound < ks->S * ks->cap ? n_bound : ks->S * ks->cap;
    hipLaunchKernelGGL(k_kpset_triangulate, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, view_of(ks), T, (const int *)ks->work, (const int *)ks->ntot);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}

int slam_kpset_triangulate_temporal(slam_ctx *ctx, slam_kpset *ks, const double *params, const double *tab, int nkf,
                                    const int32_t *kf_cur, const int32_t *kf_lo, double max_error, double min_depth, double min_parallax,
                                    int n_bound)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && params && tab && kf_cur && kf_lo && nkf >= 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int S = ks->S;
    const size_t nc = (size_t)S * ks->cap, tb = al256((size_t)S * nkf * 64 * 8), ib = al256((size_t)S * 4);
    char *scr;
    int rc = slam_scratch2(ctx, tb + 2 * ib + al256(nc), (void **)&scr);
    if (rc) return rc;
    void *hv;
    rc = slam_pinned(ctx, tb + 2 * ib, &hv);
    if (rc) return rc;
    char *h = (char *)hv;
    memcpy(h, tab, (size_t)S * nkf * 64 * 8); memcpy(h + tb, kf_cur, (size_t)S * 4); memcpy(h + tb + ib, kf_lo, (size_t)S * 4);
    HIP_TRY(ctx, hipMemcpyAsync(scr, h, tb + 2 * ib, hipMemcpyHostToDevice, ctx->stream));
    KTempArgs T;
    rc = kpset_stage_params(ctx, ks, params, (size_t)S * 32, &T.par);
    if (rc) return rc;
    T.tab = (const double *)scr; T.kf_cur = (const int *)(scr + tb); T.kf_lo = (const int *)(scr + tb + ib); T.nkf = nkf;
    T.max_error = max_error; T.min_depth = min_depth; T.min_parallax = min_parallax;
    T.flags = (uint8_t *)(scr + tb + 2 * ib);
    HIP_TRY(ctx, hipMemsetAsync(T.flags, 0, nc, ctx->stream));
    rc = kpset_build_worklist(ctx, ks);
    if (rc) return rc;
    const int nb = n_bound > 0 && n_bound < S * ks->cap ? n_bound : S * ks->cap;
    hipLaunchKernelGGL(k_kpset_tri_temporal, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, view_of(ks), T, (const int *)ks->work, (const int *)ks->ntot);
    HIP_TRY(ctx, hipGetLastError());
    rc = kpset_compact(ctx, ks, 1, T.flags);
    if (rc) return rc;
    // the pinned table is read by the copy above: it must be gone from the host block before the next call reuses it
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

}  // extern "C"
