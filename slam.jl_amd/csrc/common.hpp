// common.hpp -- shared host-side plumbing for libslamhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/slamhip.h"

#define SLAM_MAX_LEVELS 8

struct slam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int cus = 0;                          // compute units the stream may use (0: all of the device)
    int dev_cus = 0;                      // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    bool arch_ok = false;                 // gfx942 / gfx950: atomics with agent scope are performed at the memory side (k_cum_fused's row segments rely on it)
    bool xwg_ok = false;                  // cross-workgroup hand-overs through memory-side atomics are relied on only where they were validated:
                                          // gfx942 / gfx950 (workgroups b and b + 8 share an XCD's L2), stream not CU-masked (ctx_probe_device)
    int pool_class = 99;                  // scheduling class the stream is parked under when the context goes (ctx.hip); 99: not pooled (CU-masked)
    std::string err;
    // grow-only device scratch and pinned host staging
    void *scratch = nullptr; size_t scratch_bytes = 0;
    void *scratch2 = nullptr; size_t scratch2_bytes = 0;
    void *pinned = nullptr; size_t pinned_bytes = 0;
    hipEvent_t wait_event = nullptr;      // slam_ctx_wait_for
    hipStream_t stream2 = nullptr;        // a second stream of the same scheduling class (ctx_aux_stream: the second half of a batch of BA windows), with its fork / join events
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    // optional device-side timing (hipEvents on ctx->stream), see slam_prof_*
    bool prof_on = false;
    struct ProfSpan { int id; hipEvent_t a, b; };
    std::vector<ProfSpan> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    std::vector<std::string> prof_names;
    std::vector<double> prof_ms;
    std::vector<long long> prof_cnt;
};

extern "C" hipStream_t ctx_aux_stream(slam_ctx *c);          // nullptr if it cannot be created (ctx.hip)
extern "C" void ctx_probe_device(slam_ctx *c);      // fills dev_cus / xwg_ok (ctx.hip)

// RAII span: records a hipEvent pair around the enclosed launches when profiling is on
struct ProfScope {
    slam_ctx *c; int idx;
    ProfScope(slam_ctx *ctx, const char *name);
    ~ProfScope();
};

// Device-side view of one pyramid level: planes are column-major H x W (y fastest) with a column
// pitch of P >= H doubles, P a multiple of 16 (every column starts on a 128-byte line).
struct LevelView {
    double *L, *Iy, *Ix, *Iyy, *Ixx, *Iyx;
    int H, W, P;
};
struct PyrView {
    LevelView lv[SLAM_MAX_LEVELS];
    int levels;
};

struct slam_pyr {
    int device = 0;
    int levels = 0;                       // total layers = pyramid_levels + 1
    int H[SLAM_MAX_LEVELS], W[SLAM_MAX_LEVELS];
    int P[SLAM_MAX_LEVELS];               // column pitch in doubles (H rounded up to 16)
    int64_t off[SLAM_MAX_LEVELS + 1];     // plane offsets in doubles (sum of P_l * W_l)
    struct Alloc { double *base = nullptr; double *ck = nullptr; double *tot = nullptr; const void **srctab = nullptr; double *xc = nullptr; int *xf = nullptr; int xseg = 0; int refs = 0; std::mutex graph_mu; };   // xc / xf / xseg: carry rows and flags of k_cum_fused's row segments (frames taller than 512 rows)   // shared by the members of a batch; srctab: 64 source-image pointers (fused ingest)
    Alloc *alloc = nullptr;
    size_t zstride = 0;                   // doubles between consecutive images of a batch (7 * off[levels])
    int batch_index = 0, batch_size = 1;
    bool tol_planes = false;              // last update ran in tolerance mode (mode 3): matches between two such pyramids take the contracted-arithmetic tracking kernels
    bool target_only = false;             // last update built the gradient / integral planes of level 0 only (SLAM_PYR_TARGET_ONLY): usable as the `to` pyramid of a match
    double *planes = nullptr;             // 6 planes x off[levels] doubles (inside alloc)
    double *tmp = nullptr;                // blur scratch, off[levels] doubles
    double *ck = nullptr;                 // batches: checkpoint scratch of the bandwidth-bound row kernel (owned by alloc)
    double *norm = nullptr;               // NA() normaliser per level (ctor mode), lazily built
    double norm_sigma = -1.0;
    // hipGraph replay of the build (constructed lazily per (mode, sigma, S, source kind): explicit kernel nodes, no stream capture)
    struct Graph { int mode; double sigma; int S; size_t ckmin; int src_kind; hipGraphExec_t exec; };
    std::vector<Graph> graphs;
    bool graph_failed = false;
    PyrView view;
    double *plane(int p, int l) const { return planes + (int64_t)p * off[levels] + off[l]; }
};

// Device-resident keypoint lists of S lock-stepped streams: stream s owns the slots
// [s * cap, s * cap + count[s]) of every per-keypoint array; all of it lives in one allocation.
struct slam_kpset {
    int device = 0, S = 0, cap = 0;
    char *base = nullptr;
    double *yx = nullptr;        // [S cap][2] pixel (y, x) in the current left image
    double *oyx = nullptr;       // [S cap][2] positions returned by the last temporal match (scratch)
    double *syx = nullptr;       // [S cap][2] stereo pixel (right image), valid where stereo != 0
    double *xyz = nullptr;       // [S cap][3] map point, valid where is3d != 0
    double *kyx = nullptr;       // [S cap][2] pixel (y, x) in the previous key-frame, valid where haskf != 0 (slam_kpset_keyframe)
    uint8_t *haskf = nullptr;    // [S cap] the keypoint is observed by the previous key-frame
    double *fyx = nullptr;       // [S cap][2] pixel (y, x) in the FIRST key-frame that observed the keypoint (the one that detected it)
    int *fkf = nullptr;          // [S cap] that key-frame's id (per-stream counter), valid where haskf != 0
    int *kfcount = nullptr;      // [S] number of key-frames created so far = id of the next one
    int64_t *id = nullptr;       // [S cap] keypoint id (per stream, ascending in creation order)
    uint8_t *is3d = nullptr, *stereo = nullptr, *st = nullptr;   // flags; st: status of the last match (0 lost, 1 tracked, 2 skipped)
    int *count = nullptr;        // [S]
    int *work = nullptr;         // [S cap] live slots, streams back to back
    int *ntot = nullptr;         // [4]: number of live slots, ...
    int64_t *next_id = nullptr;  // [S]
    // per-stream parameters of a call (prior shift / pose): ring of 8 slots of S x 32 doubles, staged through pinned host
    // memory with an event per slot, so that the enqueue-only calls never wait for an earlier call's copy
    double *par = nullptr, *par_host = nullptr;
    hipEvent_t par_ev[8] = {};
    int par_slot = 0;
};
// stage `n` doubles (<= S x 32) of per-stream parameters into the next ring slot; returns the device pointer
int kpset_stage_params(slam_ctx *ctx, slam_kpset *ks, const double *host, size_t n, const double **dev_out);

extern thread_local std::string g_slam_err;

// The synchronous seams end with a wait for the context's stream.  The interrupt-driven hipStreamSynchronize costs tens of
// microseconds per call; instead of switching the whole DEVICE to hipDeviceScheduleSpin (process-wide: torch and every other
// user of the runtime would spin too), only the library's own waits poll -- for at most SLAMHIP_SPIN_US microseconds
// (default 2000, 0 = never), then they block.
hipError_t slam_stream_wait(hipStream_t s);

int slam_fail(slam_ctx *ctx, int code, const char *fmt, ...);
int slam_scratch(slam_ctx *ctx, size_t bytes, void **out);
int slam_scratch2(slam_ctx *ctx, size_t bytes, void **out);
int slam_pinned(slam_ctx *ctx, size_t bytes, void **out);

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return slam_fail((ctx), SLAM_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define ARG_TRY(ctx, cond)                                                                  \
    do {                                                                                    \
        if (!(cond)) return slam_fail((ctx), SLAM_ERR_ARG, "argument check failed: %s (%s:%d)", #cond, __FILE__, __LINE__); \
    } while (0)

// KernelFactors.IIRGaussian coefficients (host side; passed to kernels by value)
#define SLAM_PYR_TARGET_ONLY 16          /* update mode flags, see include/slamhip.h */
#define SLAM_PYR_CHAIN 32
#define SLAM_PYR_FLAGS (SLAM_PYR_TARGET_ONLY | SLAM_PYR_CHAIN)
struct IIRCoef {
    double a1, a2, a3, scale, M[9], inv1masum /* 1-asum */, inv1mbsum /* 1-bsum */;
};
IIRCoef slam_iir_coef(double sigma);
int slam_gaussian_taps(double sigma, double *w);   // Kernel.gaussian 1-D factor

// pinhole intrinsics; the argument block of one single-pose refinement (k_pnp / k_pnp_batch in ba_single.hip; filled on the device by
// the keypoint-set pose seam in pose.hip)
struct Cam { double fx, fy, cx, cy; };
struct PnPArgs {
    Cam cam; const double *px; const double *pts; int n;
    double X0[6]; int iters_fast, iterations; double depth_eps, repr_eps;
    uint8_t *outl; double *result;   // [X(6), err_init, err_final, n_outliers, identity, iters1, iters2]
};
int pnp_launch_device(slam_ctx *ctx, int S, const PnPArgs *args_dev);     // S problems, argument blocks already in device memory

// kpset plumbing shared by kpset.hip / lk.hip / detect.hip
int kpset_build_worklist(slam_ctx *ctx, slam_kpset *ks);
int kpset_compact(slam_ctx *ctx, slam_kpset *ks, int mode, const uint8_t *flags_dev);
// per-module entry points used across files
int slam_detect_device(slam_ctx *ctx, const double *img_dev, int H, int W, int pitch, const double *cur_yx, int n_cur,
                       int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                       double sigma_mask, double min_response, int64_t *out_rc, int cap, int *n_out);
