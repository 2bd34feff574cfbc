// ba_batch.hip -- bundle_adjustment! for S windows in one set of launches (slam_local_ba_batch, _begin / _end): the window-indexed wrappers of the
// kernels of ba_device.hpp (table of BAWin read through the constant address space), the matrix-core Schur build, the host half of the call
// (worker pool, arena layout, retry of k_ba_window on one workgroup).  reference: src/estimator.jl:78-99, :317-347; src/bundle_adjustment.jl:1-111.
#include "ba_device.hpp"

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
    if (w.pad || w.pad2) return;                             // the window is k_ba_window's / k_schur_groups_m's (pad2: the matrix-core build takes it)
    if ((int)blockIdx.x >= w.d.ngrp) return;
    schur_groups_body<TT>(w.d, 0.0, ignore_outliers, 1);
}
template <int TT> __global__ __launch_bounds__(TT) void k_schur_groups_m(const BAWin *tab, int ignore_outliers)
{
    const BAWin w = ba_win(tab);
    if (w.pad || !w.pad2) return;                            // the window is k_ba_window's / the vector build's
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
    if ((int)blockIdx.x >= w.d.ngrp || (w.pad2 != 0) != RECOMP) return;      // (a window of the matrix-core build stores nothing of its evaluation: RECOMP forms it again; the others read it back)
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



// Worker threads for the host half of a batch (structure analysis, staging, result scatter of S windows): created once, parked on a
// condition variable between calls -- starting 15 threads per phase cost more than the work they did (128 windows: 2.3 ms of a 6.3 ms
// call).  Callers from several contexts take turns (run_mu).  Windows are handed out one at a time from an atomic counter.
namespace {
// A run hands out task indices from one atomic word tagged with the run's generation, so that a worker only ever executes a task with the function of
// the run the index belongs to; only as many workers are woken as there are tasks beside the caller's (a one-window set-up of four tasks does not
// wake and collect 31 threads), and a worker that wakes late finds nothing to claim and goes back to sleep.
struct BAPool {
    std::vector<std::thread> th;
    std::mutex mu, run_mu;
    std::condition_variable cv, cv_done;
    const std::function<void(int)> *fn = nullptr;
    int n = 0; unsigned gen = 0; bool stop = false;
    std::atomic<unsigned long long> next{0};      // generation << 32 | next task index
    std::atomic<int> done{0};
    // claims are one fetch_add each (no retry under contention).  A worker that read run g's function and then draws an index of a LATER run g' (it was
    // late: run g is over) owns that task of run g' -- which therefore cannot have completed -- and reads g's successor under the mutex before executing it.
    void take(const std::function<void(int)> *f, int count, unsigned g)
    {
        for (;;) {
            const unsigned long long cur = next.fetch_add(1, std::memory_order_acq_rel);
            if ((unsigned)(cur >> 32) != g) { std::lock_guard<std::mutex> lk(mu); f = fn; count = n; g = gen; if ((unsigned)(cur >> 32) != g) return; }
            if ((int)(unsigned)cur >= count) return;
            (*f)((int)(unsigned)cur);
            if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == count) { std::lock_guard<std::mutex> lk(mu); cv_done.notify_one(); }
        }
    }
    explicit BAPool(int workers)
    {
        for (int t = 0; t < workers; t++)
            th.emplace_back([this] {
                unsigned seen = 0;
                for (;;) {
                    const std::function<void(int)> *f; int count;
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; f = fn; count = n; }
                    take(f, count, seen);
                }
            });
    }
    ~BAPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &x : th) x.join(); }
    void run(int count, const std::function<void(int)> &f)
    {
        std::lock_guard<std::mutex> turn(run_mu);
        if (th.empty() || count <= 1) { for (int z = 0; z < count; z++) f(z); return; }
        unsigned g;
        { std::lock_guard<std::mutex> lk(mu); fn = &f; n = count; g = ++gen; done.store(0, std::memory_order_relaxed); next.store((unsigned long long)g << 32, std::memory_order_release); }
        if (count - 1 >= (int)th.size()) cv.notify_all(); else for (int k = 0; k < count - 1; k++) cv.notify_one();
        take(&f, count, g);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done.load(std::memory_order_acquire) >= count; });
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
void ba_parallel_for(int count, const std::function<void(int)> &fn) { ba_pool().run(count, fn); }
int ba_pool_threads() { return (int)ba_pool().th.size() + 1; }

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
        // a large upload travels in parts: the staged bytes of the first windows go out while the pool stages the next ones (128 x P20: 203 MB = 4 ms of PCIe
        // beside 2.5 ms of staging); the tables and the last part follow with the launch sequence below
        // (a part keeps every thread of the pool busy -- one window per task -- and a small upload is not worth the extra runs of the pool: measured, 128 reference-shaped
        //  windows = 33 MB gained 0.2 ms per call and lost 25 % with two calls in flight; 32 x P100 in four parts of eight windows lost 4 ms)
        static const int parts_cap = [] { const char *v = getenv("SLAMHIP_BA_UPLOAD_PARTS"); return v ? atoi(v) : 4; }();      // (measurement knob)
        const int n_parts = up_total < ((size_t)64 << 20) ? 1 : std::max(1, std::min(parts_cap, NB / std::max(nthr, 1)));
        const int n_early = n_parts > 1 ? NB - (NB + n_parts - 1) / n_parts : 0;            // windows whose bytes leave before the launch sequence: all but the last part
        const int early_step = n_parts > 1 ? std::max(1, (n_early + n_parts - 2) / (n_parts - 1)) : NB;
        auto emit_window = [&](int zz) {
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
        };
        bool early_sent = false;
        if (n_early > 0) {
            early_sent = true;
            for (int b0 = 0; b0 < NB;) {
                const int b1 = b0 < n_early ? std::min(n_early, b0 + early_step) : NB;
                pool.run(b1 - b0, [&](int zz) { emit_window(b0 + zz); });
                if (b1 <= n_early && early_sent) early_sent = hipMemcpyAsync(A + up[b0], stage + up[b0], up[b1] - up[b0], hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
                b0 = b1;
            }
        } else pool.run(S, emit_window);
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
        // LDS beside two more workgroups; SLAMHIP_BA_NO_MFMA=1 keeps the vector kernel (A/B timing, and the parity reference of the tests).  The vector kernel's lds_sg
        // and the cost-only linearisation are shared: the vector build stores its own evaluation, the matrix-core build none
        int ug_n = 6, ug_ob = 8, ug_sb = 8;                     // k_update_groups_b's LDS arrays at the batch's own sizes
        for (int k = 0; k < NB; k++) if (!pl[batch[k]].err && !tab_h[k].pad) { ug_n = std::max(ug_n, tab_h[k].d.n); ug_ob = std::max(ug_ob, tab_h[k].d.sg_ob); ug_sb = std::max(ug_sb, tab_h[k].d.sg_sb); }
        const size_t lds_ug = ug_lds_bytes(ug_n, ug_ob, ug_sb);
        static const bool no_mfma = getenv("SLAMHIP_BA_NO_MFMA") != nullptr;
        // per window (BAWin::pad2): one window whose matrix Y does not fit (a wide band with few observations per point) does not take the matrix cores from the others
        size_t lds_m = 0; int n_mfma = 0, n_vec = 0;
        for (int k = 0; k < NB; k++) if (!pl[batch[k]].err && !tab_h[k].pad) {
            const size_t b = sgm_lds_bytes(tab_h[k].d.whb, tab_h[k].d.P, tab_h[k].d.sg_ob, tab_h[k].d.sg_sb, tab_h[k].d.sg_hp);
            const bool m = !no_mfma && TT == 256 && b <= 64 * 1024;
            tab_h[k].pad2 = m ? 1 : 0;
            if (m) { lds_m = std::max(lds_m, b); n_mfma++; } else n_vec++;
        }
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
        // the launch-per-phase windows [n0, n0 + nb) on stream q: both passes.  A batch of >= 32 such windows runs as TWO halves on two streams: while one half sits
        // in its latency-bound kernels (k_band_solve_b: one workgroup per window, k_trial_poses_b, k_control_b -- ~75 us of an iteration) the other half's builds fill the chip
        auto run_pass = [&](hipStream_t q, int n0, int nb, int ignore, int iters) {
            const BAWin *tb = tab + n0;
            hipLaunchKernelGGL(k_linearize_b, dim3(gx_obs, nb), dim3(256), 0, q, tb, ignore, 0);
            hipLaunchKernelGGL(k_pass_start_b, dim3(1, nb), dim3(256), 0, q, tb, ignore ? 1 : 0);
            for (int it = 1; it <= iters; it++) {
                if (n_mfma) hipLaunchKernelGGL(k_schur_groups_m<256>, dim3(gx_grp, nb), dim3(256), lds_m, q, tb, ignore);
                if (n_vec && TT == 256) hipLaunchKernelGGL(k_schur_groups_b<256>, dim3(gx_grp, nb), dim3(256), lds_sg, q, tb, ignore);
                else if (n_vec) hipLaunchKernelGGL(k_schur_groups_b<SG_T>, dim3(gx_grp, nb), dim3(SG_T), lds_sg, q, tb, ignore);
                hipLaunchKernelGGL(k_schur_reduce_b, dim3(gx_red, nb), dim3(256), 0, q, tb);
                hipLaunchKernelGGL(k_band_solve_b, dim3(1, nb), dim3(BS_T), lds_band, q, tb);
                hipLaunchKernelGGL(k_trial_poses_b, dim3(1, nb), dim3(64), 0, q, tb);
                if (n_mfma) hipLaunchKernelGGL((k_update_groups_b<256, true>), dim3(gx_grp, nb), dim3(256), lds_ug, q, tb, ignore, ug_n, ug_ob, ug_sb);
                if (n_vec && TT == 256) hipLaunchKernelGGL((k_update_groups_b<256, false>), dim3(gx_grp, nb), dim3(256), lds_ug, q, tb, ignore, ug_n, ug_ob, ug_sb);
                else if (n_vec) hipLaunchKernelGGL((k_update_groups_b<SG_T, false>), dim3(gx_grp, nb), dim3(SG_T), lds_ug, q, tb, ignore, ug_n, ug_ob, ug_sb);
                hipLaunchKernelGGL(k_control_b, dim3(1, nb), dim3(256), 0, q, tb);
            }
        };
        auto both_passes = [&](hipStream_t q, int n0, int nb) {
            run_pass(q, n0, nb, 0, iters_fast);
            hipLaunchKernelGGL(k_outliers_b, dim3(gx_obs, nb), dim3(256), 0, q, tab + n0, repr_eps, 1e-6);
            hipLaunchKernelGGL(k_outlier_count_b, dim3(1, nb), dim3(256), 0, q, tab + n0);
            run_pass(q, n0, nb, 1, iterations);
        };
        static const bool one_stream = getenv("SLAMHIP_BA_ONE_STREAM") != nullptr;      // (measurement knob)
        hipStream_t st2 = (!all_small && NB >= 32 && !one_stream) ? ctx_aux_stream(ctx) : nullptr;
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
            if (attempt == 0 && early_sent) {                      // (the first half of the windows' bytes is on its way already)
                e = hipMemcpyAsync(A, stage, up[0], hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(A + up[n_early], stage + up[n_early], up_total - up[n_early], hipMemcpyHostToDevice, st);
            } else e = hipMemcpyAsync(A, stage, up_total, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemsetAsync(A + zero_base, 0, ze[NB], st);
            if (e != hipSuccess) break;
            (void)hipEventRecord(e0, st);
            if (NS_ > 0) {
                const int *list_d = (const int *)(A + tab_bytes + rtab_bytes - al((size_t)NB * 4));
                hipLaunchKernelGGL(k_ba_window, dim3(two ? two_grid : NS_), dim3(BW_T), lds_bw, st, tab, list_d, NS_, two, iters_fast, iterations, repr_eps, 1e-6, xlimit);
            }
            if (!all_small && st2) {
                const int nA = NB / 2;
                (void)hipEventRecord(ctx->fork_ev, st); (void)hipStreamWaitEvent(st2, ctx->fork_ev, 0);
                both_passes(st2, nA, NB - nA);
                both_passes(st, 0, nA);
                (void)hipEventRecord(ctx->join_ev, st2); (void)hipStreamWaitEvent(st, ctx->join_ev, 0);
            } else if (!all_small) both_passes(st, 0, NB);
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
// worker pool, as slam_local_ba_batch uses it; -N: one window at a time with its passes over the observations split into <= N tasks of the pool, as slam_local_ba does; -1: the same on the calling thread) into malloc'ed staging; out_us = {plan, emit, a 52-bit hash of everything staged and of the observation orders}; returns the number of windows whose set-up failed.  Measurement aid for tuning the host side on any machine (scripts/probes/ba_host_time.py).
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
    if (threads < 0) for (int z = 0; z < S; z++) { pl[z].small_groups = false; pl[z].nthreads = -threads; ba_plan(pl[z]); }     // slam_local_ba's way: one window at a time, its passes split over the pool
    else parallel([&](int z) { ba_plan(pl[z]); });
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<size_t> up(S + 1, 0);
    for (int z = 0; z < S; z++) up[z + 1] = up[z] + pl[z].up_bytes;
    std::vector<char> stage(up[S] + 64);                       // (allocated and zero-filled outside the two timed spans)
    char *fake = (char *)(uintptr_t)0x100000000ull;
    const auto t1b = std::chrono::steady_clock::now();
    if (threads < 0) { for (int z = 0; z < S; z++) if (!pl[z].err) ba_emit(pl[z], fake, fake, fake, stage.data() + up[z]); }
    else parallel([&](int z) { if (!pl[z].err) ba_emit(pl[z], fake, fake, fake, stage.data() + up[z]); });
    const auto t2 = std::chrono::steady_clock::now();
    out_us[0] = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() * 1e-3;
    out_us[1] = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1b).count() * 1e-3;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    for (int z = 0; z < S; z++) if (!pl[z].err) { mix(stage.data() + up[z], pl[z].up_bytes); mix(pl[z].ba->perm.data(), pl[z].ba->perm.size() * sizeof(int)); }
    out_us[2] = (double)(h >> 12);
    int bad = 0;
    for (int z = 0; z < S; z++) bad += pl[z].err != 0;
    return bad;
}

} // extern "C"
