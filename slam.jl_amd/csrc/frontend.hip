// frontend.hip -- one live stream, one C call per frame (include/slamhip.h: slam_frontend_*): the per-frame work of the reference's front-end
// task (src/front_end.jl:58-113, :454-470) and of the mapper's stereo step at key-frames (src/mapper.jl:51-66, :142-183), enqueued back to back
// on the library's streams with the keypoint list resident in HBM.  Host code only: every kernel it launches belongs to the seams it calls
// (slam_pyr_update_u8_dev, slam_kpset_*).  Round 5's live loop paid ~200 us of synchronous Python per frame for the same enqueues.
#include "common.hpp"
#include <chrono>

struct slam_frontend {
    slam_frontend_config c;
    slam_ctx *ctx = nullptr, *ctx_build = nullptr, *ctx_right = nullptr;
    slam_pyr *left[3] = {nullptr, nullptr, nullptr}, *right[2] = {nullptr, nullptr};
    slam_kpset *ks = nullptr;
    slam_event *built[3] = {nullptr, nullptr, nullptr}, *rbuilt[2] = {nullptr, nullptr}, *tracked = nullptr;
    uint8_t *pin = nullptr, *dev8 = nullptr;                 // 5 frames each: left ring (3), right ring (2)
    size_t npix = 0;
    long fed = 0, done = 0;                                  // frames fed / processed
    bool has_right[3] = {false, false, false};
    int n_bound = 0;
    std::string err;
    double us_feed = 0, us_enq = 0, us_wait = 0; long ncalls = 0;   // host time of the three parts of a call (SLAMHIP_FE_HOSTTIME=1 prints the means at destroy)
};

namespace {
int fe_fail(slam_frontend *fe, int rc, slam_ctx *c, const char *what)
{
    fe->err = std::string(what) + ": " + (c ? slam_last_error(c) : slam_last_error(nullptr));
    return rc;
}
#define FE_TRY(fe, c, expr) do { const int rc_ = (expr); if (rc_ != SLAM_OK) return fe_fail(fe, rc_, c, #expr); } while (0)

// upload + build of one fed frame (enqueue only): left on the build context, the key-frame's right image on the right context
int fe_feed(slam_frontend *fe, const uint8_t *left_u8, const uint8_t *right_u8)
{
    const long t = fe->fed;
    const int sl = (int)(t % 3), sr = (int)(t % 2);
    const size_t n = fe->npix;
    // the slot's buffers are free: the frame that used them (t - 3) was processed two calls ago, and slam_frontend_step waits for the
    // tracking stream's read-back before it returns
    // a frame that already lies in pinned (page-locked) host memory -- hipHostMalloc / hipHostRegister, e.g. a capture buffer the application
    // registered once -- is copied from where it is; the caller then keeps it unchanged until the NEXT call returns.  Anything else goes through
    // the front-end's own pinned ring first (450 KB: ~45 us of the calling thread per KITTI frame).
    auto pinned = [](const void *q) { hipPointerAttribute_t a; if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; } return a.type == hipMemoryTypeHost; };
    const uint8_t *lsrc = left_u8;
    if (!pinned(left_u8)) { memcpy(fe->pin + (size_t)sl * n, left_u8, n); lsrc = fe->pin + (size_t)sl * n; }
    hipStream_t sb = (hipStream_t)slam_ctx_stream(fe->ctx_build);
    if (fe->tracked && fe->ctx_build != fe->ctx) FE_TRY(fe, fe->ctx_build, slam_ctx_wait_event(fe->ctx_build, fe->tracked));      // the pyramid being rebuilt is no longer read by a match
    if (hipMemcpyAsync(fe->dev8 + (size_t)sl * n, lsrc, n, hipMemcpyHostToDevice, sb) != hipSuccess) { fe->err = "slam_frontend: copy of the left frame failed"; return SLAM_ERR_HIP; }
    // (the build as one chain, SLAM_PYR_CHAIN -- a third of the graph launch's host time, 15 % more device time -- measured slower here: 2 020 / 4 500 frames/s)
    FE_TRY(fe, fe->ctx_build, slam_pyr_update_u8_dev(fe->ctx_build, fe->left[sl], fe->dev8 + (size_t)sl * n, fe->c.pyr_mode, fe->c.pyr_sigma, 0));
    if (fe->ctx_build != fe->ctx) FE_TRY(fe, fe->ctx_build, slam_event_record(fe->ctx_build, fe->built[sl]));
    fe->has_right[sl] = right_u8 != nullptr;
    if (right_u8) {
        const uint8_t *rsrc = right_u8;
        if (!pinned(right_u8)) { memcpy(fe->pin + (size_t)(3 + sr) * n, right_u8, n); rsrc = fe->pin + (size_t)(3 + sr) * n; }
        hipStream_t sr_ = (hipStream_t)slam_ctx_stream(fe->ctx_right);
        if (fe->tracked) FE_TRY(fe, fe->ctx_right, slam_ctx_wait_event(fe->ctx_right, fe->tracked));
        if (hipMemcpyAsync(fe->dev8 + (size_t)(3 + sr) * n, rsrc, n, hipMemcpyHostToDevice, sr_) != hipSuccess) { fe->err = "slam_frontend: copy of the right frame failed"; return SLAM_ERR_HIP; }
        FE_TRY(fe, fe->ctx_right, slam_pyr_update_u8_dev(fe->ctx_right, fe->right[sr], fe->dev8 + (size_t)(3 + sr) * n,
                                                         fe->c.pyr_mode | (fe->c.right_target_only ? SLAM_PYR_TARGET_ONLY : 0), fe->c.pyr_sigma, 0));
        FE_TRY(fe, fe->ctx_right, slam_event_record(fe->ctx_right, fe->rbuilt[sr]));
    }
    fe->fed++;
    return SLAM_OK;
}

// the oldest fed frame: matching against the frame before it and the key-frame work, enqueued ...
int fe_track(slam_frontend *fe, const double *params, int prior, const double *stereo_params, int stereo_prior, const double *tri,
             const uint8_t *cull_flags_dev)
{
    const long t = fe->done;
    const int sl = (int)(t % 3), sp = (int)((t + 2) % 3), sr = (int)(t % 2);
    const slam_frontend_config &c = fe->c;
    const auto h0t = std::chrono::steady_clock::now();
    if (fe->ctx_build != fe->ctx) FE_TRY(fe, fe->ctx, slam_ctx_wait_event(fe->ctx, fe->built[sl]));
    if (t > 0 && fe->n_bound > 0) {
        if (!params) { fe->err = "slam_frontend_step: params is NULL"; return SLAM_ERR_ARG; }
        FE_TRY(fe, fe->ctx, slam_kpset_flow_match(fe->ctx, fe->ks, fe->left[sp], fe->left[sl], params, prior, c.pyramid_levels, c.pyramid_levels_3d, c.window,
                                                  c.iterations, c.eig_thr, c.eps, c.max_distance, fe->n_bound));
    }
    if (fe->has_right[sl]) {                                   // key-frame: cull, detect, stereo match, triangulate
        if (cull_flags_dev) FE_TRY(fe, fe->ctx, slam_kpset_remove(fe->ctx, fe->ks, cull_flags_dev));
        FE_TRY(fe, fe->ctx, slam_kpset_detect(fe->ctx, fe->ks, fe->left[sl], c.max_points, c.radius, c.grid_rows, c.grid_cols, c.cell_size, c.sigma_mask, c.min_response));
        FE_TRY(fe, fe->ctx, slam_kpset_keyframe(fe->ctx, fe->ks));
        if (stereo_params && tri) {
            FE_TRY(fe, fe->ctx, slam_ctx_wait_event(fe->ctx, fe->rbuilt[sr]));
            FE_TRY(fe, fe->ctx, slam_kpset_stereo_match(fe->ctx, fe->ks, fe->left[sl], fe->right[sr], stereo_params, stereo_prior, c.pyramid_levels, c.pyramid_levels_3d,
                                                        c.window, c.iterations, c.eig_thr, c.eps, c.max_distance, c.epipolar_error, 0));
            FE_TRY(fe, fe->ctx, slam_kpset_triangulate(fe->ctx, fe->ks, tri, tri + 16, tri + 32, tri + 48, tri + 52, tri + 56, c.max_error, c.min_depth, 0));
        }
    }
    FE_TRY(fe, fe->ctx, slam_event_record(fe->ctx, fe->tracked));
    fe->us_enq += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0t).count();
    return SLAM_OK;
}

// ... and the frame's list length: the call's one device -> host copy (synchronises the tracking stream)
int fe_finish(slam_frontend *fe, int32_t *frame_out, int32_t *count_out)
{
    const long t = fe->done;
    const auto h1 = std::chrono::steady_clock::now();
    int32_t cnt = 0;
    FE_TRY(fe, fe->ctx, slam_kpset_counts(fe->ctx, fe->ks, &cnt));   // synchronises the tracking stream
    fe->us_wait += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h1).count();
    fe->n_bound = cnt;
    fe->done++;
    if (frame_out) *frame_out = (int32_t)t;
    if (count_out) *count_out = cnt;
    return SLAM_OK;
}
}  // namespace

extern "C" {

int slam_frontend_create(int device, const slam_frontend_config *cfg, slam_frontend **out)
{
    if (!cfg || !out) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_frontend_create: NULL argument");
    if (cfg->H < 16 || cfg->W < 16 || cfg->pyramid_levels < 0 || cfg->cap < cfg->max_points + cfg->grid_rows * cfg->grid_cols || (cfg->pyr_mode != 1 && cfg->pyr_mode != 3) || !(cfg->pyr_sigma > 0))
        return slam_fail(nullptr, SLAM_ERR_ARG, "slam_frontend_create: bad configuration (cap must hold max_points + grid_rows x grid_cols keypoints, pyr_mode 1 or 3)");
    slam_frontend *fe = new slam_frontend();
    fe->c = *cfg; fe->npix = (size_t)cfg->H * cfg->W;
    // every stream in the DEFAULT scheduling class: once a stream of another class exists in the process the runtime maps the default-class
    // queues differently and a latency-bound single stream loses ~40 % (measured twice: bench.py's single-stream legs, and this entry with its
    // tracking stream in the low class: 1 536 instead of 2 5xx frames/s).  Without look-ahead the build runs on the tracking stream itself:
    // nothing to overlap, and every cross-queue dependency costs ~10 us.
    int rc = slam_ctx_create(device, &fe->ctx);
    if (!rc && cfg->lookahead) rc = slam_ctx_create(device, &fe->ctx_build);
    if (!rc && !cfg->lookahead) fe->ctx_build = fe->ctx;
    if (!rc) rc = slam_ctx_create(device, &fe->ctx_right);
    for (int k = 0; k < 3 && !rc; k++) { rc = slam_pyr_create(fe->ctx_build, cfg->H, cfg->W, cfg->pyramid_levels, &fe->left[k]); if (!rc) rc = slam_event_create(fe->ctx_build, &fe->built[k]); }
    for (int k = 0; k < 2 && !rc; k++) { rc = slam_pyr_create(fe->ctx_right, cfg->H, cfg->W, cfg->pyramid_levels, &fe->right[k]); if (!rc) rc = slam_event_create(fe->ctx_right, &fe->rbuilt[k]); }
    if (!rc) rc = slam_kpset_create(fe->ctx, 1, cfg->cap, &fe->ks);
    if (!rc && hipHostMalloc((void **)&fe->pin, 5 * fe->npix, hipHostMallocDefault) != hipSuccess) rc = slam_fail(nullptr, SLAM_ERR_HIP, "slam_frontend_create: pinned staging");
    if (!rc && hipMalloc((void **)&fe->dev8, 5 * fe->npix) != hipSuccess) rc = slam_fail(nullptr, SLAM_ERR_HIP, "slam_frontend_create: device staging");
    if (rc) { slam_frontend_destroy(fe); return rc; }
    *out = fe;
    return SLAM_OK;
}

int slam_frontend_destroy(slam_frontend *fe)
{
    if (!fe) return SLAM_OK;
    if (getenv("SLAMHIP_FE_HOSTTIME") && fe->ncalls) fprintf(stderr, "slam_frontend host time per call: feed %.1f us, enqueue of the frame's work %.1f us, wait for the list length %.1f us (%ld calls)\n", fe->us_feed / fe->ncalls, fe->us_enq / fe->ncalls, fe->us_wait / fe->ncalls, fe->ncalls);
    if (fe->ctx) (void)slam_ctx_synchronize(fe->ctx);
    if (fe->ctx_build && fe->ctx_build != fe->ctx) (void)slam_ctx_synchronize(fe->ctx_build);
    if (fe->ctx_right) (void)slam_ctx_synchronize(fe->ctx_right);
    if (fe->ks) slam_kpset_destroy(fe->ks);
    for (int k = 0; k < 3; k++) { if (fe->left[k]) slam_pyr_destroy(fe->left[k]); if (fe->built[k]) slam_event_destroy(fe->built[k]); }
    for (int k = 0; k < 2; k++) { if (fe->right[k]) slam_pyr_destroy(fe->right[k]); if (fe->rbuilt[k]) slam_event_destroy(fe->rbuilt[k]); }
    if (fe->tracked) slam_event_destroy(fe->tracked);
    if (fe->pin) (void)hipHostFree(fe->pin);
    if (fe->dev8) (void)hipFree(fe->dev8);
    if (fe->ctx_right) slam_ctx_destroy(fe->ctx_right);
    if (fe->ctx_build && fe->ctx_build != fe->ctx) slam_ctx_destroy(fe->ctx_build);
    if (fe->ctx) slam_ctx_destroy(fe->ctx);
    delete fe;
    return SLAM_OK;
}

int slam_frontend_step(slam_frontend *fe, const uint8_t *left_u8, const uint8_t *right_u8, const double *params, int prior,
                       const double *stereo_params, int stereo_prior, const double *tri, const uint8_t *cull_flags_dev,
                       int32_t *frame_out, int32_t *count_out)
{
    if (!fe || !left_u8) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_frontend_step: NULL argument");
    if (!fe->tracked) { const int rc = slam_event_create(fe->ctx, &fe->tracked); if (rc) return fe_fail(fe, rc, fe->ctx, "slam_event_create"); (void)slam_event_record(fe->ctx, fe->tracked); }
    // look-ahead: the new frame's copy and build are enqueued FIRST, then the work of the frame whose build ran during the call before, then the
    // read-back.  (The other order -- tracking first, so that it runs during the 80-150 us of host time the build graph's launch takes -- is SLOWER:
    // 3 610 instead of 5 250 frames/s in tolerance mode, 2 060 instead of 2 470 bit-exact; also with the read-back requested ahead of the
    // feed; the runtime serves the graph's packets first.  The tracking stream in another scheduling class, low or high, costs more still: 1 550-2 200
    // frames/s.  Neither variant is kept in the code.)
    const auto h0 = std::chrono::steady_clock::now();
    int rc = fe_feed(fe, left_u8, right_u8);
    if (rc) return rc;
    fe->us_feed += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count(); fe->ncalls++;
    if (fe->c.lookahead && fe->fed - fe->done < 2) { if (frame_out) *frame_out = -1; if (count_out) *count_out = 0; return SLAM_OK; }
    rc = fe_track(fe, params, prior, stereo_params, stereo_prior, tri, cull_flags_dev);
    if (rc) return rc;
    return fe_finish(fe, frame_out, count_out);
}

int slam_frontend_flush(slam_frontend *fe, const double *params, int prior, const double *stereo_params, int stereo_prior,
                        const double *tri, const uint8_t *cull_flags_dev, int32_t *frame_out, int32_t *count_out)
{
    if (!fe) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_frontend_flush: NULL argument");
    if (fe->fed == fe->done) { if (frame_out) *frame_out = -1; if (count_out) *count_out = fe->n_bound; return SLAM_OK; }
    const int rc = fe_track(fe, params, prior, stereo_params, stereo_prior, tri, cull_flags_dev);
    return rc ? rc : fe_finish(fe, frame_out, count_out);
}

slam_kpset *slam_frontend_keypoints(slam_frontend *fe) { return fe ? fe->ks : nullptr; }
slam_ctx *slam_frontend_ctx(slam_frontend *fe) { return fe ? fe->ctx : nullptr; }
const char *slam_frontend_last_error(slam_frontend *fe) { return fe ? fe->err.c_str() : "slam_frontend: NULL handle"; }

}  // extern "C"
