/*
 * slamhip.h -- C ABI of libslamhip.so: the MI355X (gfx950) implementation of
 * SLAM.jl's data-parallel hot path (extractor -> LK pyramid -> forward-backward
 * Lucas-Kanade -> local bundle adjustment).
 *
 * The reference (pxl-th/SLAM.jl) is pure Julia and has no FFI: the boundary is a
 * set of Julia function seams.  Each entry point below replaces one seam; the
 * Julia-side `ccall` shim that rebinds the seams is slam.jl_amd/julia/SLAMHip.jl
 * (see INTEGRATION.md).  Reference file:line cited per function are relative to
 * the reference repository root.
 *
 * Conventions (identical to the reference so the shim is copy-free):
 *   - Float64 everywhere; images are column-major H x W (Julia `Matrix`, y
 *     fastest), values in [0,1];
 *   - points are (y, x) Float64 pairs, 1-based pixel coordinates; keypoint
 *     indices are (row, col) Int64 pairs, 1-based; ids are Int64, 1-based;
 *   - the caller owns every host buffer; the library copies in/out before
 *     returning and never retains host pointers.  Pointers named `*_dev` are
 *     device (HBM) pointers owned by the caller;
 *   - every call returns 0 on success, a negative slam_status on failure;
 *     slam_last_error(ctx) gives the message.  No C++ exception crosses the ABI;
 *   - one slam_ctx per calling task/thread (it owns a HIP stream and scratch);
 *     calls on one ctx are synchronous unless the name ends in `_async` or the
 *     call has a `sync` argument / is documented as enqueue-only;
 *     pyramid handles may be READ by several contexts concurrently.  A pyramid (or
 *     a pyramid batch: its members share checkpoint scratch, the blur scratch
 *     plane and the cached hipGraph of the build) is BUILT through one stream at a
 *     time: enqueue its updates from one context, or order the contexts with
 *     slam_ctx_wait_for / slam_event_* before switching -- concurrent builds of
 *     one pyramid or of two members of one batch race on that scratch.
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slam_ctx slam_ctx;
typedef struct slam_pyr slam_pyr;   /* device-resident LKPyramid */
typedef struct slam_ba  slam_ba;    /* device-resident sharded-BA state */

enum slam_status {
    SLAM_OK = 0,
    SLAM_ERR_ARG = -1,          /* bad argument */
    SLAM_ERR_HIP = -2,          /* HIP runtime error (message has hipGetErrorString) */
    SLAM_ERR_LAYERS = -3,       /* "Not enough layers in pyramids." lucas_kanade.jl:15 */
    SLAM_ERR_CAPACITY = -4,     /* output buffer too small */
    SLAM_ERR_NUMERIC = -5       /* reduced camera system not positive definite */
};

/* ---- context --------------------------------------------------------------- */
int  slam_ctx_create(int device, slam_ctx **out);
/* the same, with the context's stream restricted to the compute units set in cu_mask (bit i of word i / 32 = CU i):
 * partitions the chip between stages that run concurrently on different contexts */
int  slam_ctx_create_cumask(int device, const uint32_t *cu_mask, int n_words, slam_ctx **out);
/* the same, with the stream in its own scheduling class (priority > 0 high, < 0 low): its own hardware queue, which a
 * replayed hipGraph's branches never share -- for the latency-critical stage of a multi-context pipeline */
int  slam_ctx_create_priority(int device, int priority, slam_ctx **out);
/* waits for the context's stream, frees its scratch; the STREAM itself is parked for the next context of the same device and scheduling
 * class instead of being destroyed -- events another library recorded on it (PyTorch's pinned-memory allocator does, for copies issued on
 * the stream) stay valid for the life of the process */
int  slam_ctx_destroy(slam_ctx *ctx);
int  slam_ctx_synchronize(slam_ctx *ctx);
/* Device-side ordering between two contexts of one device: work enqueued on `ctx`
 * after this call waits for everything enqueued on `other` so far (no host block).
 * Lets a caller overlap asynchronous calls on two contexts the way the reference
 * overlaps its front-end and mapper tasks (SLAM.jl:166, mapper.jl:37-66), e.g.
 * slam_pyr_update_dev(other, next frame, sync=0) while `ctx` tracks the current one. */
int  slam_ctx_wait_for(slam_ctx *ctx, slam_ctx *other);
/* Markers for deeper pipelines: slam_event_record(ctx, e) marks what `ctx` has enqueued so far; slam_ctx_wait_event(ctx2, e)
 * makes later work of ctx2 wait (on the device) for that point, even after more work has been enqueued on `ctx`
 * (e.g. the pyramid builds of frames t+1 and t+2 are both in flight while tracking waits for t+1 only). */
typedef struct slam_event slam_event;
int  slam_event_create(slam_ctx *ctx, slam_event **out);
int  slam_event_record(slam_ctx *ctx, slam_event *e);
int  slam_ctx_wait_event(slam_ctx *ctx, slam_event *e);
int  slam_event_destroy(slam_event *e);
/* the same with timing enabled: a pair brackets a stage on a context's stream; slam_event_elapsed_ms waits for `b` */
int  slam_event_create_timed(slam_ctx *ctx, slam_event **out);
int  slam_event_elapsed_ms(slam_event *a, slam_event *b, double *ms);
/* the ctx's hipStream_t (as void*), for callers that record HIP events on it */
void *slam_ctx_stream(slam_ctx *ctx);
/* message of the last failing call on ctx (ctx == NULL: last ctx-less failure) */
const char *slam_last_error(slam_ctx *ctx);
/* library version / build arch string, e.g. "slamhip 0.1 gfx950" */
const char *slam_version(void);
/* Optional device-side timing: when enabled, named spans (e.g. "pyr_update",
 * "k_iir_rows", "fb_track", "detect", "local_ba") are bracketed with hipEvents
 * on the ctx stream; slam_prof_get synchronises and returns the accumulated
 * device time and span count.  The reference's equivalent is its @debug
 * time() prints (front_end.jl:82-114, mapper.jl:50-132, estimator.jl:90-106). */
int  slam_prof_enable(slam_ctx *ctx, int on);
int  slam_prof_reset(slam_ctx *ctx);
int  slam_prof_get(slam_ctx *ctx, const char *name, double *total_ms, int64_t *count);

/* ---- Extractor -------------------------------------------------------------- */
/* detect(e::Extractor, image, current_points; sigma_mask) -- src/extractor.jl:63-95
 * (+ get_mask :116-122, _shi_tomasi :24-42).  Extractor fields (extractor.jl:7-20,
 * built at SLAM.jl:149-160) are passed explicitly.  out_rc receives (row, col)
 * pairs in the reference's order (cells row-major, column-major inside a cell);
 * cap = capacity in keypoints; n_out may exceed max_points (as in the reference).
 * Returns n_out = 0 when n_cur >= max_points (extractor.jl:64). */
int slam_detect(slam_ctx *ctx, const double *image, int H, int W,
                const double *cur_yx, int n_cur,
                int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                double sigma_mask, double min_response,
                int64_t *out_rc, int cap, int *n_out);
/* Same, on the image already resident as layer 1 of a device pyramid (the frame
 * passed to create_keyframe! is the one the current pyramid was built from:
 * front_end.jl:461-464, map_manager.jl:98-105).  No image upload. */
int slam_detect_pyr(slam_ctx *ctx, const slam_pyr *pyr,
                    const double *cur_yx, int n_cur,
                    int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                    double sigma_mask, double min_response,
                    int64_t *out_rc, int cap, int *n_out);
/* detect() for the S members of a pyramid batch (slam_pyr_create_batch) in one launch:
 * cur_yx holds the current keypoints of all streams back to back, stream s owning
 * [cur_off[s], cur_off[s+1]) (cur_off[0] = 0); the new keypoints of stream s are
 * out_rc[out_off[s] .. out_off[s+1]).  cap = total capacity in keypoints.  Per stream
 * identical to slam_detect_pyr on that stream's pyramid (map_manager.jl:105 called for S
 * independent streams). */
int slam_detect_batch(slam_ctx *ctx, const slam_pyr *pyr0, int S,
                      const double *cur_yx, const int32_t *cur_off,
                      int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                      double sigma_mask, double min_response,
                      int64_t *out_rc, int cap, int32_t *out_off);

/* describe(e, image, keypoints) -> create_descriptor(img, kps, BRIEF) --
 * src/extractor.jl:103-105.  pattern: n_bits x 4 int32 (dy1, dx1, dy2, dx2)
 * sampling table supplied by the caller (ImageFeatures draws it from Julia's
 * seeded RNG, which cannot be reproduced outside Julia).  n_bits % 64 == 0.
 * out_bits: n x (n_bits/64) uint64; out_rc: surviving keypoints (border-dropped
 * ones removed, order kept). */
int slam_describe(slam_ctx *ctx, const double *image, int H, int W,
                  const int64_t *rc, int n, const int32_t *pattern, int n_bits,
                  double sigma, int window,
                  uint64_t *out_bits, int64_t *out_rc, int *n_out);

/* ---- LKPyramid -------------------------------------------------------------- */
/* LKPyramid(image, levels; sigma, reusable=true) / update!(lk, img) / copy! /
 * deepcopy -- src/optical_flow/pyramid.jl:16-137, lucas_kanade.jl:102-138.
 * pyramid_levels = levels above the base (Params.pyramid_levels = 3 -> 4 layers). */
int slam_pyr_create(slam_ctx *ctx, int H, int W, int pyramid_levels, slam_pyr **out);
int slam_pyr_destroy(slam_pyr *pyr);
/* Update mode flag (OR it into `mode` = 1 or 3): the pyramid will only be the TARGET of matches -- the mapper's right pyramid
 * (mapper.jl:51-56, optical_flow_matching!(..., left, right, true)).  optflow! samples a target's layers at every level and
 * fb_tracking!'s backward pass (pyramid_levels = 0, tracker.jl:51-57) uses it as template at level 1 only, so the gradient,
 * product and integral planes of the coarser levels are never read: level 0 is built in full, the other levels get their layers
 * only (about a quarter of the build).  Passing such a pyramid as the SOURCE of a match is an error (SLAM_ERR_ARG). */
#define SLAM_PYR_TARGET_ONLY 16
/* Update mode flag: replay the build as ONE chain on the calling context's stream (no forked integral-image branch).  One build
 * alone takes ~15 % longer (441 vs 376 us for a 370 x 1226 image), but it occupies one hardware queue instead of two and costs the
 * host a third per call (31 vs 84 us): the choice for a host that keeps the builds of several consecutive frames in flight on several
 * contexts (preprocess! of frames t+1 .. t+3 while frame t is tracked: 170 us per build with three in flight). */
#define SLAM_PYR_CHAIN 32
/* mode 0: constructor semantics (pyramid.jl:40-79: NA() blur, Fill(0) Scharr);
 * mode 1: update! semantics (pyramid.jl:81-137: replicate borders);
 * mode 3: update! semantics with SEGMENTED recurrences: each IIR / cumulative-sum
 *         line is cut into <= 16 segments whose entry states are obtained from
 *         zero-state end states and powers of the filter's companion matrix.  Same
 *         arithmetic inside a segment, different rounding of the entry states:
 *         planes agree with mode 1 to ~1e-13 relative (mode 1 is bit-exact against
 *         the sequential reference order and is what the parity tests pin). */
int slam_pyr_update(slam_ctx *ctx, slam_pyr *pyr, const double *image, int mode, double sigma);
/* image already in HBM (column-major f64); returns after enqueueing when sync == 0 */
int slam_pyr_update_dev(slam_ctx *ctx, slam_pyr *pyr, const double *image_dev, int mode, double sigma, int sync);
/* Same, from the 8-bit image the KITTI reader decodes (example/kitty/kitty.jl:52-102) before
 * `Gray{Float64}.(frame)` (example/kitty/main.jl:39-41): the conversion raw/255 runs on the device,
 * so the host link carries 1 byte per pixel instead of 8. */
int slam_pyr_update_u8(slam_ctx *ctx, slam_pyr *pyr, const uint8_t *image_u8, int mode, double sigma);
/* the same from an 8-bit image already resident in HBM; returns after enqueueing when sync == 0 */
int slam_pyr_update_u8_dev(slam_ctx *ctx, slam_pyr *pyr, const uint8_t *image_u8_dev, int mode, double sigma, int sync);
/* copy!(dst, src) pyramid.jl:28-38 (same shape required) */
int slam_pyr_copy(slam_ctx *ctx, slam_pyr *dst, const slam_pyr *src);
/* deepcopy(lk) SLAM.jl:218 */
int slam_pyr_clone(slam_ctx *ctx, const slam_pyr *src, slam_pyr **out);
/* introspection for parity tests: plane 0..5 = layers, Iy, Ix, Iyy, Ixx, Iyx;
 * level 0-based.  out must hold H_l * W_l doubles. */
int slam_pyr_shape(const slam_pyr *pyr, int level, int *H, int *W);
int slam_pyr_levels(const slam_pyr *pyr);
int slam_pyr_download(slam_ctx *ctx, const slam_pyr *pyr, int plane, int level, double *out);

/* ---- forward-backward Lucas-Kanade ----------------------------------------- */
/* fb_tracking!(prev, cur, keypoints; displacement, iterations, window_size,
 * pyramid_levels, max_distance) -- src/tracker.jl:17-82 over optflow!
 * src/optical_flow/lucas_kanade.jl:9-100.  disp0_yx may be NULL (zeros);
 * it is given in coarsest-level pixels like the reference's `displacement`.
 * out_yx[i] is defined only where status[i] != 0.  n == 0 is a no-op. */
int slam_fb_track(slam_ctx *ctx, const slam_pyr *prev, const slam_pyr *cur,
                  const double *pts_yx, const double *disp0_yx, int n,
                  int pyramid_levels, int window, int iterations,
                  double eig_thr, double eps, double max_distance,
                  double *out_yx, uint8_t *status);

/* The array-level body of optical_flow_matching!(map_manager, frame, from, to, stereo)
 * -- src/map_manager.jl:451-564 -- in one call: keypoints flagged is_3d are first
 * tracked with the prior displacement (proj - pt) / 2^pyramid_levels_3d on
 * pyramid_levels_3d levels (:494,504,517-521); the ones that fail join the 2-D
 * keypoints and are tracked without prior on pyramid_levels levels (:533-552).
 * Results are identical to issuing the two slam_fb_track calls of the reference
 * protocol (points are independent).  proj_yx is read only where is_3d != 0. */
int slam_flow_match(slam_ctx *ctx, const slam_pyr *from, const slam_pyr *to,
                    const double *pts_yx, const uint8_t *is_3d, const double *proj_yx, int n,
                    int pyramid_levels, int pyramid_levels_3d, int window, int iterations,
                    double eig_thr, double eps, double max_distance,
                    double *out_yx, uint8_t *status);

/* ---- lock-stepped batches of streams --------------------------------------- */
/* The recurrences of the pyramid build are latency-bound for one image; S independent images
 * (left + right of a key-frame, or S camera streams) share every launch when their pyramids live in one batch.
 * slam_pyr_create_batch fills out[0..S) (1 <= S <= 128) with ordinary pyramid handles backed by one allocation; each can be used
 * with every single-pyramid call above.  slam_pyr_update_batch_dev rebuilds all S in one launch set (pyrs must
 * be the S members of one batch, in order; images already in HBM); slam_flow_match_batch tracks the keypoints
 * of all S streams in one launch (img_index[i] = stream of point i; from0 / to0 = member 0 of two batches).
 * Batch updates: mode 1 or 3 as for slam_pyr_update.
 *   mode 1 -- every plane bit-identical to mode 1 on a single pyramid (the parity reference and the default); launches with >= 40 MB
 *     of plane data use the checkpointed IIR kernels (forward state every 32 samples, forward values recomputed: 2 reads + 1 write
 *     per sample instead of 2 + 2) and the fused dim-1 stage, still bit-identical.
 *   mode 3 with S >= 4 (round 4) -- the TOLERANCE-mode batch build: per level one dim-1 kernel that leaves the product planes as
 *     suffix sums along y and one dim-2 kernel that finishes filter, running sum and imresize! with one read and one write per plane
 *     (2.2x instead of 3.9x the algorithmic bytes).  Every plane within 1e-11 of the mode-1 plane relative to the
 *     plane's largest magnitude; tracked positions within 1e-6 px.  Keypoint INDICES are not affected: slam_detect* work on the raw
 *     frame / the base layer, which is the same bytes in both modes.  Levels too small for those kernels (and S < 4: the segmented
 *     single-image kernels of slam_pyr_update's mode 3) fall back per level; a batch may be updated in either mode at any time.
 *     NOTE: in this mode the ARITHMETIC is no longer the reference's operation sequence -- the dim-2 kernel contracts its multiply-adds
 *     into FMAs and evaluates the recurrences through affine maps of row segments, the dim-1 kernel leaves suffix sums instead of
 *     filtered values -- only the RESULT is held to the reference's (1e-11 relative).  Matches between two pyramids last built in
 *     mode 3 (slam_kpset_flow_match / _stereo_match) likewise take a tracking kernel with contracted arithmetic (two-operation
 *     lerps, FMA accumulation; positions within 1e-6 px of the sequential-order oracle, csrc/lk.hip TOL). */
int slam_pyr_create_batch(slam_ctx *ctx, int H, int W, int pyramid_levels, int S, slam_pyr **out);
int slam_pyr_update_batch_dev(slam_ctx *ctx, slam_pyr *const *pyrs, const double *const *images_dev, int S,
                              int mode, double sigma, int sync);
/* Same from 8-bit frames resident in HBM (column-major H x W bytes, the KITTI reader's decode before
 * `Gray{Float64}.(frame)`, example/kitty/main.jl:39-41): raw / 255 on the device. */
int slam_pyr_update_batch_u8_dev(slam_ctx *ctx, slam_pyr *const *pyrs, const uint8_t *const *images_u8_dev, int S,
                                 int mode, double sigma, int sync);
int slam_flow_match_batch(slam_ctx *ctx, const slam_pyr *from0, const slam_pyr *to0, int S, const int32_t *img_index,
                          const double *pts_yx, const uint8_t *is_3d, const double *proj_yx, int n,
                          int pyramid_levels, int pyramid_levels_3d, int window, int iterations,
                          double eig_thr, double eps, double max_distance, double *out_yx, uint8_t *status);
/* slam_flow_match_batch followed by the list surgery of optical_flow_matching! (map_manager.jl:523-560): the keypoints
 * whose tracking succeeded, in input order, with their new positions (kept_yx), 3-D flags, stream indices and input
 * indices (kept_src); arrays sized for n entries.  status (n, nullable) reports every input keypoint. */
int slam_flow_match_batch_kept(slam_ctx *ctx, const slam_pyr *from0, const slam_pyr *to0, int S, const int32_t *img_index,
                               const double *pts_yx, const uint8_t *is_3d, const double *proj_yx, int n,
                               int pyramid_levels, int pyramid_levels_3d, int window, int iterations,
                               double eig_thr, double eps, double max_distance,
                               double *kept_yx, uint8_t *kept_is3d, int32_t *kept_img, int32_t *kept_src, int *n_kept,
                               uint8_t *status);

/* ---- device-resident keypoint lists --------------------------------------------------------------
 * The reference keeps a frame's keypoints in a Dict and optical_flow_matching! (map_manager.jl:451-564) copies them into
 * arrays per call; here the lists of S lock-stepped streams live in HBM: stream s owns slots [s cap, s cap + count[s]) of
 * every per-keypoint array (pixel, is_3d, map point, id, stereo pixel / flag).  Tracking, removal of lost keypoints, map
 * culling, the avoidance list and merge of key-frame detection, stereo matching and triangulation read and write those
 * arrays; compaction is stable and runs on the device (wave ballot + prefix counts).  Every call except create / upload /
 * download / counts returns after ENQUEUEING on ctx's stream; slam_kpset_counts is the one small device -> host copy a step
 * needs.  n_bound: an upper bound of the number of live keypoints known to the host (sizes launches; <= 0: S x cap).
 * Per-stream call parameters `params`: S x 32 doubles, [0..15] Tcw of the target camera (column-major 4x4), [16..19] fx fy cx
 * cy and [20..23] k1 k2 p1 p2 of the target camera, [24..25] prior shift (y, x); prior = 0: none, 1: projection of the map
 * point through Tcw and the lens model (project_world_to_image_distort, frame.jl:478-484), 2: pixel + shift. */
typedef struct slam_kpset slam_kpset;
int slam_kpset_create(slam_ctx *ctx, int S, int cap, slam_kpset **out);
int slam_kpset_destroy(slam_kpset *ks);
int slam_kpset_streams(const slam_kpset *ks);
int slam_kpset_capacity(const slam_kpset *ks);
/* replace / read stream s's list (initialisation, tests, host consumers); NULL arrays are skipped; ids == NULL: 0 .. n-1 */
int slam_kpset_upload(slam_ctx *ctx, slam_kpset *ks, int s, const double *yx, const uint8_t *is_3d, const double *xyz,
                      const int64_t *ids, int n);
int slam_kpset_download(slam_ctx *ctx, slam_kpset *ks, int s, double *yx, uint8_t *is_3d, double *xyz, int64_t *ids,
                        double *stereo_yx, uint8_t *has_stereo, int cap_out, int *n_out);
int slam_kpset_counts(slam_ctx *ctx, slam_kpset *ks, int32_t *counts /* S */);
/* optical_flow_matching!(map_manager, frame, from, to, false): 3-D keypoints first with their prior on pyramid_levels_3d
 * levels, failures and 2-D keypoints without prior on pyramid_levels levels (map_manager.jl:517-552); 3-D keypoints whose
 * projection is outside the image are left as they are (:501-506); lost keypoints are removed (:559), the others take
 * their new position.  from0 / to0: member 0 of two pyramid batches with >= S members. */
int slam_kpset_flow_match(slam_ctx *ctx, slam_kpset *ks, const slam_pyr *from0, const slam_pyr *to0, const double *params, int prior,
                          int pyramid_levels, int pyramid_levels_3d, int window, int iterations, double eig_thr, double eps,
                          double max_distance, int n_bound);
/* optical_flow_matching!(..., true) left0 -> right0 (params: the RIGHT camera): a match passes maybe_stereo_update!
 * (map_manager.jl:579-590: |left row - undistorted right row| <= epipolar_error; the left row is kept) and is stored as the
 * keypoint's stereo pixel; 3-D keypoints projected outside the right image are removed (:491-498).  The left camera is taken as
 * rectified (its undistorted row = the pixel row). */
int slam_kpset_stereo_match(slam_ctx *ctx, slam_kpset *ks, const slam_pyr *left0, const slam_pyr *right0, const double *params, int prior,
                            int pyramid_levels, int pyramid_levels_3d, int window, int iterations, double eig_thr, double eps,
                            double max_distance, double epipolar_error, int n_bound);
/* remove the keypoints whose flag is set (flags_dev: S x cap bytes in HBM, slot order): map culling, outliers of the pose
 * estimators (estimator.jl:283-292, front_end.jl:174-219) */
int slam_kpset_remove(slam_ctx *ctx, slam_kpset *ks, const uint8_t *flags_dev);
/* extract_keypoints! (map_manager.jl:98-113) for every stream: detect() with the stream's own list as avoidance list; the
 * new keypoints are appended (is_3d = 0, fresh ids).  Needs cap >= max_points + grid_rows x grid_cols. */
int slam_kpset_detect(slam_ctx *ctx, slam_kpset *ks, const slam_pyr *pyr0, int max_points, int radius, int grid_rows, int grid_cols,
                      int cell_size, double sigma_mask, double min_response);
/* triangulate_stereo! (mapper.jl:142-183) for every 2-D keypoint with a stereo match: success -> map point Twc[s] X and
 * is_3d = 1, failure -> the stereo observation is dropped.  P1, P2, T21, cam1, cam2 as slam_triangulate; Twc: S x 16.
 * The left and right pixels are used as stored: the set's stereo path is for RECTIFIED, zero-distortion pairs (KITTI, the reference's
 * stereo example), where kp.undistorted_pixel == kp.pixel (mapper.jl:162-163); for cameras with lens distortion undistort on the
 * host and use slam_triangulate, or leave the stereo seams on the host lists. */
int slam_kpset_triangulate(slam_ctx *ctx, slam_kpset *ks, const double *P1, const double *P2, const double *T21,
                           const double *cam1, const double *cam2, const double *Twc, double max_error, double min_depth, int n_bound);

/* compute_pose! (front_end.jl:132-219) for every stream, on the set: the is_3d keypoints in list order (:139-160; undistorted
 * pixel and normalised bearing from params[s][16..23] = fx fy cx cy k1 k2 p1 p2) -> P3P RANSAC (threshold = max_reprojection_error,
 * `iters` triples per stream from a counter-based generator seeded with `seed`: three distinct indices,
 * splitmix64(seed ^ s << 48 ^ iteration << 16 ^ attempt) mod n, restated in keypoint_set.py) -> its outliers leave the list
 * (:187-189) -> pnp_bundle_adjustment of the inliers from the P3P pose (:203-206) -> its outliers leave the list (:213-215).
 * status[s] = 1: poses_cw[16 s ..] is the refined world -> camera transform (column-major 4 x 4); 0: one of the reference's
 * reset exits (fewer than 5 3-D keypoints :133, fewer than 5 P3P inliers :179, refinement left fewer than 5 inliers or a larger
 * error :207; in the last case the P3P outliers are already gone, as in the reference), pose = identity.  n_inliers (P3P) and
 * counts (list lengths after the removals) may be NULL.  Synchronous: the call is the step's one device -> host copy. */
int slam_kpset_compute_pose(slam_ctx *ctx, slam_kpset *ks, const double *params, double threshold, int iters, uint64_t seed,
                            int pnp_iters_fast, int pnp_iterations, double depth_eps, double repr_eps,
                            double *poses_cw, int32_t *status, int32_t *n_inliers, int32_t *counts);

/* create_keyframe! (map_manager.jl:60-96) for the lists: the current frame becomes the previous key-frame of every keypoint it
 * holds (call it after slam_kpset_detect: the key-frame contains the new keypoints).  The observation travels with the keypoint
 * through every compaction; keypoints detected later have none until the next call. */
int slam_kpset_keyframe(slam_ctx *ctx, slam_kpset *ks);
/* triangulate_temporal! (mapper.jl:185-262) for every 2-D keypoint whose first observer -- the key-frame that detected it; the set
 * keeps that observation and the key-frame's id (a per-stream counter advanced by slam_kpset_keyframe) beside the keypoint -- is an
 * earlier key-frame than the frame's (kf_cur[s]) and still in the table (>= kf_lo[s]).  tab: S x nkf x 64 doubles, entry
 * [s][kf % nkf] = { P2 = K * rel_pose_inv, rel_pose_inv, rel_pose = observer.cw * frame.wc, observer.wc }, column-major 4 x 4 each
 * (what mapper.jl:226-231 computes once per observer).  Success: map point = observer.wc * X, is_3d = 1; a gate fails with the
 * rotation-compensated parallax above min_parallax: the observation is removed (:244-258); otherwise the point is accepted. */
int slam_kpset_triangulate_temporal(slam_ctx *ctx, slam_kpset *ks, const double *params, const double *tab, int nkf,
                                    const int32_t *kf_cur, const int32_t *kf_lo, double max_error, double min_depth, double min_parallax,
                                    int n_bound);
/* the first observations of stream s's list, host <-> device (restoring state, tests): first_yx n x 2, first_kf n ids, and the
 * stream's key-frame counter */
int slam_kpset_upload_first(slam_ctx *ctx, slam_kpset *ks, int s, const double *first_yx, const int32_t *first_kf, int n, int kf_count);
int slam_kpset_download_first(slam_ctx *ctx, slam_kpset *ks, int s, double *first_yx, int32_t *first_kf, int cap_out, int *n_out, int *kf_count);
/* the key-frame observations of stream s's list, host <-> device (restoring state, tests): kyx n x 2 (y, x), has_kf n flags */
int slam_kpset_upload_keyframe(slam_ctx *ctx, slam_kpset *ks, int s, const double *kyx, const uint8_t *has_kf, int n);
int slam_kpset_download_keyframe(slam_ctx *ctx, slam_kpset *ks, int s, double *kyx, uint8_t *has_kf, int cap_out, int *n_out);
/* compute_pose_5pt! (front_end.jl:242-332, run every frame at :105) for every stream, on the set: the keypoints the previous
 * key-frame also observes -> undistorted pixels and normalised coordinates of both views (:263-272), the rotation-compensated
 * average parallax (:277-281; params[s][0..8] = R_compensation, column-major; [16..23] camera and distortion) -> five-point
 * RANSAC (`iters` 5-tuples per stream from the generator of slam_kpset_compute_pose) -> its outliers leave the list (:310-318).
 * status[s] = 1: P[12 s ..] is [R | t] (column-major 3 x 4, key-frame -> frame, |t| = 1) of the best essential matrix; 0: one
 * of the `nothing` exits (fewer than 8 keypoints :243 / in the key-frame :283, parallax below min_parallax :290, fewer than 5
 * inliers :305), nothing removed.  The composition with the motion-model scale (:320-330) is the caller's.  n_inliers, parallax
 * (the average) and counts (list lengths after the removals) may be NULL.  Synchronous -- unless P and status are both NULL:
 * then the call only enqueues (the epipolar filter acts on the lists; compute_pose! right behind it brings the step's copy). */
int slam_kpset_compute_pose_5pt(slam_ctx *ctx, slam_kpset *ks, const double *params, double min_parallax, double max_repr_error,
                                int iters, uint64_t seed, double *P, int32_t *status, int32_t *n_inliers, double *parallax,
                                int32_t *counts);

/* ---- one live stream, one call per frame (round 6) -----------------------------------------------------------------------------------
 * What run!() does per frame (src/front_end.jl:58-113: preprocess! :454-470 = copy!(previous_pyramid, current_pyramid) + update!,
 * klt_tracking! -> optical_flow_matching! map_manager.jl:451-564; at a key-frame create_keyframe! -> extract_keypoints!
 * map_manager.jl:98-113 and the mapper's right pyramid + stereo matching + triangulate_stereo! mapper.jl:51-66, :142-183) as ONE entry:
 * the frame's bytes are copied to HBM, the single-image build graph, the 3-D / 2-D matching passes, the removal of lost keypoints and
 * (key-frame) detection, right build, stereo match and triangulation are ENQUEUED back to back -- no host code between them -- and the call
 * returns with the one number a front-end needs per frame, the length of the keypoint list.  The lists live in HBM (a slam_kpset with one
 * stream); three left pyramids rotate (copy! = a handle rotation).
 *   lookahead = 0: step(frame t) returns frame t's list: latency = build + match.
 *   lookahead = 1: step(frame t + 1) starts that frame's build on the build stream and returns frame t's list (whose build ran during the
 *                  call before): one frame of latency buys the overlap of build and matching -- a recorded sequence, or a camera whose
 *                  next frame has arrived while the current one is processed (the "next frame only" figure of bench.py).
 * config: plain scalars mirroring Params (params.jl:58-82) and the Extractor (extractor.jl:7-22).  Per-call: params / stereo_params =
 * 32 doubles each as for slam_kpset_flow_match / _stereo_match (prior as there); tri = P1 (16), P2 (16), T21 (16), cam1 (4), cam2 (4),
 * Twc (16) as for slam_kpset_triangulate (read at key-frames with a right image); cull_flags_dev (nullable): cap bytes in HBM, 1 = remove
 * before detection (map culling).  right_u8 != NULL marks the fed frame as a key-frame.  frame_out: index (0-based, feed order) of the frame
 * whose list length count_out reports, -1 while the pipeline fills (first call with lookahead = 1); slam_frontend_flush processes the
 * frame still in flight.  The per-frame arguments describe the frame that is PROCESSED by the call. */
typedef struct slam_frontend slam_frontend;
typedef struct slam_frontend_config {
    int32_t H, W, pyramid_levels, pyramid_levels_3d, window, iterations;
    int32_t max_points, radius, grid_rows, grid_cols, cell_size, cap;
    int32_t pyr_mode;            /* 1: bit-exact update!, 3: tolerance mode */
    int32_t lookahead;           /* 0 / 1 */
    int32_t right_target_only;   /* build the right pyramid with SLAM_PYR_TARGET_ONLY */
    int32_t reserved;
    double eig_thr, eps, max_distance, sigma_mask, min_response, epipolar_error, max_error, min_depth, pyr_sigma;
} slam_frontend_config;
int slam_frontend_create(int device, const slam_frontend_config *cfg, slam_frontend **out);
int slam_frontend_destroy(slam_frontend *fe);
int slam_frontend_step(slam_frontend *fe, const uint8_t *left_u8, const uint8_t *right_u8, const double *params, int prior,
                       const double *stereo_params, int stereo_prior, const double *tri, const uint8_t *cull_flags_dev,
                       int32_t *frame_out, int32_t *count_out);
int slam_frontend_flush(slam_frontend *fe, const double *params, int prior, const double *stereo_params, int stereo_prior,
                        const double *tri, const uint8_t *cull_flags_dev, int32_t *frame_out, int32_t *count_out);
/* the stream's keypoint set (slam_kpset_download ...) and its tracking context (owned by the front-end) */
slam_kpset *slam_frontend_keypoints(slam_frontend *fe);
slam_ctx   *slam_frontend_ctx(slam_frontend *fe);
const char *slam_frontend_last_error(slam_frontend *fe);

/* ---- bundle adjustment ------------------------------------------------------ */
/* Array-level body of triangulate_stereo! (parallax == NULL: every gate applies, src/mapper.jl:142-183) and
 * triangulate_temporal! (a gate removes the observation only when parallax[i] > min_parallax, :185-262), for n
 * keypoints in one launch: DLT triangulation from the two undistorted pixels (RecoverPose.triangulate, :162,242),
 * division by the 4th component, depth gates in both cameras (min_depth = 0.1), reprojection gates against
 * max_error in both images.  P1, P2 (projection matrices), T21 (camera 1 -> camera 2: right_camera.Ti0 or
 * inv(rel_pose)): 4x4 column-major as Julia SMatrix{4,4}; cam = (fx, fy, cx, cy); pixels (y, x).
 * out_xyz[3 i..]: the point in camera-1 coordinates (the caller applies project_camera_to_world);
 * status[i] = 1: update the map point, 0: remove the keypoint / observation. */
int slam_triangulate(slam_ctx *ctx, const double *P1, const double *P2, const double *T21,
                     const double *cam1, const double *cam2,
                     const double *px1_yx, const double *px2_yx, int n,
                     double max_error, double min_depth,
                     const double *parallax, double min_parallax,
                     double *out_xyz, uint8_t *status);

/* p3p_ransac(points, pixels, pdn, K; threshold) of compute_pose! (src/front_end.jl:164-167; result consumed at
 * :174-186: n_inliers, (KP, inliers, error)).  pts3d n x 3 map points, px_xy n x 2 undistorted pixels in (x, y)
 * order (front_end.jl:150-151), pdn n x 3 bearing vectors normalize(kp.position) (:149), K 3x3 column-major.
 * `samples`: iters x 3 point indices, 0-BASED, drawn by the caller (the reference's RNG stream cannot be
 * reproduced outside Julia; same convention as the BRIEF pattern) -- every triple is solved (Grunert P3P, up to
 * 4 poses) and scored on the GPU, one workgroup per triple; triples with a repeated or out-of-range index are
 * skipped.  Winner: most inliers (depth > 0 and reprojection error < threshold), ties to the lower iteration.
 * KP = K [R | t] (3x4 column-major; the reference recovers the pose as iK * KP, :182), Rt = [R | t] itself (may be
 * NULL), inliers n bytes, *error = sum of the inliers' reprojection errors (may be NULL), *best_iter (may be NULL).
 * *n_inliers = 0 (and zeros elsewhere) when no triple gave a pose -- the reference's `res === nothing` branch. */
int slam_p3p_ransac(slam_ctx *ctx, const double *pts3d, const double *px_xy, const double *pdn, int n,
                    const double *K, double threshold, const int32_t *samples, int iters,
                    double *KP, double *Rt, uint8_t *inliers, int *n_inliers, double *error, int *best_iter);

/* five_point_ransac(previous_points, current_points, previous_pd, current_pd, K, K, cache; max_repr_error) of
 * compute_pose_5pt! (src/front_end.jl:305-308; result consumed at :305-331: n_inliers, (_, P, inliers, _)).
 * px1/px2: n x 2 undistorted pixels (x, y) of the previous key-frame / the current frame (:271-272), pd1/pd2:
 * n x 2 normalised coordinates position[[1, 2]] (:273-274), K1/K2 3x3 column-major.  `samples`: iters x 5 point
 * indices, 0-BASED, drawn by the caller; every 5-tuple is solved (Nister's five-point algorithm, <= 10 essential
 * matrices; pose of each by Horn's closed form + cheirality on the five points) and scored on the GPU: a
 * correspondence is an inlier iff its DLT triangulation lies in front of both cameras and both reprojection
 * errors are < max_repr_error.  Winner: most inliers, ties to the earlier tuple.
 * E: 3x3 column-major (may be NULL), P = [R | t] 3x4 column-major, previous -> current, |t| = 1 (the caller
 * rescales t, front_end.jl:321-329), inliers n bytes, *error = sum of both errors over the inliers (may be
 * NULL), *best_iter (may be NULL).  *n_inliers = 0 when no tuple gave a pose. */
int slam_five_point_ransac(slam_ctx *ctx, const double *px1_xy, const double *px2_xy, const double *pd1_xy,
                           const double *pd2_xy, int n, const double *K1, const double *K2, double max_repr_error,
                           const int32_t *samples, int iters, double *E, double *P, uint8_t *inliers,
                           int *n_inliers, double *error, int *best_iter);

/* The pose seams for S lock-stepped streams (no reference counterpart, like the other *_batch entry points): S
 * independent problems in one set of launches, results identical to S single calls.  Problem z owns the elements
 * [offsets[z], offsets[z+1]) of the concatenated point arrays (offsets[0] = 0; offsets has S + 1 entries; empty
 * problems are allowed and report n_inliers = 0); per-problem matrices are stored back to back (K: S x 9, KP / Rt /
 * P: S x 12, E: S x 9, poses: S x 16, cams: S x 4 as fx, fy, cx, cy); samples are S x iters x 3 (x 5), indices
 * local to their problem. */
int slam_p3p_ransac_batch(slam_ctx *ctx, int S, const int32_t *offsets, const double *pts3d, const double *px_xy,
                          const double *pdn, const double *K, double threshold, const int32_t *samples, int iters,
                          double *KP, double *Rt, uint8_t *inliers, int *n_inliers, double *error, int *best_iter);
int slam_five_point_ransac_batch(slam_ctx *ctx, int S, const int32_t *offsets, const double *px1_xy, const double *px2_xy,
                                 const double *pd1_xy, const double *pd2_xy, const double *K1, const double *K2,
                                 double max_repr_error, const int32_t *samples, int iters, double *E, double *P,
                                 uint8_t *inliers, int *n_inliers, double *error, int *best_iter);
/* slam_pnp_ba for S poses, one workgroup per problem */
int slam_pnp_ba_batch(slam_ctx *ctx, int S, const int32_t *offsets, const double *cams, const double *poses_cw,
                      const double *pixels_yx, const double *points_xyz, int iters_fast, int iterations,
                      double depth_eps, double repr_eps, double *out_poses, double *err_init, double *err_final,
                      uint8_t *outliers, int *n_outliers);

/* bundle_adjustment!(cache::LocalBACache, camera; iterations, repr_eps) --
 * src/bundle_adjustment.jl:1-111 on the flat arrays of src/estimator.jl:16-40:
 * theta = [6P (RotZYX t1,t2,t3, tx,ty,tz) ; 3M], pixels (y,x) 2 x O, 1-based ids.
 * Two passes: iters_fast LM iterations on all observations, outlier flagging
 * (depth < 1e-6 or squared pixel error > repr_eps), then `iterations` LM
 * iterations with outliers zeroed.  The LM outer loop is LeastSquaresOptim's;
 * the step is the exact one from the Schur-complement reduced camera system.
 * stats (may be NULL, 8 doubles): ssr_init, ssr_pass1, ssr_final, iters_pass1,
 * iters_pass2, n_outliers, device_ms, 0. */
int slam_local_ba(slam_ctx *ctx, double fx, double fy, double cx, double cy,
                  int P, int M, int O,
                  double *theta, const uint8_t *theta_const, const double *pixels_yx,
                  const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                  int iters_fast, int iterations, double repr_eps, double *stats);

/* bundle_adjustment! for S windows in one set of launches (no reference counterpart, like the other *_batch entry points): the
 * caller is the estimator task of S lock-stepped SlamManagers -- every key-frame of every manager owes one local_bundle_adjustment!
 * (src/estimator.jl:78-99, :317-347; src/bundle_adjustment.jl:35-54).  Window z has Pn[z] poses, Mn[z] points, On[z] observations and
 * camera cams[4 z ..] = fx, fy, cx, cy; the arrays of the windows are stored back to back in window order: theta (6 Pn[z] + 3 Mn[z]
 * doubles each, in / out), theta_const (Pn[z] bytes), pixels_yx (2 On[z] doubles), pose_ids / point_ids (On[z], 1-based, local to the
 * window), outliers (On[z] bytes, out), stats (8 doubles per window as slam_local_ba's, may be NULL).  Each window runs its own
 * device-side Levenberg-Marquardt state (a converged window idles) and its results equal slam_local_ba's on its arrays TO ROUNDING
 * (theta <= 1e-6 relative, costs 1e-8; the batch kernels sum a window's points in a different order and solve the reduced system with
 * another elimination, so the outlier flag of an observation whose squared residual lies within rounding of repr_eps may differ
 * between the two entry points -- none does in the test suite's windows); windows the
 * banded group kernels do not cover (no banded pose order, a point with > 448 observations, no observations) are solved one by one
 * after the batch.  status (S ints, may be NULL): per-window code -- 0, SLAM_ERR_NUMERIC (reduced system not positive definite: that
 * window's theta / outliers are left unchanged) or SLAM_ERR_ARG; with status the call returns SLAM_OK unless the batch itself failed,
 * without it the first window error is returned.  stats[6] of every window is the device time of the WHOLE batch (one set of
 * launches).  Host set-up runs on a parked pool of min(max(hardware threads / 4, 4), 32) threads (SLAMHIP_BA_THREADS overrides).
 * Windows of <= 5 consecutive free poses are solved by one launch with one or two workgroups each; the two-workgroup form waits
 * for its partner with a bound and, should the partner not show up (compute units held by other processes), the call is solved
 * again with one workgroup per window -- a slow call, never a hung queue. */
int slam_local_ba_batch(slam_ctx *ctx, int S, const double *cams, const int32_t *Pn, const int32_t *Mn, const int32_t *On,
                        double *theta, const uint8_t *theta_const, const double *pixels_yx,
                        const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                        int iters_fast, int iterations, double repr_eps, double *stats, int32_t *status);

/* The same call in two halves (round 6): _begin hands everything -- structure analysis, staging, upload, solve, download, scatter into the
 * caller's arrays -- to a thread of the library's own and returns at once; _end waits for it and returns the call's code (message:
 * slam_last_error).  The estimator task (src/estimator.jl:78-99) prepares the next key-frame's windows while this one's are planned and
 * solved, without a host thread of its own.  Between the two calls the context belongs to the job -- no other call on it; one job per
 * context, several contexts for several batches in flight -- and every array passed to _begin stays valid and untouched (theta, outliers,
 * stats and status are written by the job).  slam_ctx_destroy waits for a job still in flight. */
int slam_local_ba_batch_begin(slam_ctx *ctx, int S, const double *cams, const int32_t *Pn, const int32_t *Mn, const int32_t *On,
                              double *theta, const uint8_t *theta_const, const double *pixels_yx,
                              const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                              int iters_fast, int iterations, double repr_eps, double *stats, int32_t *status);
int slam_local_ba_batch_end(slam_ctx *ctx);

/* pnp_bundle_adjustment(camera, pose, pixels, points; iterations, depth_eps,
 * repr_eps) -- src/bundle_adjustment.jl:113-171.  pose_cw/out_pose: 4x4
 * column-major.  out_pose = identity when fewer than 5 inliers remain (:157-161). */
int slam_pnp_ba(slam_ctx *ctx, double fx, double fy, double cx, double cy,
                const double pose_cw[16], const double *pixels_yx, const double *points_xyz, int n,
                int iters_fast, int iterations, double depth_eps, double repr_eps,
                double out_pose[16], double *err_init, double *err_final,
                uint8_t *outliers, int *n_outliers);

/* Point-sharded BA for multi-GPU windows: each rank owns the
 * observations of a subset of map points; per LM iteration it produces its
 * contribution [S (6P x 6P col-major) ; rhs (6P) ; diag(Jp'Jp) (6P) ; ssr ; pad]
 * into a caller-provided DEVICE buffer of slam_ba_reduce_len(P) doubles, which
 * the host all-reduces (RCCL via torch.distributed) before slam_ba_solve.
 * theta holds ALL poses (replicated) and this shard's points. */
int    slam_ba_create(slam_ctx *ctx, double fx, double fy, double cx, double cy,
                      int P, int M_local, int O_local,
                      const double *theta, const uint8_t *theta_const, const double *pixels_yx,
                      const int64_t *pose_ids, const int64_t *point_ids_local, slam_ba **out);
int    slam_ba_destroy(slam_ba *ba);
int64_t slam_ba_reduce_len(int P);
/* linearise at the current theta and write the local contribution.  A shard keeps ONE reduce buffer for its lifetime: from the second
 * build on only the blocks inside the band of the reduced system (and rhs / diag / ssr) are rewritten -- the rest of the buffer was
 * zeroed by the first build and must not be written by anyone else (an in-place all-reduce keeps zeros zero). */
int    slam_ba_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double inv_delta, double *reduce_dev);
/* solve the (all-reduced) system, back-substitute local points, evaluate the
 * trial step: writes [trial_ssr_local, predicted_ssr_local, max|dx| local, 0]
 * to trial_dev (4 doubles, device) */
int    slam_ba_solve(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, double inv_delta, double *trial_dev);
/* accept (1) or reject (0) the trial step */
int    slam_ba_commit(slam_ctx *ctx, slam_ba *ba, int accept);
/* flag outliers at the current theta; returns local count through n_out */
int    slam_ba_flag_outliers(slam_ctx *ctx, slam_ba *ba, double repr_eps, double depth_eps, int *n_out);
/* read back theta (6P + 3 M_local) and the outlier flags (O_local) */
int    slam_ba_download(slam_ctx *ctx, slam_ba *ba, double *theta, uint8_t *outliers);
/* The pose order slam_local_ba solves in (host work only, no device call; ctx-free).  The reduced camera system of a window of
 * consecutive key-frames is block-banded in the caller's order (S_pq = 0 for |p - q| > hb) and takes the single-launch banded
 * solver when hb <= 20.  A window that is not -- loop closures: the first and the last key-frames share map points
 * (src/map_manager.jl:300-449 re-associates old map points), so the covisibility chain is a ring -- is solved on relabelled poses when
 * a fold of the ring or a Cuthill-McKee order of the free poses' covisibility graph brings hb <= 20 (constant poses first); theta
 * comes back in the caller's order.  order_out (P entries, may be NULL): the caller's 0-based pose at the solver's position k;
 * hb_out (may be NULL): the half-bandwidth in that order.  Returns 1 if the window is reordered, 0 if the caller's order is kept,
 * < 0 on bad arguments.  (SLAMHIP_BA_NO_REORDER=1 in the environment keeps the caller's order: measurement knob.) */
int    slam_ba_plan_order(int P, int M, int O, const uint8_t *theta_const, const int64_t *pose_ids, const int64_t *point_ids,
                          int32_t *order_out, int *hb_out);
/* Block half-bandwidth of the shard's reduced system (S_pq = 0 for |p - q| > hb).  The all-reduced system has the maximum
 * over the ranks: the driver sets that on every rank before the first solve (single-launch banded solver). */
int    slam_ba_halfband(const slam_ba *ba);
int    slam_ba_set_halfband(slam_ba *ba, int hb);

/* Device-paced LM pass of the sharded path.  Every call below returns after ENQUEUEING on ctx's stream; the accept / reject
 * decision of LeastSquaresOptim's outer loop (bundle_adjustment.jl:35-54) is taken on the device from the
 * gathered trial costs, identically on every rank, so a pass contains no host synchronisation.  Protocol per pass:
 *     slam_ba_lm_begin(ignore, red);  slam_comm_allreduce_sum(red);  slam_ba_lm_start(red, first_pass);
 *     for it = 1 .. iters:
 *         if (it > 1) { slam_ba_lm_build(ignore, red);  slam_comm_allreduce_sum(red); }
 *         slam_ba_lm_solve(red, ignore, trial);  slam_comm_allgather(trial -> gathered, 4);
 *         slam_ba_lm_step(gathered, nranks, it);
 *     slam_ba_lm_state(out8);                       -- the only synchronising call
 * Once the device-side state says "converged" the remaining iterations are no-ops (every kernel early-outs). */
int    slam_ba_lm_begin(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double *reduce_dev);
int    slam_ba_lm_start(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, int first_pass);
int    slam_ba_lm_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double *reduce_dev);
int    slam_ba_lm_solve(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, int ignore_outliers, double *trial_dev);
int    slam_ba_lm_step(slam_ctx *ctx, slam_ba *ba, const double *gathered_dev, int nranks, int iter_tag);
/* synchronises; out8 = {ssr, iterations, converged, delta, chol_fail, ssr at the start of the first pass, trial ssr, max|dx|} */
int    slam_ba_lm_state(slam_ctx *ctx, slam_ba *ba, double *out8);

/* ---- RCCL collectives for the sharded BA ("ncclAllReduce(sum, f64) of [S ; g ; cost] over xGMI") -----------
 * One communicator per process / GPU.  Rank 0 obtains a 128-byte id (slam_comm_unique_id) and hands it to the other
 * ranks through the host's own channel (the Julia side: MPI.jl / Distributed / a file); every rank then calls
 * slam_comm_create.  The collectives are enqueued on ctx's stream (device-ordered with the slam_ba_lm_* calls) and
 * return immediately.  librccl is bound at run time: a process that never creates a communicator does not load it. */
typedef struct slam_comm slam_comm;
int    slam_comm_unique_id(void *id128);
int    slam_comm_create(slam_ctx *ctx, int nranks, int rank, const void *id128, slam_comm **out);
int    slam_comm_destroy(slam_comm *comm);
int    slam_comm_size(const slam_comm *comm);
int    slam_comm_rank(const slam_comm *comm);
/* in-place sum of buf_dev[0 .. count) (Float64) over all ranks */
int    slam_comm_allreduce_sum(slam_ctx *ctx, slam_comm *comm, double *buf_dev, int64_t count);
/* recv_dev[r * count .. (r + 1) * count) = rank r's send_dev[0 .. count) */
int    slam_comm_allgather(slam_ctx *ctx, slam_comm *comm, const double *send_dev, double *recv_dev, int64_t count);

#ifdef __cplusplus
}
#endif
#endif
