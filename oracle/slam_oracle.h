/*
 * slam_oracle.h -- CPU restatement ("oracle") of the SLAM.jl hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under slam.jl_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (pxl-th/SLAM.jl, /root/reference) is 100 %
 * Julia, has no tests / golden vectors, Julia is not installed here and most of
 * the arithmetic lives in un-vendored, range-pinned third-party packages
 * (Project.toml:28-45: Images 0.24, ImageFiltering 0.6/0.7, Interpolations 0.13,
 * ImageDraw 0.2, ImageFeatures 0.4, LeastSquaresOptim 0.8, Rotations 1,
 * SparseDiffTools 1).  This file restates the reference's own source line by
 * line and those packages' published algorithms from their documentation; it
 * is cross-checked against independent numpy/scipy restatements and analytic
 * known-answer tests (tests/test_oracle_*.py), NOT against Julia output.
 *
 * Conventions (reference: src/SLAM.jl:22-26, src/extractor.jl:61):
 *   - all arithmetic Float64, compiled with -ffp-contract=off (Julia does not
 *     contract a*b+c);
 *   - images are column-major H x W (y fastest): pixel (y,x), 1-based, lives at
 *     img[(y-1) + (x-1)*H];
 *   - points are (y, x) Float64, 1-based; ids Int64, 1-based.
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 8

/* ---------- filters (ImageFiltering semantics) ------- */
/* Kernel.gaussian(sigma) 1-D factor: length 4*ceil(sigma)+1, normalised. */
int  orc_gaussian_taps(double sigma, double *w /* >= 4*ceil(sigma)+1 */);
/* Separable FIR correlation, dim 1 (y) pass then dim 2 (x) pass; border: 0 =
 * Pad(:replicate), 1 = Fill(0).  k1 along y, k2 along x, both centred, odd. */
void orc_imfilter_sep(double *out, const double *in, int H, int W,
                      const double *k1, int n1, const double *k2, int n2, int border);
/* KernelFactors.IIRGaussian(sigma) applied along dim 1 then dim 2, in place
 * semantics (out may alias in).  border: 0 = replicate, 1 = Fill(0), 2 = NA()
 * (= Fill(0) result divided by Fill(0)-filtered ones). */
void orc_iir_gaussian(double *out, const double *in, int H, int W, double sigma, int border);
/* coefficients, for the tests: a[3], scale, M[9] row-major, asum */
void orc_iir_coeffs(double sigma, double *a, double *scale, double *M, double *asum);
/* ImageTransformations.imresize!(dst, interpolate!(src, BSpline(Linear()))) */
void orc_imresize(double *dst, int Hd, int Wd, const double *src, int Hs, int Ws);
/* Interpolations BSpline(Linear()) on-grid evaluation at 1-based (r, c). */
double orc_bilinear(const double *img, int H, int W, double r, double c);

/* ---------- extractor (src/extractor.jl) ---------------------------------- */
/* get_mask, extractor.jl:116-122 + ImageDraw CirclePointRadius fill rule. */
void orc_get_mask(double *mask, int H, int W, const double *pts_yx, int n, int radius);
/* Images.shi_tomasi on one (cell) view with leading dimension ld. */
void orc_shi_tomasi(double *resp, const double *cell, int h, int w, int ld);
/* _shi_tomasi, extractor.jl:24-42: corners mask (h*w bytes), returns count. */
int  orc_shi_tomasi_cell(uint8_t *corners, double *resp, const double *cell,
                         int h, int w, int ld, int n_keypoints, double min_response);
/* detect, extractor.jl:63-95.  out_rc: (row, col) int64 pairs, 1-based.
 * Returns number of keypoints (may exceed max_points), or -1 if cap too small. */
int  orc_detect(const double *img, int H, int W, const double *cur_yx, int n_cur,
                int max_points, int radius, int grid_rows, int grid_cols, int cell_size,
                double sigma_mask, double min_response, int64_t *out_rc, int cap);
/* describe -> ImageFeatures.create_descriptor(img, kps, BRIEF), extractor.jl:103-105.
 * pattern: n_bits x 4 int32 (dy1, dx1, dy2, dx2) supplied by the caller
 * (Julia's RNG stream is not reproducible outside Julia).
 * out_bits: n_out x (n_bits/64) uint64 words, bit k of the descriptor in word
 * k/64 bit k%64.  Returns n_out. */
int  orc_describe(const double *img, int H, int W, const int64_t *rc, int n,
                  const int32_t *pattern, int n_bits, double sigma, int window,
                  uint64_t *out_bits, int64_t *out_rc);

/* ---------- LK pyramid (src/optical_flow/pyramid.jl) ---------------------- */
typedef struct {
    int levels;                 /* total levels = pyramid_levels + 1 */
    int H[ORC_MAX_LEVELS], W[ORC_MAX_LEVELS];
    int64_t off[ORC_MAX_LEVELS + 1]; /* plane offsets (in doubles) per level */
    double *layers, *Iy, *Ix, *Iyy, *Ixx, *Iyx; /* each off[levels] doubles */
} orc_pyr;

int64_t orc_pyr_layout(int H, int W, int total_levels, int *Hs, int *Ws, int64_t *off);
/* mode 0: constructor semantics (pyramid.jl:40-79: NA() blur, Fill(0) Scharr);
 * mode 1: update! semantics (pyramid.jl:81-137: replicate for both). */
void orc_pyr_build(orc_pyr *p, const double *img, double sigma, int mode);
/* flat-buffer entry for ctypes */
void orc_pyr_build_flat(const double *img, int H, int W, int total_levels, double sigma, int mode,
                        double *layers, double *Iy, double *Ix, double *Iyy, double *Ixx, double *Iyx);

/* ---------- Lucas-Kanade (src/optical_flow/lucas_kanade.jl, tracker.jl) --- */
void orc_svd2x2(const double M[4] /*col-major*/, double U[4], double S[2], double V[4]);
void orc_pinv2x2(const double M[4], double Ginv[4], double S[2]);
/* optflow!, lucas_kanade.jl:9-100.  disp: n x (y,x), in/out.  status in/out is
 * initialised to all-true inside (trues(n)).  sum_order: 0 = reference order
 * (q outer, p inner, one accumulator; lucas_kanade.jl:163-170); 1 = "wave
 * order": element e = q*P + p goes to accumulator e % 64, accumulators are
 * folded with a 6-step butterfly (xor 32,16,1,2,4,8), the order the HIP kernel
 * uses.  Returns n_good (counted from status; the reference's racy counter,
 * SURVEY F9, is not reproduced).  Returns -1 when "Not enough layers". */
int  orc_optflow(double *disp_yx, const orc_pyr *first, const orc_pyr *second,
                 const double *pts_yx, int n, int iterations, int window, int pyramid_levels,
                 double eig_thr, double eps, uint8_t *status, int sum_order, int threads);
/* fb_tracking!, tracker.jl:17-66.  disp0 may be NULL (zeros).  out_yx[i] is
 * only written where status[i].  Returns 0, or -1 for "Not enough layers". */
int  orc_fb_tracking(const orc_pyr *prev, const orc_pyr *cur, const double *pts_yx,
                     const double *disp0_yx, int n, int iterations, int window, int pyramid_levels,
                     double eig_thr, double eps, double max_distance,
                     double *out_yx, uint8_t *status, int sum_order, int threads);
int  orc_fb_tracking_flat(int H, int W, int total_levels,
                          const double *p_layers, const double *p_Iy, const double *p_Ix,
                          const double *p_Iyy, const double *p_Ixx, const double *p_Iyx,
                          const double *c_layers, const double *c_Iy, const double *c_Ix,
                          const double *c_Iyy, const double *c_Ixx, const double *c_Iyx,
                          const double *pts_yx, const double *disp0_yx, int n,
                          int iterations, int window, int pyramid_levels,
                          double eig_thr, double eps, double max_distance,
                          double *out_yx, uint8_t *status, int sum_order, int threads);

/* ---------- bundle adjustment (src/bundle_adjustment.jl) ------------------ */
typedef struct {
    double fx, fy, cx, cy;
    int P, M, O;
    double *theta;              /* 6P + 3M, in/out */
    const uint8_t *theta_const; /* P */
    const double *pixels_yx;    /* 2 x O */
    const int64_t *pose_ids;    /* O, 1-based */
    const int64_t *point_ids;   /* O, 1-based */
    uint8_t *outliers;          /* O, out */
} orc_ba_problem;

typedef struct {
    double ssr_init, ssr_pass1, ssr_final; /* sum of squared residuals */
    int iters_pass1, iters_pass2;          /* LM iterations actually run */
    int n_outliers;
    int64_t inner_iters;                   /* LSMR iterations (solver 0) */
} orc_ba_stats;

/* RotZYX(t1,t2,t3) as a row-major 3x3 (Rotations.jl). */
void orc_rotzyx(double t1, double t2, double t3, double R[9]);
void orc_rotzyx_angles(const double R[9], double *t1, double *t2, double *t3);
/* residue!, bundle_adjustment.jl:13-33 */
void orc_ba_residuals(const orc_ba_problem *p, const double *theta, int ignore_outliers, double *Y);
/* _ba_detect_outliers!, bundle_adjustment.jl:90-111 */
int  orc_ba_detect_outliers(const orc_ba_problem *p, const double *theta, double repr_eps, double depth_eps);
/* bundle_adjustment!, bundle_adjustment.jl:1-55.  solver: 0 = reference style
 * (LeastSquaresOptim LM + Jacobi-preconditioned LSMR, btol=0.5, on the full
 * [6P;3M] system); 1 = same LM outer loop with the exact step from
 * the Schur-complement reduced camera system + Cholesky (what the HIP path
 * computes). */
int  orc_bundle_adjustment(orc_ba_problem *p, int iters_fast, int iterations, double repr_eps,
                           int solver, orc_ba_stats *stats);
int  orc_bundle_adjustment_flat(double fx, double fy, double cx, double cy, int P, int M, int O,
                                double *theta, const uint8_t *theta_const, const double *pixels_yx,
                                const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                                int iters_fast, int iterations, double repr_eps, int solver,
                                double *stats_out /* 8 doubles */);
/* reduced camera system contribution of the map points [m_begin, m_end) for one
 * linearisation (checker for the point-sharded multi-GPU path): S (6P x 6P
 * col-major, no pose damping), g2 = [rhs (6P); diag(Jp'Jp) (6P)], ssr of those
 * points' observations.  inv_delta = 1/Delta (LM trust-region radius). */
void orc_ba_reduced_system(const orc_ba_problem *p, const double *theta, int ignore_outliers,
                           double inv_delta, int m_begin, int m_end, double *S, double *g2, double *ssr);
/* pnp_bundle_adjustment, bundle_adjustment.jl:113-171.  pose_cw / out_pose are
 * 4x4 column-major.  Returns 0. */
int  orc_pnp_ba(double fx, double fy, double cx, double cy, const double pose_cw[16],
                const double *pixels_yx, const double *points_xyz, int n, int iters_fast, int iterations,
                double depth_eps, double repr_eps, double out_pose[16],
                double *err_init, double *err_final, uint8_t *outliers, int *n_outliers);

/* ---- two-view triangulation + gating (mapper.jl:142-262; RecoverPose.triangulate restated) -- orc_tri.c ---- */
void orc_sym4_min_eigvec(double S[16], double v[4]);
void orc_sym4_min_eigvec_invit(const double S[16], double v[4]);
int  orc_triangulate_point(const double *P1, const double *P2, const double *T21, const double *cam1, const double *cam2,
                           const double *px1_yx, const double *px2_yx, double max_error, double min_depth,
                           int gate_always, double parallax, double min_parallax, double *xyz);
int  orc_triangulate(const double *P1, const double *P2, const double *T21, const double *cam1, const double *cam2,
                     const double *px1_yx, const double *px2_yx, int n, double max_error, double min_depth,
                     const double *parallax, double min_parallax, double *out_xyz, unsigned char *status);

/* ---- P3P RANSAC of compute_pose! (front_end.jl:132-219; RecoverPose.p3p_ransac restated) -- orc_p3p.c ---- */
int  orc_quartic_real_roots(const double A[5], double roots[4]);
int  orc_p3p_solve(const double X[9], const double F[9], double Rt[48]);
int  orc_p3p_ransac(const double *pts3d, const double *px_xy, const double *pdn, int n, const double *K, double threshold,
                    const int32_t *samples, int iters, double *KP, double *Rt_out, unsigned char *inliers, double *error,
                    int *best_iter);

/* ---- five-point RANSAC of compute_pose_5pt! (front_end.jl:243-332; RecoverPose.five_point_ransac restated) -- orc_5pt.c ---- */
int  orc_poly_real_roots(const double *p, int deg, double *roots);
int  orc_five_point_solve(const double q1[10], const double q2[10], double Es[90]);
int  orc_essential_poses(const double E[9], double Rt[48]);
int  orc_essential_pose_cheirality(const double E[9], const double q1[10], const double q2[10], double Rt[12]);
int  orc_five_point_ransac(const double *px1, const double *px2, const double *pd1, const double *pd2, int n,
                           const double *K1, const double *K2, double max_repr_error, const int32_t *samples, int iters,
                           double *E_out, double *P_out, unsigned char *inliers, double *error, int *best_iter);

#ifdef __cplusplus
}
#endif
#endif
