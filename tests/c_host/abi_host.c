/*
 * abi_host.c -- a host of libslamhip.so that is NOT Python and has NO PyTorch in the process: the executable stand-in for the Julia
 * `ccall` shim (slam.jl_amd/julia/SLAMHip.jl), which cannot run in this image (no Julia).  Plain C, links libslamhip.so and nothing
 * else of ours; the HIP runtime is whatever the dynamic loader binds for libslamhip.so's NEEDED libamdhip64.so -- the SYSTEM ROCm
 * runtime under /opt/rocm, as in a Julia process (reference: src/SLAM.jl:187-230 is where run!() reaches the seams).
 *
 * It calls the six seams of SURVEY 8b with exactly the argument lists SLAMHip.jl passes (same constants: min_response 1e-4,
 * eigenvalue threshold 1e-4, eps 1e-2, iters_fast 5) plus slam_local_ba_batch and the pose seams, on the raw fixtures
 * tests/c_host/export_fixtures.py writes from tests/golden/hotpath_v1.npz / pose_v1.npz, and compares: keypoint indices, tracking
 * status, descriptors, outlier flags, inlier masks byte for byte; pyramid planes bit for bit; positions / theta / poses at the
 * test-suite tolerances.  Then the threading contract of SURVEY 8b: three pthreads (front-end, mapper, estimator), one context each,
 * running their seams concurrently and checking the same expectations.
 *
 * usage: abi_host FIXTURES.bin      exit status 0 = every check passed.  Prints which libamdhip64 the process mapped.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slamhip.h"

/* ---- fixture container (export_fixtures.py) ---------------------------------------------------------------------------------- */
typedef struct { char name[32]; int32_t dtype, ndim; int64_t dims[4]; int64_t nbytes; const void *data; } entry_t;
static entry_t *g_ent; static int g_nent;

static int load_fixtures(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return -1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 8);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return -1; }
    fclose(f);
    if (memcmp(buf, "SLAMFIX1", 8) != 0) { fprintf(stderr, "%s: not a fixture container\n", path); return -1; }
    int32_t n; memcpy(&n, buf + 8, 4);
    g_ent = (entry_t *)calloc((size_t)n, sizeof(entry_t)); g_nent = n;
    size_t off = 12;
    for (int i = 0; i < n; i++) {
        entry_t *e = &g_ent[i];
        memcpy(e->name, buf + off, 32); off += 32;
        memcpy(&e->dtype, buf + off, 4); off += 4; memcpy(&e->ndim, buf + off, 4); off += 4;
        memcpy(e->dims, buf + off, 32); off += 32; memcpy(&e->nbytes, buf + off, 8); off += 8;
        e->data = buf + off; off += (size_t)((e->nbytes + 7) & ~7LL);
    }
    return 0;
}
static const entry_t *fx(const char *name)
{
    for (int i = 0; i < g_nent; i++) if (strncmp(g_ent[i].name, name, 32) == 0) return &g_ent[i];
    fprintf(stderr, "fixture '%s' missing\n", name); exit(2);
}
#define F64(n) ((const double *)fx(n)->data)
#define I64(n) ((const int64_t *)fx(n)->data)
#define I32(n) ((const int32_t *)fx(n)->data)
#define U8(n) ((const uint8_t *)fx(n)->data)
#define U64(n) ((const uint64_t *)fx(n)->data)
#define ROWS(n) ((int)fx(n)->dims[0])

/* ---- checks ------------------------------------------------------------------------------------------------------------------ */
static int g_fail;                       /* (written under g_mu from the threads) */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static void report(const char *who, const char *what, int ok, double val)
{
    pthread_mutex_lock(&g_mu);
    if (!ok) g_fail++;
    printf("[%s] %-52s %s", who, what, ok ? "ok" : "FAILED");
    if (!isnan(val)) printf("  (%.3g)", val);
    printf("\n");
    pthread_mutex_unlock(&g_mu);
}
#define CALL(who, ctx, expr) do { int rc_ = (expr); if (rc_ != 0) { pthread_mutex_lock(&g_mu); g_fail++; \
    printf("[%s] %s -> %d: %s\n", who, #expr, rc_, slam_last_error(ctx)); pthread_mutex_unlock(&g_mu); return 1; } } while (0)
static double maxabs(const double *a, const double *b, size_t n) { double m = 0; for (size_t i = 0; i < n; i++) { double d = fabs(a[i] - b[i]); if (!(d <= m)) m = d; } return m; }
static double maxmag(const double *a, size_t n) { double m = 0; for (size_t i = 0; i < n; i++) if (fabs(a[i]) > m) m = fabs(a[i]); return m; }

/* ---- the seams, as SLAMHip.jl calls them ---------------------------------------------------------------------------------------- */
static double *image_f64(const char *name, int H, int W)              /* Gray{Float64}.(frame): raw / 255 */
{
    const uint8_t *u = U8(name);
    double *d = (double *)malloc((size_t)H * W * 8);
    for (size_t i = 0; i < (size_t)H * W; i++) d[i] = (double)u[i] / 255.0;
    return d;
}

/* detect + describe (src/extractor.jl:63-105) */
static int seam_extractor(const char *who, slam_ctx *ctx)
{
    const int H = I32("shape")[0], W = I32("shape")[1];
    double *img0 = image_f64("img0_u8", H, W);
    const int gr = (H + 34) / 35, gc = (W + 34) / 35, cap = 256;
    int64_t out[2 * 256]; int n = 0;
    CALL(who, ctx, slam_detect(ctx, img0, H, W, NULL, 0, 60, 17, gr, gc, 35, 3.0, 1e-4, out, cap, &n));
    report(who, "detect, no current keypoints: indices byte-equal", n == ROWS("kp_nomask") && memcmp(out, I64("kp_nomask"), (size_t)n * 16) == 0, NAN);
    CALL(who, ctx, slam_detect(ctx, img0, H, W, F64("cur"), ROWS("cur"), 60, 17, gr, gc, 35, 3.0, 1e-4, out, cap, &n));
    report(who, "detect, avoidance mask: indices byte-equal", n == ROWS("kp_mask") && memcmp(out, I64("kp_mask"), (size_t)n * 16) == 0, NAN);
    const int nk = ROWS("kp_nomask"), nb = ROWS("brief_pattern");
    uint64_t *bits = (uint64_t *)calloc((size_t)nk * (nb / 64), 8); int64_t *orc = (int64_t *)calloc((size_t)nk * 2, 8); int m = 0;
    CALL(who, ctx, slam_describe(ctx, img0, H, W, I64("kp_nomask"), nk, I32("brief_pattern"), nb, sqrt(2.0), 9, bits, orc, &m));
    report(who, "describe: BRIEF-256 bits and surviving keypoints byte-equal",
           m == ROWS("brief_bits") && memcmp(bits, U64("brief_bits"), (size_t)m * (nb / 64) * 8) == 0 && memcmp(orc, I64("brief_rc"), (size_t)m * 16) == 0, NAN);
    free(bits); free(orc); free(img0);
    return 0;
}

static int plane_equal(const char *who, slam_ctx *ctx, slam_pyr *p, int plane, int level, const char *name, const char *what)
{
    int h, w; slam_pyr_shape(p, level, &h, &w);
    double *buf = (double *)malloc((size_t)h * w * 8);
    CALL(who, ctx, slam_pyr_download(ctx, p, plane, level, buf));
    const int ok = fx(name)->nbytes == (int64_t)h * w * 8 && memcmp(buf, F64(name), (size_t)h * w * 8) == 0;
    report(who, what, ok, NAN);
    free(buf);
    return 0;
}

/* LKPyramid / update! / copy! / deepcopy (src/optical_flow/pyramid.jl:28-137) + fb_tracking! (src/tracker.jl:70-82) */
static int seam_pyramid_and_tracking(const char *who, slam_ctx *ctx, int with_ctor)
{
    const int H = I32("shape")[0], W = I32("shape")[1];
    double *img0 = image_f64("img0_u8", H, W), *img1 = image_f64("img1_u8", H, W);
    slam_pyr *p0 = NULL, *p1 = NULL, *pc = NULL, *pu = NULL, *cl = NULL;
    CALL(who, ctx, slam_pyr_create(ctx, H, W, 2, &p0));
    CALL(who, ctx, slam_pyr_create(ctx, H, W, 2, &p1));
    CALL(who, ctx, slam_pyr_update(ctx, p0, img0, 1, 1.0));
    CALL(who, ctx, slam_pyr_update(ctx, p1, img1, 1, 1.0));
    if (plane_equal(who, ctx, p0, 1, 1, "upd_Iy_l1", "update!: Iy level 1 bit-equal")) return 1;
    if (plane_equal(who, ctx, p0, 5, 2, "upd_Iyx_l2", "update!: Iyx level 2 bit-equal")) return 1;
    if (plane_equal(who, ctx, p0, 0, 2, "upd_layer_l2", "update!: layer 2 bit-equal")) return 1;
    if (with_ctor) {
        CALL(who, ctx, slam_pyr_create(ctx, H, W, 2, &pc));
        CALL(who, ctx, slam_pyr_update(ctx, pc, img0, 0, 1.0));                       /* constructor semantics */
        if (plane_equal(who, ctx, pc, 4, 1, "ctor_Ixx_l1", "LKPyramid(image): Ixx level 1 bit-equal")) return 1;
        if (plane_equal(who, ctx, pc, 0, 1, "ctor_layer_l1", "LKPyramid(image): layer 1 bit-equal")) return 1;
        CALL(who, ctx, slam_pyr_create(ctx, H, W, 2, &pu));
        CALL(who, ctx, slam_pyr_update_u8(ctx, pu, U8("img0_u8"), 1, 1.0));           /* the KITTI reader's bytes, raw / 255 on the device */
        if (plane_equal(who, ctx, pu, 1, 1, "upd_Iy_l1", "update! from uint8: Iy level 1 bit-equal")) return 1;
        CALL(who, ctx, slam_pyr_copy(ctx, pc, p0));                                   /* copy!(dst, src) */
        if (plane_equal(who, ctx, pc, 5, 2, "upd_Iyx_l2", "copy!: Iyx level 2 bit-equal")) return 1;
        CALL(who, ctx, slam_pyr_clone(ctx, p0, &cl));                                 /* deepcopy */
        if (plane_equal(who, ctx, cl, 0, 2, "upd_layer_l2", "deepcopy: layer 2 bit-equal")) return 1;
    }
    const int n = ROWS("kp_nomask");
    double *pts = (double *)malloc((size_t)n * 16), *out = (double *)malloc((size_t)n * 16); uint8_t *st = (uint8_t *)calloc((size_t)n, 1);
    for (int i = 0; i < 2 * n; i++) pts[i] = (double)I64("kp_nomask")[i];
    CALL(who, ctx, slam_fb_track(ctx, p0, p1, pts, NULL, n, 2, 9, 30, 1e-4, 1e-2, 1.0, out, st));
    const uint8_t *est = U8("lk_status"); const double *eo = F64("lk_out");
    double worst = 0; int same = memcmp(st, est, (size_t)n) == 0;
    for (int i = 0; i < n; i++) if (est[i]) { worst = fmax(worst, fabs(out[2 * i] - eo[2 * i])); worst = fmax(worst, fabs(out[2 * i + 1] - eo[2 * i + 1])); }
    report(who, "fb_tracking!: status byte-equal, positions <= 1e-9 px", same && worst <= 1e-9, worst);
    free(pts); free(out); free(st); free(img0); free(img1);
    slam_pyr_destroy(p0); slam_pyr_destroy(p1);
    if (pc) slam_pyr_destroy(pc);
    if (pu) slam_pyr_destroy(pu);
    if (cl) slam_pyr_destroy(cl);
    return 0;
}

/* bundle_adjustment! (src/bundle_adjustment.jl:1-111) single and as a batch of three windows */
static int seam_local_ba(const char *who, slam_ctx *ctx, int with_batch)
{
    const int P = ROWS("ba_const"), O = ROWS("ba_pose_ids"), nth = ROWS("ba_theta0"), M = (nth - 6 * P) / 3;
    const double *cam = F64("ba_cam");
    double *th = (double *)malloc((size_t)nth * 8); uint8_t *ol = (uint8_t *)calloc((size_t)O, 1); double stats[8];
    memcpy(th, F64("ba_theta0"), (size_t)nth * 8);
    CALL(who, ctx, slam_local_ba(ctx, cam[0], cam[1], cam[2], cam[3], P, M, O, th, U8("ba_const"), F64("ba_pixels"), I64("ba_pose_ids"), I64("ba_point_ids"),
                                 ol, 5, 10, 5.0, stats));
    const double dth = maxabs(th, F64("ba_theta"), (size_t)nth), ssr = F64("ba_ssr")[2];
    report(who, "bundle_adjustment!: outliers byte-equal, theta <= 1e-6", memcmp(ol, U8("ba_outliers"), (size_t)O) == 0 && dth <= 1e-6, dth);
    report(who, "bundle_adjustment!: final cost <= 1e-8 relative", fabs(stats[2] - ssr) <= 1e-8 * ssr, fabs(stats[2] - ssr) / ssr);
    if (with_batch) {
        enum { S = 3 };
        double cams[4 * S], *tb = (double *)malloc((size_t)S * nth * 8), *px = (double *)malloc((size_t)S * O * 16), st8[8 * S];
        int32_t Pn[S], Mn[S], On[S], status[S]; uint8_t tc[S * 64], *ob = (uint8_t *)calloc((size_t)S * O, 1);
        int64_t *pi = (int64_t *)malloc((size_t)S * O * 8), *li = (int64_t *)malloc((size_t)S * O * 8);
        for (int z = 0; z < S; z++) {
            memcpy(cams + 4 * z, cam, 32); Pn[z] = P; Mn[z] = M; On[z] = O;
            memcpy(tb + (size_t)z * nth, F64("ba_theta0"), (size_t)nth * 8); memcpy(tc + z * P, U8("ba_const"), (size_t)P);
            memcpy(px + (size_t)z * O * 2, F64("ba_pixels"), (size_t)O * 16);
            memcpy(pi + (size_t)z * O, I64("ba_pose_ids"), (size_t)O * 8); memcpy(li + (size_t)z * O, I64("ba_point_ids"), (size_t)O * 8);
        }
        CALL(who, ctx, slam_local_ba_batch(ctx, S, cams, Pn, Mn, On, tb, tc, px, pi, li, ob, 5, 10, 5.0, st8, status));
        int ok = 1; double worst = 0;
        for (int z = 0; z < S; z++) {
            ok = ok && status[z] == 0 && memcmp(ob + (size_t)z * O, U8("ba_outliers"), (size_t)O) == 0;
            worst = fmax(worst, maxabs(tb + (size_t)z * nth, F64("ba_theta"), (size_t)nth));
        }
        report(who, "slam_local_ba_batch (3 windows): outliers equal, theta <= 1e-6", ok && worst <= 1e-6, worst);
        free(tb); free(px); free(ob); free(pi); free(li);
    }
    free(th); free(ol);
    return 0;
}

/* pnp_bundle_adjustment (src/bundle_adjustment.jl:113-171) */
static int seam_pnp(const char *who, slam_ctx *ctx)
{
    const int n = ROWS("pnp_px"); const double *cam = F64("pnp_cam");
    double pose[16], e0 = 0, e1 = 0; int no = 0; uint8_t *ol = (uint8_t *)calloc((size_t)n, 1);
    CALL(who, ctx, slam_pnp_ba(ctx, cam[0], cam[1], cam[2], cam[3], F64("pnp_pose0"), F64("pnp_px"), F64("pnp_pts"), n, 5, 10, 1e-6, 3.0, pose, &e0, &e1, ol, &no));
    const double *sc = F64("pnp_scal"); const double dp = maxabs(pose, F64("pnp_pose"), 16);
    report(who, "pnp_bundle_adjustment: outliers byte-equal, pose <= 1e-8", memcmp(ol, U8("pnp_outl"), (size_t)n) == 0 && no == (int)sc[2] && dp <= 1e-8, dp);
    report(who, "pnp_bundle_adjustment: errors <= 1e-9 / 1e-8 relative", fabs(e0 - sc[0]) <= 1e-9 * sc[0] && fabs(e1 - sc[1]) <= 1e-8 * sc[1], fabs(e1 - sc[1]) / sc[1]);
    free(ol);
    return 0;
}

/* the mapper's and the front-end's pose seams (src/mapper.jl:142-262, src/front_end.jl:132-332) */
static int seam_triangulate(const char *who, slam_ctx *ctx)
{
    const int n = ROWS("tri_px1");
    double *xyz = (double *)malloc((size_t)n * 24); uint8_t *st = (uint8_t *)calloc((size_t)n, 1);
    CALL(who, ctx, slam_triangulate(ctx, F64("tri_P1"), F64("tri_P2"), F64("tri_T21"), F64("tri_cam"), F64("tri_cam"), F64("tri_px1"), F64("tri_px2"), n,
                                    3.0, 0.1, NULL, 20.0, xyz, st));
    const double *ex = F64("tri_xyz"); double worst = 0;
    for (int i = 0; i < n; i++) { const double mg = maxmag(ex + 3 * i, 3); worst = fmax(worst, maxabs(xyz + 3 * i, ex + 3 * i, 3) / (mg > 0 ? mg : 1)); }
    report(who, "triangulate: status byte-equal, points <= 1e-12 relative", memcmp(st, U8("tri_status"), (size_t)n) == 0 && worst <= 1e-12, worst);
    free(xyz); free(st);
    return 0;
}
static int seam_pose_ransac(const char *who, slam_ctx *ctx)
{
    {
        const int n = ROWS("p3p_pts"), it = ROWS("p3p_samples");
        double KP[12], Rt[12], err = 0; int cnt = 0, best = 0; uint8_t *inl = (uint8_t *)calloc((size_t)n, 1);
        CALL(who, ctx, slam_p3p_ransac(ctx, F64("p3p_pts"), F64("p3p_px"), F64("p3p_pdn"), n, F64("p3p_K"), 3.0, I32("p3p_samples"), it, KP, Rt, inl, &cnt, &err, &best));
        const double *sc = F64("p3p_scal");
        report(who, "p3p_ransac: winner, inlier mask, KP, [R t], error bit-equal",
               cnt == (int)sc[0] && err == sc[1] && best == (int)sc[2] && memcmp(inl, U8("p3p_inliers"), (size_t)n) == 0 && memcmp(KP, F64("p3p_KP"), 96) == 0 && memcmp(Rt, F64("p3p_Rt"), 96) == 0, NAN);
        free(inl);
    }
    {
        const int n = ROWS("fp_px1"), it = ROWS("fp_samples");
        double E[9], Pm[12], err = 0; int cnt = 0, best = 0; uint8_t *inl = (uint8_t *)calloc((size_t)n, 1);
        CALL(who, ctx, slam_five_point_ransac(ctx, F64("fp_px1"), F64("fp_px2"), F64("fp_pd1"), F64("fp_pd2"), n, F64("fp_K"), F64("fp_K"), 3.0, I32("fp_samples"), it,
                                              E, Pm, inl, &cnt, &err, &best));
        const double *sc = F64("fp_scal");
        report(who, "five_point_ransac: winner, inlier mask, E, [R t], error bit-equal",
               cnt == (int)sc[0] && err == sc[1] && best == (int)sc[2] && memcmp(inl, U8("fp_inliers"), (size_t)n) == 0 && memcmp(E, F64("fp_E"), 72) == 0 && memcmp(Pm, F64("fp_P"), 96) == 0, NAN);
        free(inl);
    }
    return 0;
}

/* ---- SURVEY 8b threading contract: front-end, mapper and estimator tasks, one context each ---------------------------------------- */
typedef struct { const char *who; int role, rounds, rc; } task_t;
static void *task_main(void *arg)
{
    task_t *t = (task_t *)arg;
    slam_ctx *ctx = NULL;
    if (slam_ctx_create(0, &ctx) != 0) { report(t->who, "slam_ctx_create", 0, NAN); t->rc = 1; return NULL; }
    for (int r = 0; r < t->rounds && !t->rc; r++) {
        if (t->role == 0) t->rc = seam_pyramid_and_tracking(t->who, ctx, 0) || seam_pose_ransac(t->who, ctx) || seam_pnp(t->who, ctx);      /* front-end: preprocess! + track + compute_pose! */
        else if (t->role == 1) t->rc = seam_extractor(t->who, ctx) || seam_triangulate(t->who, ctx);                                         /* mapper: extract_keypoints! + triangulate */
        else t->rc = seam_local_ba(t->who, ctx, 1);                                                                                          /* estimator: local_bundle_adjustment! */
    }
    slam_ctx_destroy(ctx);
    return NULL;
}

static void print_hip_runtime(void)
{
    FILE *f = fopen("/proc/self/maps", "r");
    char line[1024], last[1024] = ""; int torch = 0;
    if (!f) return;
    while (fgets(line, sizeof line, f)) {
        char *p = strchr(line, '/');
        if (!p) continue;
        p[strcspn(p, "\n")] = 0;
        if (strstr(p, "torch")) torch = 1;
        if (strstr(p, "libamdhip64") && strcmp(p, last) != 0) { printf("HIP runtime mapped: %s\n", p); strncpy(last, p, sizeof last - 1); }
    }
    fclose(f);
    printf("PyTorch libraries in this process: %s\n", torch ? "YES (unexpected)" : "none");
    if (torch) g_fail++;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s FIXTURES.bin\n", argv[0]); return 2; }
    if (load_fixtures(argv[1])) return 2;
    setvbuf(stdout, NULL, _IOLBF, 0);
    printf("%s\n", slam_version());
    slam_ctx *ctx = NULL;
    if (slam_ctx_create(0, &ctx) != 0) { printf("slam_ctx_create: %s\n", slam_last_error(NULL)); return 1; }
    print_hip_runtime();
    int rc = seam_extractor("main", ctx) || seam_pyramid_and_tracking("main", ctx, 1) || seam_local_ba("main", ctx, 1) || seam_pnp("main", ctx)
             || seam_triangulate("main", ctx) || seam_pose_ransac("main", ctx);
    /* an error of the reference's own kind: "Not enough layers in pyramids." (lucas_kanade.jl:15) must come back as a code, not a crash */
    {
        slam_pyr *p = NULL; double pt[2] = {10, 10}, o[2]; uint8_t s;
        if (slam_pyr_create(ctx, 70, 105, 1, &p) == 0) {
            const int e = slam_fb_track(ctx, p, p, pt, NULL, 1, 3, 9, 30, 1e-4, 1e-2, 1.0, o, &s);
            report("main", "fb_tracking! with too few layers -> SLAM_ERR_LAYERS + message", e == SLAM_ERR_LAYERS && strstr(slam_last_error(ctx), "layers") != NULL, NAN);
            slam_pyr_destroy(p);
        }
    }
    slam_ctx_destroy(ctx);
    if (!rc) {
        task_t tasks[3] = {{"front-end", 0, 6, 0}, {"mapper", 1, 6, 0}, {"estimator", 2, 6, 0}};
        pthread_t th[3];
        for (int i = 0; i < 3; i++) pthread_create(&th[i], NULL, task_main, &tasks[i]);
        for (int i = 0; i < 3; i++) { pthread_join(th[i], NULL); rc = rc || tasks[i].rc; }
    }
    printf("%s: %d failed check(s)\n", (rc || g_fail) ? "FAILED" : "ALL OK", g_fail);
    return (rc || g_fail) ? 1 : 0;
}
