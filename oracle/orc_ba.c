/*
 * orc_ba.c -- oracle (TEST INFRASTRUCTURE ONLY, parity unpinned; see
 * slam_oracle.h): local bundle adjustment and single-pose refinement.
 *
 * Follows /root/reference/src/bundle_adjustment.jl:1-171, src/camera.jl:62-67,
 * src/frame.jl:432-450 and the published LeastSquaresOptim 0.8
 * Levenberg-Marquardt / LSMR algorithm and
 * Rotations.jl RotZYX convention (A.9).  The reference differentiates
 * residue! with forward-mode AD; AD of this closed-form residual equals the
 * analytic Jacobian below to rounding.
 */
#include "slam_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* LeastSquaresOptim levenberg_marquardt.jl constants */
#define LM_MAX_DELTA 1e16
#define LM_MIN_DELTA 1e-16
#define LM_MIN_STEP_QUALITY 1e-3
#define LM_MIN_DIAGONAL 1e-6
#define LM_MAX_DIAGONAL 1e32
#define LM_DELTA0 10.0
#define LM_XTOL 1e-8
#define LM_FTOL 1e-8
#define LM_GTOL 1e-8

/* RotZYX(t1,t2,t3) = Rz(t1)*Ry(t2)*Rx(t3), row-major 3x3 */
void orc_rotzyx(double t1, double t2, double t3, double R[9])
{
    double s1 = sin(t1), c1 = cos(t1), s2 = sin(t2), c2 = cos(t2), s3 = sin(t3), c3 = cos(t3);
    R[0] = c1 * c2; R[1] = c1 * s2 * s3 - s1 * c3; R[2] = c1 * s2 * c3 + s1 * s3;
    R[3] = s1 * c2; R[4] = s1 * s2 * s3 + c1 * c3; R[5] = s1 * s2 * c3 - c1 * s3;
    R[6] = -s2;     R[7] = c2 * s3;                R[8] = c2 * c3;
}

/* Rotations.jl RotZYX(::RotMatrix) (frame.jl:434, bundle_adjustment.jl:118) */
void orc_rotzyx_angles(const double R[9], double *t1, double *t2, double *t3)
{
    double a1 = atan2(R[3], R[0]);
    double s1 = sin(a1), c1 = cos(a1);
    *t1 = a1;
    *t2 = atan2(-R[6], sqrt(R[0] * R[0] + R[3] * R[3]));
    *t3 = atan2(R[2] * s1 - R[5] * c1, R[4] * c1 - R[1] * s1);
}

/* residual of one observation + analytic Jacobian.
 * r = (py - (fy*Y/Z + cy), px - (fx*X/Z + cx)), bundle_adjustment.jl:25-30.
 * Jp: 2x6 row-major (d r / d (t1,t2,t3,tx,ty,tz)), Jl: 2x3 row-major (d r / d X). */
static void obs_eval(const double *pose, const double *X, double py, double px,
                     double fx, double fy, double cx, double cy,
                     double r[2], double *Jp, double *Jl, double *depth)
{
    double s1 = sin(pose[0]), c1 = cos(pose[0]), s2 = sin(pose[1]), c2 = cos(pose[1]);
    double s3 = sin(pose[2]), c3 = cos(pose[2]);
    double R[9] = {c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3,
                   s1 * c2, s1 * s2 * s3 + c1 * c3, s1 * s2 * c3 - c1 * s3,
                   -s2, c2 * s3, c2 * c3};
    double x = (R[0] * X[0] + R[1] * X[1] + R[2] * X[2]) + pose[3];
    double y = (R[3] * X[0] + R[4] * X[1] + R[5] * X[2]) + pose[4];
    double z = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2]) + pose[5];
    double iz = 1.0 / z;
    r[0] = py - (fy * y * iz + cy);
    r[1] = px - (fx * x * iz + cx);
    if (depth) *depth = z;
    if (!Jl) return;
    /* d r / d pt (pt = (x,y,z)) */
    double dy[3] = {0.0, -fy * iz, fy * y * iz * iz};
    double dx[3] = {-fx * iz, 0.0, fx * x * iz * iz};
    for (int k = 0; k < 3; k++) {
        Jl[k] = dy[0] * R[k] + dy[1] * R[3 + k] + dy[2] * R[6 + k];
        Jl[3 + k] = dx[0] * R[k] + dx[1] * R[3 + k] + dx[2] * R[6 + k];
    }
    if (!Jp) return;
    double d1[9] = {-s1 * c2, -s1 * s2 * s3 - c1 * c3, -s1 * s2 * c3 + c1 * s3,
                    c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3,
                    0, 0, 0};
    double d2[9] = {-c1 * s2, c1 * c2 * s3, c1 * c2 * c3,
                    -s1 * s2, s1 * c2 * s3, s1 * c2 * c3,
                    -c2, -s2 * s3, -s2 * c3};
    double d3[9] = {0, c1 * s2 * c3 + s1 * s3, -c1 * s2 * s3 + s1 * c3,
                    0, s1 * s2 * c3 - c1 * s3, -s1 * s2 * s3 - c1 * c3,
                    0, c2 * c3, -c2 * s3};
    const double *dR[3] = {d1, d2, d3};
    for (int k = 0; k < 3; k++) {
        const double *D = dR[k];
        double vx = D[0] * X[0] + D[1] * X[1] + D[2] * X[2];
        double vy = D[3] * X[0] + D[4] * X[1] + D[5] * X[2];
        double vz = D[6] * X[0] + D[7] * X[1] + D[8] * X[2];
        Jp[k] = dy[0] * vx + dy[1] * vy + dy[2] * vz;
        Jp[6 + k] = dx[0] * vx + dx[1] * vy + dx[2] * vz;
    }
    for (int k = 0; k < 3; k++) { Jp[3 + k] = dy[k]; Jp[9 + k] = dx[k]; }
}

void orc_ba_residuals(const orc_ba_problem *p, const double *theta, int ignore_outliers, double *Y)
{
    const double *pts = theta + 6 * (size_t)p->P;
    for (int i = 0; i < p->O; i++) {
        if (ignore_outliers && p->outliers[i]) { Y[2 * i] = 0.0; Y[2 * i + 1] = 0.0; continue; }
        obs_eval(theta + 6 * (p->pose_ids[i] - 1), pts + 3 * (p->point_ids[i] - 1),
                 p->pixels_yx[2 * i], p->pixels_yx[2 * i + 1], p->fx, p->fy, p->cx, p->cy,
                 Y + 2 * i, NULL, NULL, NULL);
    }
}

int orc_ba_detect_outliers(const orc_ba_problem *p, const double *theta, double repr_eps, double depth_eps)
{
    const double *pts = theta + 6 * (size_t)p->P;
    int n = 0;
    for (int i = 0; i < p->O; i++) {
        double r[2], z;
        obs_eval(theta + 6 * (p->pose_ids[i] - 1), pts + 3 * (p->point_ids[i] - 1),
                 p->pixels_yx[2 * i], p->pixels_yx[2 * i + 1], p->fx, p->fy, p->cx, p->cy, r, NULL, NULL, &z);
        int out = z < depth_eps || (r[0] * r[0] + r[1] * r[1]) > repr_eps;
        p->outliers[i] = (uint8_t)out;
        n += out;
    }
    return n;
}

/* -------- sparse Jacobian: per observation Jp (2x6, zero if constant pose or
 * ignored outlier) and Jl (2x3, zero if ignored outlier): the sparsity of
 * _get_jacobian_sparsity, bundle_adjustment.jl:57-88 ---------------------- */
typedef struct { double *Jp, *Jl; uint8_t *has_p, *active; } jac_t;

static void jac_eval(const orc_ba_problem *p, const double *theta, int ignore_outliers, jac_t *J)
{
    const double *pts = theta + 6 * (size_t)p->P;
    for (int i = 0; i < p->O; i++) {
        double r[2];
        int pi = (int)p->pose_ids[i] - 1;
        J->active[i] = !(ignore_outliers && p->outliers[i]);
        J->has_p[i] = J->active[i] && !p->theta_const[pi];
        if (!J->active[i]) { memset(J->Jp + 12 * (size_t)i, 0, 12 * sizeof(double)); memset(J->Jl + 6 * (size_t)i, 0, 6 * sizeof(double)); continue; }
        obs_eval(theta + 6 * pi, pts + 3 * (p->point_ids[i] - 1), p->pixels_yx[2 * i], p->pixels_yx[2 * i + 1],
                 p->fx, p->fy, p->cx, p->cy, r, J->Jp + 12 * (size_t)i, J->Jl + 6 * (size_t)i, NULL);
        if (!J->has_p[i]) memset(J->Jp + 12 * (size_t)i, 0, 12 * sizeof(double));
    }
}

static void colsumabs2(const orc_ba_problem *p, const jac_t *J, double *d)
{
    int n = 6 * p->P + 3 * p->M;
    memset(d, 0, sizeof(double) * n);
    for (int i = 0; i < p->O; i++) {
        double *dp = d + 6 * (p->pose_ids[i] - 1), *dl = d + 6 * p->P + 3 * (p->point_ids[i] - 1);
        const double *jp = J->Jp + 12 * (size_t)i, *jl = J->Jl + 6 * (size_t)i;
        for (int k = 0; k < 6; k++) dp[k] += jp[k] * jp[k] + jp[6 + k] * jp[6 + k];
        for (int k = 0; k < 3; k++) dl[k] += jl[k] * jl[k] + jl[3 + k] * jl[3 + k];
    }
}

/* y (2O) = J x */
static void jmul(const orc_ba_problem *p, const jac_t *J, const double *x, double *y)
{
    for (int i = 0; i < p->O; i++) {
        const double *xp = x + 6 * (p->pose_ids[i] - 1), *xl = x + 6 * p->P + 3 * (p->point_ids[i] - 1);
        const double *jp = J->Jp + 12 * (size_t)i, *jl = J->Jl + 6 * (size_t)i;
        double a = 0, b = 0;
        for (int k = 0; k < 6; k++) { a += jp[k] * xp[k]; b += jp[6 + k] * xp[k]; }
        for (int k = 0; k < 3; k++) { a += jl[k] * xl[k]; b += jl[3 + k] * xl[k]; }
        y[2 * i] = a; y[2 * i + 1] = b;
    }
}

/* x (n) += J' y */
static void jtmul_add(const orc_ba_problem *p, const jac_t *J, const double *y, double *x)
{
    for (int i = 0; i < p->O; i++) {
        double *xp = x + 6 * (p->pose_ids[i] - 1), *xl = x + 6 * p->P + 3 * (p->point_ids[i] - 1);
        const double *jp = J->Jp + 12 * (size_t)i, *jl = J->Jl + 6 * (size_t)i;
        double a = y[2 * i], b = y[2 * i + 1];
        for (int k = 0; k < 6; k++) xp[k] += jp[k] * a + jp[6 + k] * b;
        for (int k = 0; k < 3; k++) xl[k] += jl[k] * a + jl[3 + k] * b;
    }
}

static double nrm2(const double *v, size_t n) { double s = 0; for (size_t i = 0; i < n; i++) s += v[i] * v[i]; return sqrt(s); }
static double sumsq(const double *v, size_t n) { double s = 0; for (size_t i = 0; i < n; i++) s += v[i] * v[i]; return s; }

/* stable Givens (Fong & Saunders sym_ortho) */
static void symortho(double a, double b, double *c, double *s, double *r)
{
    if (b == 0) { *c = a == 0 ? 1.0 : (a > 0 ? 1.0 : -1.0); *s = 0; *r = fabs(a); }
    else if (a == 0) { *c = 0; *s = b > 0 ? 1.0 : -1.0; *r = fabs(b); }
    else if (fabs(b) > fabs(a)) { double t = a / b; *s = (b > 0 ? 1.0 : -1.0) / sqrt(1 + t * t); *c = *s * t; *r = b / *s; }
    else { double t = b / a; *c = (a > 0 ? 1.0 : -1.0) / sqrt(1 + t * t); *s = *c * t; *r = a / *c; }
}

/* Solve min || [J; diag(sqrt(dtd))] dx - [f; 0] || with Jacobi-preconditioned
 * LSMR (Fong & Saunders 2011), atol = 1e-6, btol = 0.5, conlim = 1e8: the
 * LeastSquaresOptim `LSMR()` solver.  Returns LSMR iterations. */
typedef struct { const orc_ba_problem *p; const jac_t *J; double *nrm, *sd, *tmp; int n, m; } lsop_t;

static void op_mul(const lsop_t *A, const double *vin, double *uout) /* u = A v */
{
    const int n = A->n, O2 = 2 * A->p->O;
    for (int j = 0; j < n; j++) A->tmp[j] = vin[j] * A->nrm[j];
    jmul(A->p, A->J, A->tmp, uout);
    for (int j = 0; j < n; j++) uout[O2 + j] = A->sd[j] * A->tmp[j];
}
static void op_tmul(const lsop_t *A, const double *uin, double *vout) /* v = A' u */
{
    const int n = A->n, O2 = 2 * A->p->O;
    memset(A->tmp, 0, sizeof(double) * n);
    jtmul_add(A->p, A->J, uin, A->tmp);
    for (int j = 0; j < n; j++) vout[j] = (A->tmp[j] + A->sd[j] * uin[O2 + j]) * A->nrm[j];
}

static int lsmr_solve(const orc_ba_problem *p, const jac_t *J, const double *f, const double *dtd, double *dx)
{
    const int n = 6 * p->P + 3 * p->M, m = 2 * p->O + n;
    const double atol = 1e-6, btol = 0.5, conlim = 1e8;
    const int maxiter = m > n ? m : n;
    lsop_t A; A.p = p; A.J = J; A.n = n; A.m = m;
    A.nrm = (double *)malloc(sizeof(double) * n); A.sd = (double *)malloc(sizeof(double) * n);
    A.tmp = (double *)malloc(sizeof(double) * n);
    double *u = (double *)malloc(sizeof(double) * m), *tu = (double *)malloc(sizeof(double) * m);
    double *v = (double *)calloc(n, sizeof(double)), *tv = (double *)malloc(sizeof(double) * n);
    double *h = (double *)malloc(sizeof(double) * n), *hbar = (double *)calloc(n, sizeof(double));
    double *x = (double *)calloc(n, sizeof(double));
    colsumabs2(p, J, A.nrm);
    for (int j = 0; j < n; j++) { double t = A.nrm[j] + dtd[j]; A.nrm[j] = t > 0 ? 1.0 / sqrt(t) : 0.0; A.sd[j] = sqrt(dtd[j]); }
    memcpy(u, f, sizeof(double) * 2 * p->O); memset(u + 2 * p->O, 0, sizeof(double) * n);
    double beta = nrm2(u, m), alpha = 0;
    int it = 0;
    if (beta > 0) {
        for (int i = 0; i < m; i++) u[i] /= beta;
        op_tmul(&A, u, v);
        alpha = nrm2(v, n);
        if (alpha > 0) for (int j = 0; j < n; j++) v[j] /= alpha;
    }
    double zetabar = alpha * beta, alphabar = alpha, rho = 1, rhobar = 1, cbar = 1, sbar = 0;
    memcpy(h, v, sizeof(double) * n);
    double betadd = beta, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, d = 0;
    double normA2 = alpha * alpha, maxrbar = 0, minrbar = 1e100, normb = beta;
    const double ctol = conlim > 0 ? 1 / conlim : 0;
    if (alpha * beta != 0) {
        for (it = 1; it <= maxiter; it++) {
            op_mul(&A, v, tu);
            for (int i = 0; i < m; i++) u[i] = tu[i] - alpha * u[i];
            beta = nrm2(u, m);
            if (beta > 0) {
                for (int i = 0; i < m; i++) u[i] /= beta;
                op_tmul(&A, u, tv);
                for (int j = 0; j < n; j++) v[j] = tv[j] - beta * v[j];
                alpha = nrm2(v, n);
                if (alpha > 0) for (int j = 0; j < n; j++) v[j] /= alpha;
            }
            /* damping lives inside A: chat = 1, shat = 0, alphahat = alphabar */
            double alphahat = alphabar, chat = 1.0, shat = 0.0;
            double rhoold = rho, c, s;
            symortho(alphahat, beta, &c, &s, &rho);
            double thetanew = s * alpha; alphabar = c * alpha;
            double rhobarold = rhobar, zetaold = zeta, thetabar = sbar * rho, rhotemp = cbar * rho;
            symortho(cbar * rho, thetanew, &cbar, &sbar, &rhobar);
            zeta = cbar * zetabar; zetabar = -sbar * zetabar;
            double k1 = thetabar * rho / (rhoold * rhobarold), k2 = zeta / (rho * rhobar), k3 = thetanew / rho;
            for (int j = 0; j < n; j++) {
                hbar[j] = h[j] - k1 * hbar[j];
                x[j] = x[j] + k2 * hbar[j];
                h[j] = v[j] - k3 * h[j];
            }
            double betaacute = chat * betadd, betacheck = -shat * betadd;
            double betahat = c * betaacute; betadd = -s * betaacute;
            double thetatildeold = thetatilde, ctildeold, stildeold, rhotildeold;
            symortho(rhodold, thetabar, &ctildeold, &stildeold, &rhotildeold);
            thetatilde = stildeold * rhobar; rhodold = ctildeold * rhobar;
            betad = -stildeold * betad + ctildeold * betahat;
            tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
            double taud = (zeta - thetatilde * tautildeold) / rhodold;
            d = d + betacheck * betacheck;
            double normr = sqrt(d + (betad - taud) * (betad - taud) + betadd * betadd);
            normA2 += beta * beta;
            double normA = sqrt(normA2);
            normA2 += alpha * alpha;
            maxrbar = fmax(maxrbar, rhobarold);
            if (it > 1) minrbar = fmin(minrbar, rhobarold);
            double condA = fmax(maxrbar, rhotemp) / fmin(minrbar, rhotemp);
            double normAr = fabs(zetabar), normx = nrm2(x, n);
            double test1 = normr / normb, test2 = normAr / (normA * normr), test3 = 1 / condA;
            double t1 = test1 / (1 + normA * normx / normb), rtol = btol + atol * normA * normx / normb;
            if (it >= maxiter) break;
            if (1 + test3 <= 1 || 1 + test2 <= 1 || 1 + t1 <= 1) break;
            if (test3 <= ctol || test2 <= atol || test1 <= rtol) break;
        }
    }
    for (int j = 0; j < n; j++) dx[j] = x[j] * A.nrm[j];
    free(A.nrm); free(A.sd); free(A.tmp); free(u); free(tu); free(v); free(tv); free(h); free(hbar); free(x);
    return it;
}

/* dense Cholesky solve of the SPD system A x = b (A col-major n x n, lower
 * triangle used, overwritten).  Returns 0 or -1 if not positive definite. */
static int chol_solve(double *A, double *b, int n)
{
    for (int j = 0; j < n; j++) {
        double d = A[j + (size_t)j * n];
        for (int k = 0; k < j; k++) d -= A[j + (size_t)k * n] * A[j + (size_t)k * n];
        if (!(d > 0)) return -1;
        d = sqrt(d); A[j + (size_t)j * n] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i + (size_t)j * n];
            for (int k = 0; k < j; k++) s -= A[i + (size_t)k * n] * A[j + (size_t)k * n];
            A[i + (size_t)j * n] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i + (size_t)k * n] * b[k]; b[i] = s / A[i + (size_t)i * n]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[k + (size_t)i * n] * b[k]; b[i] = s / A[i + (size_t)i * n]; }
    return 0;
}

static void inv3_sym(const double V[6] /* xx,xy,xz,yy,yz,zz */, double I[6])
{
    double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5];
    double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
    double det = a * A + b * B + c * C, id = 1.0 / det;
    I[0] = A * id; I[1] = B * id; I[2] = C * id;
    I[3] = (a * f - c * c) * id; I[4] = (b * c - a * e) * id; I[5] = (a * d - b * b) * id;
}

/* Points in [m_begin, m_end): accumulate, for the linearisation in J / f,
 *   U (6P x 6P col-major, += Jp'Jp on the diagonal blocks, -= W V^-1 W' everywhere),
 *   g (6P, += Jp'f - W V^-1 bl), udiag (6P, += diag(Jp'Jp)),
 * where V = Jl'Jl + D_l, D_l = clamp(diag(Jl'Jl))/delta per point.  Pose
 * damping is NOT added here (it needs the complete udiag).  by_point lists the
 * observation indices of each point (CSR). */
typedef struct { int *start, *obs; } csr_t;

static void csr_by_point(const orc_ba_problem *p, csr_t *c)
{
    c->start = (int *)calloc((size_t)p->M + 1, sizeof(int));
    c->obs = (int *)malloc(sizeof(int) * (size_t)(p->O > 0 ? p->O : 1));
    for (int i = 0; i < p->O; i++) c->start[p->point_ids[i]]++;
    for (int j = 0; j < p->M; j++) c->start[j + 1] += c->start[j];
    int *fill = (int *)malloc(sizeof(int) * (size_t)(p->M + 1));
    memcpy(fill, c->start, sizeof(int) * (size_t)(p->M + 1));
    for (int i = 0; i < p->O; i++) c->obs[fill[p->point_ids[i] - 1]++] = i;
    free(fill);
}

static void point_blocks(const orc_ba_problem *p, const jac_t *J, const double *f, const csr_t *c, int j,
                         double inv_delta, double Vinv[6], double bl[3])
{
    double V[6] = {0, 0, 0, 0, 0, 0};
    bl[0] = bl[1] = bl[2] = 0;
    for (int t = c->start[j]; t < c->start[j + 1]; t++) {
        int i = c->obs[t];
        const double *jl = J->Jl + 6 * (size_t)i;
        V[0] += jl[0] * jl[0] + jl[3] * jl[3]; V[1] += jl[0] * jl[1] + jl[3] * jl[4]; V[2] += jl[0] * jl[2] + jl[3] * jl[5];
        V[3] += jl[1] * jl[1] + jl[4] * jl[4]; V[4] += jl[1] * jl[2] + jl[4] * jl[5]; V[5] += jl[2] * jl[2] + jl[5] * jl[5];
        for (int k = 0; k < 3; k++) bl[k] += jl[k] * f[2 * i] + jl[3 + k] * f[2 * i + 1];
    }
    double dd[3] = {V[0], V[3], V[5]};
    for (int k = 0; k < 3; k++) { dd[k] = fmin(fmax(dd[k], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta; }
    V[0] += dd[0]; V[3] += dd[1]; V[5] += dd[2];
    inv3_sym(V, Vinv);
}

static void reduced_accumulate(const orc_ba_problem *p, const jac_t *J, const double *f, const csr_t *c,
                               double inv_delta, int m_begin, int m_end, double *S, double *g, double *udiag)
{
    const int n = 6 * p->P;
    for (int j = m_begin; j < m_end; j++) {
        double Vi[6], bl[3];
        point_blocks(p, J, f, c, j, inv_delta, Vi, bl);
        for (int t = c->start[j]; t < c->start[j + 1]; t++) {
            int i = c->obs[t];
            if (!J->has_p[i]) continue;
            int pi = (int)p->pose_ids[i] - 1;
            const double *jp = J->Jp + 12 * (size_t)i, *jl = J->Jl + 6 * (size_t)i;
            /* U block and rhs */
            for (int a = 0; a < 6; a++) {
                for (int b = 0; b < 6; b++) S[(6 * pi + a) + (size_t)(6 * pi + b) * n] += jp[a] * jp[b] + jp[6 + a] * jp[6 + b];
                g[6 * pi + a] += jp[a] * f[2 * i] + jp[6 + a] * f[2 * i + 1];
                udiag[6 * pi + a] += jp[a] * jp[a] + jp[6 + a] * jp[6 + a];
            }
            /* W = Jp' Jl (6x3), T = W V^-1 (6x3) */
            double Wm[18], T[18];
            for (int a = 0; a < 6; a++) for (int k = 0; k < 3; k++) Wm[3 * a + k] = jp[a] * jl[k] + jp[6 + a] * jl[3 + k];
            for (int a = 0; a < 6; a++) {
                const double *w = Wm + 3 * a;
                T[3 * a] = w[0] * Vi[0] + w[1] * Vi[1] + w[2] * Vi[2];
                T[3 * a + 1] = w[0] * Vi[1] + w[1] * Vi[3] + w[2] * Vi[4];
                T[3 * a + 2] = w[0] * Vi[2] + w[1] * Vi[4] + w[2] * Vi[5];
                g[6 * pi + a] -= T[3 * a] * bl[0] + T[3 * a + 1] * bl[1] + T[3 * a + 2] * bl[2];
            }
            for (int t2 = c->start[j]; t2 < c->start[j + 1]; t2++) {
                int i2 = c->obs[t2];
                if (!J->has_p[i2]) continue;
                int qi = (int)p->pose_ids[i2] - 1;
                const double *jp2 = J->Jp + 12 * (size_t)i2, *jl2 = J->Jl + 6 * (size_t)i2;
                for (int b = 0; b < 6; b++) {
                    double w2[3];
                    for (int k = 0; k < 3; k++) w2[k] = jp2[b] * jl2[k] + jp2[6 + b] * jl2[3 + k];
                    for (int a = 0; a < 6; a++)
                        S[(6 * pi + a) + (size_t)(6 * qi + b) * n] -= T[3 * a] * w2[0] + T[3 * a + 1] * w2[1] + T[3 * a + 2] * w2[2];
                }
            }
        }
    }
}

/* exact LM step through the reduced camera system */
static int schur_solve(const orc_ba_problem *p, const jac_t *J, const double *f, const csr_t *c,
                       double inv_delta, double *dx)
{
    const int n = 6 * p->P;
    double *S = (double *)calloc((size_t)n * n, sizeof(double));
    double *g = (double *)calloc(n, sizeof(double)), *ud = (double *)calloc(n, sizeof(double));
    reduced_accumulate(p, J, f, c, inv_delta, 0, p->M, S, g, ud);
    for (int a = 0; a < n; a++) S[a + (size_t)a * n] += fmin(fmax(ud[a], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    int rc = chol_solve(S, g, n);
    if (rc == 0) {
        memcpy(dx, g, sizeof(double) * n);
        for (int j = 0; j < p->M; j++) {
            double Vi[6], bl[3];
            point_blocks(p, J, f, c, j, inv_delta, Vi, bl);
            for (int t = c->start[j]; t < c->start[j + 1]; t++) {
                int i = c->obs[t];
                if (!J->has_p[i]) continue;
                const double *jp = J->Jp + 12 * (size_t)i, *jl = J->Jl + 6 * (size_t)i;
                const double *dp = dx + 6 * (p->pose_ids[i] - 1);
                double a = 0, b = 0;
                for (int k = 0; k < 6; k++) { a += jp[k] * dp[k]; b += jp[6 + k] * dp[k]; }
                for (int k = 0; k < 3; k++) bl[k] -= jl[k] * a + jl[3 + k] * b; /* bl - W' dp */
            }
            double *dl = dx + n + 3 * j;
            dl[0] = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
            dl[1] = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
            dl[2] = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
        }
    }
    free(S); free(g); free(ud);
    return rc;
}

void orc_ba_reduced_system(const orc_ba_problem *p, const double *theta, int ignore_outliers,
                           double inv_delta, int m_begin, int m_end, double *S, double *g, double *ssr)
{
    jac_t J; csr_t c;
    J.Jp = (double *)malloc(sizeof(double) * 12 * (size_t)p->O); J.Jl = (double *)malloc(sizeof(double) * 6 * (size_t)p->O);
    J.has_p = (uint8_t *)malloc(p->O); J.active = (uint8_t *)malloc(p->O);
    double *f = (double *)malloc(sizeof(double) * 2 * (size_t)p->O);
    const int n = 6 * p->P;
    orc_ba_residuals(p, theta, ignore_outliers, f);
    jac_eval(p, theta, ignore_outliers, &J);
    csr_by_point(p, &c);
    memset(S, 0, sizeof(double) * (size_t)n * n); memset(g, 0, sizeof(double) * 2 * n); /* g: [rhs (n); udiag (n)] */
    reduced_accumulate(p, &J, f, &c, inv_delta, m_begin, m_end, S, g, g + n);
    double s = 0;
    for (int j = m_begin; j < m_end; j++)
        for (int t = c.start[j]; t < c.start[j + 1]; t++) { int i = c.obs[t]; s += f[2 * i] * f[2 * i] + f[2 * i + 1] * f[2 * i + 1]; }
    *ssr = s;
    free(J.Jp); free(J.Jl); free(J.has_p); free(J.active); free(f); free(c.start); free(c.obs);
}

/* LeastSquaresOptim optimize!(..., LevenbergMarquardt).
 * x in/out.  Returns iterations run; *ssr_out = final ssr. */
static int lm_optimize(const orc_ba_problem *p, double *x, int ignore_outliers, int iterations, int solver,
                       double *ssr_out, int64_t *inner)
{
    const int n = 6 * p->P + 3 * p->M, m = 2 * p->O;
    jac_t J; csr_t c;
    J.Jp = (double *)malloc(sizeof(double) * 12 * (size_t)p->O); J.Jl = (double *)malloc(sizeof(double) * 6 * (size_t)p->O);
    J.has_p = (uint8_t *)malloc(p->O > 0 ? p->O : 1); J.active = (uint8_t *)malloc(p->O > 0 ? p->O : 1);
    double *fcur = (double *)malloc(sizeof(double) * (m + 1)), *ftrial = (double *)malloc(sizeof(double) * (m + 1));
    double *fpred = (double *)malloc(sizeof(double) * (m + 1));
    double *dtd = (double *)malloc(sizeof(double) * n), *dx = (double *)calloc(n, sizeof(double));
    csr_by_point(p, &c);
    double delta = LM_DELTA0, decrease_factor = 2.0;
    orc_ba_residuals(p, x, ignore_outliers, fcur);
    double ssr = sumsq(fcur, m);
    int need_jacobian = 1, converged = 0, iter = 0;
    while (!converged && iter < iterations) {
        iter++;
        if (need_jacobian) { jac_eval(p, x, ignore_outliers, &J); need_jacobian = 0; }
        colsumabs2(p, &J, dtd);
        for (int j = 0; j < n; j++) dtd[j] = fmin(fmax(dtd[j], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * (1 / delta);
        if (solver == 0) *inner += lsmr_solve(p, &J, fcur, dtd, dx);
        else if (schur_solve(p, &J, fcur, &c, 1 / delta, dx) != 0) break;
        for (int j = 0; j < n; j++) x[j] -= dx[j];
        orc_ba_residuals(p, x, ignore_outliers, ftrial);
        jmul(p, &J, dx, fpred);
        for (int i = 0; i < m; i++) fpred[i] -= fcur[i];
        double predicted_ssr = sumsq(fpred, m), trial_ssr = sumsq(ftrial, m);
        double rho = (trial_ssr - ssr) / (predicted_ssr - ssr);
        double maxdx = 0;
        for (int j = 0; j < n; j++) maxdx = fmax(maxdx, fabs(dx[j]));
        if (rho > LM_MIN_STEP_QUALITY) {
            int x_conv = maxdx <= LM_XTOL;
            int f_conv = fabs(ssr - trial_ssr) / (fabs(ssr) + LM_FTOL) <= LM_FTOL;
            memcpy(fcur, ftrial, sizeof(double) * m);
            ssr = trial_ssr;
            double t = 2.0 * rho - 1.0;
            delta = fmin(delta / fmax(1.0 / 3.0, 1.0 - t * t * t), LM_MAX_DELTA);
            decrease_factor = 2.0;
            need_jacobian = 1;
            converged = x_conv || f_conv;
        } else {
            for (int j = 0; j < n; j++) x[j] += dx[j];
            delta = fmax(delta / decrease_factor, LM_MIN_DELTA);
            decrease_factor *= 2.0;
            converged = maxdx <= LM_XTOL;
        }
    }
    *ssr_out = ssr;
    free(J.Jp); free(J.Jl); free(J.has_p); free(J.active); free(fcur); free(ftrial); free(fpred); free(dtd); free(dx);
    free(c.start); free(c.obs);
    return iter;
}

/* bundle_adjustment!, bundle_adjustment.jl:1-55 */
int orc_bundle_adjustment(orc_ba_problem *p, int iters_fast, int iterations, double repr_eps,
                          int solver, orc_ba_stats *st)
{
    const int n = 6 * p->P + 3 * p->M;
    double *x = (double *)malloc(sizeof(double) * n);
    memcpy(x, p->theta, sizeof(double) * n);
    orc_ba_stats s; memset(&s, 0, sizeof s);
    double *Y = (double *)malloc(sizeof(double) * (2 * (size_t)p->O + 1));
    orc_ba_residuals(p, x, 0, Y);
    s.ssr_init = sumsq(Y, 2 * (size_t)p->O);
    free(Y);
    s.iters_pass1 = lm_optimize(p, x, 0, iters_fast, solver, &s.ssr_pass1, &s.inner_iters);   /* :41-44 */
    s.n_outliers = orc_ba_detect_outliers(p, x, repr_eps, 1e-6);                                /* :45 */
    s.iters_pass2 = lm_optimize(p, x, 1, iterations, solver, &s.ssr_final, &s.inner_iters);    /* :48-53 */
    memcpy(p->theta, x, sizeof(double) * n);                                                     /* :54 */
    free(x);
    if (st) *st = s;
    return 0;
}

int orc_bundle_adjustment_flat(double fx, double fy, double cx, double cy, int P, int M, int O,
                               double *theta, const uint8_t *theta_const, const double *pixels_yx,
                               const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                               int iters_fast, int iterations, double repr_eps, int solver, double *so)
{
    orc_ba_problem p = {fx, fy, cx, cy, P, M, O, theta, theta_const, pixels_yx, pose_ids, point_ids, outliers};
    orc_ba_stats s;
    int rc = orc_bundle_adjustment(&p, iters_fast, iterations, repr_eps, solver, &s);
    if (so) { so[0] = s.ssr_init; so[1] = s.ssr_pass1; so[2] = s.ssr_final; so[3] = s.iters_pass1; so[4] = s.iters_pass2;
              so[5] = s.n_outliers; so[6] = (double)s.inner_iters; so[7] = 0; }
    return rc;
}

/* ---------------- pnp_bundle_adjustment, bundle_adjustment.jl:113-171 ----- */
static void pnp_residuals(const double *X, const double *px, const double *pts, int n, const uint8_t *outl, int ignore,
                          double fx, double fy, double cx, double cy, double *Y, double *Jd /* 2n x 6 row-major or NULL */)
{
    for (int i = 0; i < n; i++) {
        if (ignore && outl[i]) { Y[2 * i] = Y[2 * i + 1] = 0; if (Jd) memset(Jd + 12 * (size_t)i, 0, 12 * sizeof(double)); continue; }
        double Jl[6];
        obs_eval(X, pts + 3 * i, px[2 * i], px[2 * i + 1], fx, fy, cx, cy, Y + 2 * i, Jd ? Jd + 12 * (size_t)i : NULL, Jd ? Jl : NULL, NULL);
    }
}

static int pnp_lm(double *X, const double *px, const double *pts, int n, const uint8_t *outl, int ignore,
                  double fx, double fy, double cx, double cy, int iterations, double *ssr_out)
{
    const int m = 2 * n;
    double *fcur = (double *)malloc(sizeof(double) * (m + 1)), *ftrial = (double *)malloc(sizeof(double) * (m + 1));
    double *Jd = (double *)malloc(sizeof(double) * 12 * (size_t)(n + 1));
    double delta = LM_DELTA0, decrease_factor = 2.0;
    pnp_residuals(X, px, pts, n, outl, ignore, fx, fy, cx, cy, fcur, NULL);
    double ssr = sumsq(fcur, m);
    int need_jacobian = 1, converged = 0, iter = 0;
    while (!converged && iter < iterations) {
        iter++;
        if (need_jacobian) { pnp_residuals(X, px, pts, n, outl, ignore, fx, fy, cx, cy, ftrial, Jd); need_jacobian = 0; }
        double H[36], g[6], dtd[6];
        memset(H, 0, sizeof H); memset(g, 0, sizeof g);
        for (int i = 0; i < n; i++) {
            const double *jp = Jd + 12 * (size_t)i;
            for (int a = 0; a < 6; a++) {
                for (int b = 0; b < 6; b++) H[a + 6 * b] += jp[a] * jp[b] + jp[6 + a] * jp[6 + b];
                g[a] += jp[a] * fcur[2 * i] + jp[6 + a] * fcur[2 * i + 1];
            }
        }
        for (int a = 0; a < 6; a++) { dtd[a] = fmin(fmax(H[a + 6 * a], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * (1 / delta); H[a + 6 * a] += dtd[a]; }
        double dx[6]; memcpy(dx, g, sizeof dx);
        if (chol_solve(H, dx, 6) != 0) break;
        for (int a = 0; a < 6; a++) X[a] -= dx[a];
        pnp_residuals(X, px, pts, n, outl, ignore, fx, fy, cx, cy, ftrial, NULL);
        double predicted_ssr = 0, trial_ssr = sumsq(ftrial, m), maxdx = 0;
        for (int i = 0; i < n; i++) {
            const double *jp = Jd + 12 * (size_t)i;
            double a = 0, b = 0;
            for (int k = 0; k < 6; k++) { a += jp[k] * dx[k]; b += jp[6 + k] * dx[k]; }
            a -= fcur[2 * i]; b -= fcur[2 * i + 1];
            predicted_ssr += a * a; predicted_ssr += b * b;
        }
        for (int a = 0; a < 6; a++) maxdx = fmax(maxdx, fabs(dx[a]));
        double rho = (trial_ssr - ssr) / (predicted_ssr - ssr);
        if (rho > LM_MIN_STEP_QUALITY) {
            int x_conv = maxdx <= LM_XTOL;
            int f_conv = fabs(ssr - trial_ssr) / (fabs(ssr) + LM_FTOL) <= LM_FTOL;
            memcpy(fcur, ftrial, sizeof(double) * m);
            ssr = trial_ssr;
            double t = 2.0 * rho - 1.0;
            delta = fmin(delta / fmax(1.0 / 3.0, 1.0 - t * t * t), LM_MAX_DELTA);
            decrease_factor = 2.0; need_jacobian = 1;
            converged = x_conv || f_conv;
        } else {
            for (int a = 0; a < 6; a++) X[a] += dx[a];
            delta = fmax(delta / decrease_factor, LM_MIN_DELTA);
            decrease_factor *= 2.0;
            converged = maxdx <= LM_XTOL;
        }
    }
    *ssr_out = ssr;
    free(fcur); free(ftrial); free(Jd);
    return iter;
}

int orc_pnp_ba(double fx, double fy, double cx, double cy, const double pose[16],
               const double *px, const double *pts, int n, int iters_fast, int iterations,
               double depth_eps, double repr_eps, double out_pose[16],
               double *err_init, double *err_final, uint8_t *outliers, int *n_outliers)
{
    /* pose is 4x4 column-major: R[i][j] = pose[i + 4*j] */
    double R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = pose[i + 4 * j];
    double X[6];
    orc_rotzyx_angles(R, &X[0], &X[1], &X[2]);
    X[3] = pose[12]; X[4] = pose[13]; X[5] = pose[14];
    for (int i = 0; i < n; i++) outliers[i] = 0;
    double *Y = (double *)malloc(sizeof(double) * (2 * (size_t)n + 1));
    pnp_residuals(X, px, pts, n, outliers, 0, fx, fy, cx, cy, Y, NULL);
    *err_init = sumsq(Y, 2 * (size_t)n);
    double ssr1, ssr2;
    pnp_lm(X, px, pts, n, outliers, 0, fx, fy, cx, cy, iters_fast, &ssr1);
    int no = 0;
    for (int i = 0; i < n; i++) {
        double r[2], z;
        obs_eval(X, pts + 3 * i, px[2 * i], px[2 * i + 1], fx, fy, cx, cy, r, NULL, NULL, &z);
        int o = z < depth_eps || (r[0] * r[0] + r[1] * r[1]) > repr_eps;
        outliers[i] = (uint8_t)o; no += o;
    }
    *n_outliers = no;
    free(Y);
    if (n - no < 5) {
        for (int k = 0; k < 16; k++) out_pose[k] = (k % 5 == 0) ? 1.0 : 0.0;
        *err_final = ssr1;
        return 0;
    }
    pnp_lm(X, px, pts, n, outliers, 1, fx, fy, cx, cy, iterations, &ssr2);
    orc_rotzyx(X[0], X[1], X[2], R);
    memset(out_pose, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out_pose[i + 4 * j] = R[3 * i + j];
    out_pose[12] = X[3]; out_pose[13] = X[4]; out_pose[14] = X[5]; out_pose[15] = 1.0;
    *err_final = ssr2;
    return 0;
}
