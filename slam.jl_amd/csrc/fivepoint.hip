// fivepoint.hip -- five-point essential-matrix RANSAC of compute_pose_5pt! (src/front_end.jl:243-332; the call
// five_point_ransac(previous_points, current_points, previous_pd, current_pd, K, K, cache; max_repr_error) at :305-308).
//
// Three kernels on the context's stream:
//   k_5pt_solve   one 16-lane team per caller-supplied 5-tuple (lane 0 owns the sequential phases, the team shares
//                 the root search, one bracketing interval per lane): Nister's minimal solver (null space of the 5x9 system,
//                 the ten cubics by polynomial arithmetic, Gauss-Jordan in Nister's monomial order, det B(z) of
//                 degree 10, real roots bracketed between the derivative's roots), then Horn's closed-form pose of
//                 each essential matrix with cheirality on the five sample points.  Only + - * / sqrt: the
//                 hypotheses are bit-identical to the CPU statement.  The 10x20 elimination matrix and the
//                 derivative table live in LDS, one column of doubles per thread.
//   k_5pt_score   one 256-thread workgroup per (5-tuple, pose): the threads stride over the correspondences -- DLT
//                 triangulation (the mapper's 4x4 system, eigenvector by inverse iteration), both depths > 0, both reprojection
//                 errors < max_repr_error -- and a butterfly + LDS add the inlier counts.
//   k_5pt_select  one workgroup: winner (most inliers, ties to the earlier tuple, then root), inlier mask, summed
//                 error, E and [R | t].
// The solver is a latency chain (0.2 ms for any tuple count that fits the GPU); scoring is ~iters x 6 x n triangulations
// of ~0.4 kflop each.
#include "common.hpp"
#include "tri_device.hpp"
#include <cmath>

#define FP_TEAM 16                      // lanes per 5-tuple in the solver (one bracketing interval each in the root search)
#define FP_TPB 4                        // 5-tuples per solver workgroup (one 64-lane wave)
#define FP_MAXE 10
#define FP_SCORE_T 256                 // scoring threads per (5-tuple, pose)
#define FP_SEL_T 512                   // threads of the select kernel
#define FP_ERR_LDS 4096                 // correspondences whose errors the select kernel stages in LDS

struct FPArgs {                            // S independent problems (S = 1: slam_five_point_ransac); problem z owns [off[z], off[z+1])
    const double *px1, *px2, *pd1, *pd2;   // concatenated, n x 2 each, (x, y)
    const int32_t *samples;                // S x iters x 5, 0-based, local to the problem
    const int *off;                        // S + 1
    const int *cnt; int stride;            // keypoint-set layout instead (cnt != nullptr): problem z owns [z * stride, z * stride + cnt[z])
    const double *ks;                      // S x 8: fx, fy, cx, cy of camera 1 then of camera 2
    int iters;
    double thr;
    int *ne;                               // S x iters: number of poses of the tuple
    double *Es;                            // S x iters x 10 x 9 (row-major E)
    double *poses;                         // S x iters x 10 x 12
    int *counts;                           // S x iters x 10
    double *errs;                          // off[S]
    double *out;                           // S x 32 (mapped host): P 12 | E 9 | error | {n_inliers, best_iter} as two ints
    uint8_t *inliers;                      // off[S] (mapped host)
};

// per-thread array in LDS: element i of thread t at base[i * FP_TPB + t]
// (the pointer carries the LDS address space: as a generic pointer every access was a FLAT instruction -- slower than ds_read /
//  ds_write and, worse, a full vmcnt(0) + lgkmcnt(0) wait each)
typedef __attribute__((address_space(3))) double fp_lds;
struct Col {
    fp_lds *b;
    __device__ fp_lds &operator[](int i) const { return b[i * FP_TPB]; }
    __device__ Col at(int off) const { return Col{b + off * FP_TPB}; }
};

static __device__ const int c_T2[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
static __device__ const int c_T3[10][4] = {{0, 2, 4, 5}, {2, 3, 8, 9}, {4, 8, 10, 11}, {3, 1, 6, 7}, {8, 6, 13, 14},
                                {10, 13, 16, 17}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};

template <class A, class B, class O> __device__ static inline void mul11(const A &a, const B &b, O &o)
{
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) o[c_T2[i][j]] += a[i] * b[j];
}
template <class A, class B, class O> __device__ static inline void mul21(const A &a, const B &b, O &o)
{
    for (int i = 0; i < 10; i++) for (int j = 0; j < 4; j++) o[c_T3[i][j]] += a[i] * b[j];
}
template <class A> __device__ static inline void minor2(const A &a, const A &b, const A &c, const A &d, double *o)
{
    double t1[10], t2[10];
    for (int i = 0; i < 10; i++) { t1[i] = 0.0; t2[i] = 0.0; }
    mul11(a, b, t1); mul11(c, d, t2);
    for (int i = 0; i < 10; i++) o[i] = t1[i] - t2[i];
}
__device__ static inline void pmul(const double *a, int na, const double *b, int nb, double *o)
{
    for (int i = 0; i < na + nb - 1; i++) o[i] = 0.0;
    for (int i = 0; i < na; i++) for (int j = 0; j < nb; j++) o[i + j] += a[i] * b[j];
}
template <class P> __device__ static inline double peval(const P &p, int deg, double x)
{
    double f = p[deg];
    for (int i = deg - 1; i >= 0; i--) f = f * x + p[i];
    return f;
}

// Horner with the coefficients in registers, degree known at compile time (same operations as peval)
template <int D> __device__ static inline double peval_d(const double (&p)[11], double x)
{
    double f = p[D];
#pragma unroll
    for (int i = D - 1; i >= 0; i--) f = f * x + p[i];
    return f;
}

template <int D> __device__ static double bracket_root(const double (&p)[11], const double (&dp)[11], double lo, double hi, double flo)
{
    double x = 0.5 * (lo + hi);
    for (int it = 0; it < 200; it++) {
        const double f = peval_d<D>(p, x);
        if (f == 0.0) return x;
        if ((f < 0.0) == (flo < 0.0)) lo = x; else hi = x;
        const double df = peval_d<D - 1>(dp, x);
        double xn = x - f / df;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (xn == x || xn == lo || xn == hi) return xn;
        x = xn;
    }
    return x;
}

// Owner part of the root finder: derivative table D (11 x 11, row k = k-th derivative) and the root of the linear
// derivative in crit = D.at(121); the effective degree goes to D[143] (0: nothing to solve).
__device__ static void roots_prepare(const double *p, int deg, Col D)
{
    D[143] = 0.0;
    while (deg > 0 && p[deg] == 0.0) deg--;
    if (deg <= 0 || deg > 10) return;
    for (int i = 0; i <= deg; i++) D[i] = p[i];
    for (int k = 1; k < deg; k++)
        for (int i = 0; i <= deg - k; i++) D[11 * k + i] = (double)(i + 1) * D[11 * (k - 1) + i + 1];
    const double c0 = -D[11 * (deg - 1)] / D[11 * (deg - 1) + 1];
    D[121] = c0;
    if (isfinite(c0)) D[143] = (double)deg;
}

// One lane's interval j of the degree-D polynomial in row k of the table: exactly the sequential statement's
// operations for that interval.  nc critical points in crit.
template <int D> __device__ static void interval_root(Col Dt, int k, int j, int nc, bool &found, bool &fail, double &root)
{
    const Col crit = Dt.at(121);
    double q[11], dq[11];
#pragma unroll
    for (int i = 0; i <= D; i++) q[i] = Dt[11 * k + i];
#pragma unroll
    for (int i = 0; i < D; i++) dq[i] = Dt[11 * (k + 1) + i];
    double bound = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) { const double c = fabs(q[i] / q[D]); if (c > bound) bound = c; }
    bound = bound + 1.0;
    if (!isfinite(bound)) { fail = true; return; }
    double lo = -bound;                                   // the sequential scan's `lo`: running maximum of the accepted ends
    for (int i = 0; i < j; i++) { const double c = crit[i]; if (c > lo) lo = c; }
    const double hi = j < nc ? crit[j] : bound;
    if (!(hi > lo)) return;
    const double flo = peval_d<D>(q, lo), fhi = peval_d<D>(q, hi);
    if (flo != 0.0 && fhi != 0.0 && (flo < 0.0) != (fhi < 0.0)) {
        // an end that is only the Cauchy bound can be astronomically far from the root: walk towards it from the
        // finite end with doubling steps until the sign change is enclosed
        double a = lo, fa = flo, b = hi;
        bool a_far = j == 0, b_far = j == nc, hit = false;
        if (a_far && b_far) {
            const double f0 = peval_d<D>(q, 0.0);
            if (f0 == 0.0) hit = true;
            else if ((f0 < 0.0) == (fa < 0.0)) { a = 0.0; fa = f0; a_far = false; }
            else { b = 0.0; b_far = false; }
        }
        if (!hit && a_far) {
            double h = fabs(b) > 1.0 ? fabs(b) : 1.0;
            for (int it = 0; it < 1100; it++) {
                const double x = b - h;
                if (!(x > a)) break;
                const double fx = peval_d<D>(q, x);
                if (fx == 0.0) { hit = true; root = x; break; }
                if ((fx < 0.0) == (fa < 0.0)) { a = x; fa = fx; break; }
                b = x; h = 2.0 * h;
            }
        } else if (!hit && b_far) {
            double h = fabs(a) > 1.0 ? fabs(a) : 1.0;
            for (int it = 0; it < 1100; it++) {
                const double x = a + h;
                if (!(x < b)) break;
                const double fx = peval_d<D>(q, x);
                if (fx == 0.0) { hit = true; root = x; break; }
                if ((fx < 0.0) != (fa < 0.0)) { b = x; break; }
                a = x; fa = fx; h = 2.0 * h;
            }
        }
        if (!hit) root = bracket_root<D>(q, dq, a, b, fa);
        found = true;
    }
    else if (fhi == 0.0 && j < nc) { found = true; root = hi; }
}

// Team part: the FP_TEAM lanes of a 5-tuple take one bracketing interval each, level by level (the roots of the
// (k+1)-th derivative bracket those of the k-th).  Every lane performs exactly the operations the sequential
// statement performs for its interval, and the roots are compacted in interval order, so the result is the same
// list.  Must be called by all threads of the workgroup (it synchronises).  Returns the number of real roots of
// the polynomial, left ascending in crit = D.at(121).
__device__ static int team_real_roots(Col D, int l, int team)
{
    const Col crit = D.at(121);
    int deg = (int)D[143];
    int nc = deg > 0 ? 1 : 0;
    for (int k = 8; k >= 0; k--) {
        const bool act = deg >= 2 && k <= deg - 2;        // team-uniform
        bool found = false, fail = false;
        double root = 0.0;
        if (act && l <= nc) {
            switch (deg - k) {                            // degree of this level's polynomial (team-uniform)
            case 2: interval_root<2>(D, k, l, nc, found, fail, root); break;
            case 3: interval_root<3>(D, k, l, nc, found, fail, root); break;
            case 4: interval_root<4>(D, k, l, nc, found, fail, root); break;
            case 5: interval_root<5>(D, k, l, nc, found, fail, root); break;
            case 6: interval_root<6>(D, k, l, nc, found, fail, root); break;
            case 7: interval_root<7>(D, k, l, nc, found, fail, root); break;
            case 8: interval_root<8>(D, k, l, nc, found, fail, root); break;
            case 9: interval_root<9>(D, k, l, nc, found, fail, root); break;
            default: interval_root<10>(D, k, l, nc, found, fail, root); break;
            }
        }
        const unsigned long long mf = __ballot(found), mx = __ballot(fail);
        const unsigned tf = (unsigned)(mf >> (FP_TEAM * team)) & ((1u << FP_TEAM) - 1u);
        const bool tfail = ((mx >> (FP_TEAM * team)) & ((1ull << FP_TEAM) - 1ull)) != 0;
        __syncthreads();                                  // every lane has read crit
        if (found && !tfail) crit[__popc(tf & ((1u << l) - 1u))] = root;
        if (act) nc = tfail ? 0 : __popc(tf);
        if (tfail) deg = 0;
        __syncthreads();
    }
    return nc;
}

// A: 5 x 9 (LDS), N: 4 x 9 (LDS)
__device__ static bool nullspace_5x9(Col A, Col N)
{
    int piv[5]; unsigned used = 0;
    for (int s = 0; s < 5; s++) {
        int pr = -1, pc = -1; double best = 0.0;
        for (int r = s; r < 5; r++)
            for (int c = 0; c < 9; c++)
                if (!((used >> c) & 1) && fabs(A[9 * r + c]) > best) { best = fabs(A[9 * r + c]); pr = r; pc = c; }
        if (pr < 0) return false;
        if (pr != s) for (int c = 0; c < 9; c++) { const double t = A[9 * s + c]; A[9 * s + c] = A[9 * pr + c]; A[9 * pr + c] = t; }
        piv[s] = pc; used |= 1u << pc;
        const double inv = 1.0 / A[9 * s + pc];
        for (int c = 0; c < 9; c++) A[9 * s + c] *= inv;
        A[9 * s + pc] = 1.0;
        for (int r = 0; r < 5; r++) {
            if (r == s) continue;
            const double f = A[9 * r + pc];
            if (f == 0.0) continue;
            for (int c = 0; c < 9; c++) A[9 * r + c] -= f * A[9 * s + c];
            A[9 * r + pc] = 0.0;
        }
    }
    int k = 0;
    for (int f = 0; f < 9; f++) {
        if ((used >> f) & 1) continue;
        for (int c = 0; c < 9; c++) N[9 * k + c] = 0.0;
        N[9 * k + f] = 1.0;
        for (int s = 0; s < 5; s++) {
            int pc = 0;
            for (int t = 0; t < 5; t++) if (t == s) pc = piv[t];       // piv stays in registers (no dynamic indexing)
            N[9 * k + pc] = -A[9 * s + f];
        }
        k++;
    }
    for (int a = 0; a < 4; a++) {
        for (int b = 0; b < a; b++) {
            double d = 0.0;
            for (int c = 0; c < 9; c++) d += N[9 * a + c] * N[9 * b + c];
            for (int c = 0; c < 9; c++) N[9 * a + c] -= d * N[9 * b + c];
        }
        double nn = 0.0;
        for (int c = 0; c < 9; c++) nn += N[9 * a + c] * N[9 * a + c];
        if (!(nn > 0.0)) return false;
        const double inv = 1.0 / sqrt(nn);
        for (int c = 0; c < 9; c++) N[9 * a + c] *= inv;
    }
    return true;
}

// LDS per 5-tuple (doubles): M 200 | N 36 | Ep 36 | W 144 (A 45, then EEt 90 + tr 10, then D 121 + crit 11 + spare 11 + degree 1)
#define FP_LDS_PER_THREAD (200 + 36 + 36 + 144 + 9 * FP_MAXE)   /* + the essential matrices of the tuple (indexed by a run-time root count: LDS, not scratch) */

// Called by all threads of the workgroup (it synchronises).  Lane l == 0 of each team owns the short sequential
// phases; the cubic constraints, the Gauss-Jordan elimination and the root search are shared by the team's lanes --
// one output polynomial / one matrix row / one bracketing interval per lane, each computed with exactly the
// operations of the sequential statement.  Returns (to the owner) the number of essential matrices written to Es.
__device__ static int five_point_solve(const double *q1, const double *q2, Col L, int l, int team, bool valid)
{
    Col M = L, N = L.at(200), Ep = L.at(236), W = L.at(272), Es = L.at(416);
    const bool owner = l == 0;
    // W[142]: 1.0 while the tuple is alive; W[143]: degree for the root search
    if (owner) {
        W[143] = 0.0;
        bool ok = valid;
        if (ok) {
            Col A = W;
            for (int i = 0; i < 5; i++) {
                const double x = q1[2 * i], y = q1[2 * i + 1], u = q2[2 * i], v = q2[2 * i + 1];
                A[9 * i] = u * x; A[9 * i + 1] = u * y; A[9 * i + 2] = u; A[9 * i + 3] = v * x; A[9 * i + 4] = v * y; A[9 * i + 5] = v;
                A[9 * i + 6] = x; A[9 * i + 7] = y; A[9 * i + 8] = 1.0;
            }
            ok = nullspace_5x9(A, N);
        }
        if (ok) for (int e = 0; e < 9; e++) for (int m = 0; m < 4; m++) Ep[4 * e + m] = N[9 * m + e];
        W[142] = ok ? 1.0 : 0.0;
    }
    __syncthreads();
    bool alive = W[142] != 0.0;
    {   // E E' (nine quadratics, one per lane), tr/2, the nine cubics (E E' - tr/2 I) E and det E (one per lane)
        Col EEt = W, tr = W.at(90);
        if (alive && l < 9) {
            const int r = l / 3, c = l % 3;
            Col o = EEt.at(10 * l);
            for (int i = 0; i < 10; i++) o[i] = 0.0;
            for (int k = 0; k < 3; k++) { const Col a = Ep.at(4 * (3 * r + k)), b = Ep.at(4 * (3 * c + k)); mul11(a, b, o); }
        }
        __syncthreads();
        if (alive && l < 10) tr[l] = 0.5 * ((EEt[l] + EEt[40 + l]) + EEt[80 + l]);
        __syncthreads();
        if (alive && l < 10) for (int d = 0; d < 3; d++) EEt[40 * d + l] -= tr[l];
        __syncthreads();
        if (alive && l < 9) {
            const int r = l / 3, c = l % 3;
            Col o = M.at(20 * l);
            for (int i = 0; i < 20; i++) o[i] = 0.0;
            for (int k = 0; k < 3; k++) { const Col a = EEt.at(10 * (3 * r + k)), b = Ep.at(4 * (3 * k + c)); mul21(a, b, o); }
        }
        if (alive && l == 9) {
            double m0[10], m1[10], m2[10], t[20];
            minor2(Ep.at(16), Ep.at(32), Ep.at(20), Ep.at(28), m0);
            minor2(Ep.at(12), Ep.at(32), Ep.at(20), Ep.at(24), m1);
            minor2(Ep.at(12), Ep.at(28), Ep.at(16), Ep.at(24), m2);
            Col o = M.at(180);
            for (int i = 0; i < 20; i++) o[i] = 0.0;
            { const Col e0 = Ep.at(0); mul21(m0, e0, o); }
            for (int i = 0; i < 20; i++) t[i] = 0.0;
            { const Col e1 = Ep.at(4); mul21(m1, e1, t); }
            for (int i = 0; i < 20; i++) o[i] -= t[i];
            for (int i = 0; i < 20; i++) t[i] = 0.0;
            { const Col e2 = Ep.at(8); mul21(m2, e2, t); }
            for (int i = 0; i < 20; i++) o[i] += t[i];
        }
        __syncthreads();
    }
    for (int s = 0; s < 10; s++) {                      // Gauss-Jordan on the first ten columns, row pivoting
        int pr = s; double best = 0.0;
        if (alive) {
            best = fabs(M[20 * s + s]);
            for (int r = s + 1; r < 10; r++) if (fabs(M[20 * r + s]) > best) { best = fabs(M[20 * r + s]); pr = r; }
            if (!(best > 0.0)) alive = false;             // every lane of the team reads the same column
        }
        __syncthreads();
        if (alive && pr != s)
            for (int c = l; c < 20; c += FP_TEAM) { const double t = M[20 * s + c]; M[20 * s + c] = M[20 * pr + c]; M[20 * pr + c] = t; }
        __syncthreads();
        const double inv = alive ? 1.0 / M[20 * s + s] : 0.0;
        __syncthreads();
        if (alive) for (int c = l; c < 20; c += FP_TEAM) M[20 * s + c] = c == s ? 1.0 : M[20 * s + c] * inv;
        __syncthreads();
        if (alive && l < 10 && l != s) {
            const int r = l;
            const double f = M[20 * r + s];
            if (f != 0.0) {
                for (int c = 0; c < 20; c++) M[20 * r + c] -= f * M[20 * s + c];
                M[20 * r + s] = 0.0;
            }
        }
        __syncthreads();
    }
    double Ba[3][4], Bb[3][4], Bc[3][5];
    if (owner && alive) {
        for (int r = 0; r < 3; r++) {
            const Col e = M.at(20 * (4 + 2 * r)), f = M.at(20 * (5 + 2 * r));
            Ba[r][0] = e[12]; Ba[r][1] = e[11] - f[12]; Ba[r][2] = e[10] - f[11]; Ba[r][3] = -f[10];
            Bb[r][0] = e[15]; Bb[r][1] = e[14] - f[15]; Bb[r][2] = e[13] - f[14]; Bb[r][3] = -f[13];
            Bc[r][0] = e[19]; Bc[r][1] = e[18] - f[19]; Bc[r][2] = e[17] - f[18]; Bc[r][3] = e[16] - f[17]; Bc[r][4] = -f[16];
        }
        double P[11];
        {
            double t1[8], t2[8], u[8], w[11];
            pmul(Bb[1], 4, Bc[2], 5, t1); pmul(Bc[1], 5, Bb[2], 4, t2);
            for (int i = 0; i < 8; i++) u[i] = t1[i] - t2[i];
            pmul(Ba[0], 4, u, 8, P);
            pmul(Ba[1], 4, Bc[2], 5, t1); pmul(Bc[1], 5, Ba[2], 4, t2);
            for (int i = 0; i < 8; i++) u[i] = t1[i] - t2[i];
            pmul(Bb[0], 4, u, 8, w);
            for (int i = 0; i < 11; i++) P[i] -= w[i];
            pmul(Ba[1], 4, Bb[2], 4, t1); pmul(Bb[1], 4, Ba[2], 4, t2);
            for (int i = 0; i < 7; i++) u[i] = t1[i] - t2[i];
            pmul(Bc[0], 5, u, 7, w);
            for (int i = 0; i < 11; i++) P[i] += w[i];
        }
        bool fin = true;
        for (int i = 0; i < 11; i++) fin = fin && isfinite(P[i]);
        if (fin) roots_prepare(P, 10, W);
    }
    __syncthreads();
    const int nr = team_real_roots(W, l, team);
    if (!owner || !alive) return 0;
    const Col zr = W.at(121);
    int ne = 0;
    for (int i = 0; i < nr; i++) {
        const double z = zr[i];
        double R[3][3];
        for (int r = 0; r < 3; r++) { R[r][0] = peval(Ba[r], 3, z); R[r][1] = peval(Bb[r], 3, z); R[r][2] = peval(Bc[r], 4, z); }
        double best = 0.0, nx = 0.0, ny = 0.0, nz = 0.0;
        for (int a = 0; a < 2; a++)
            for (int b = a + 1; b < 3; b++) {
                const double c0 = R[a][1] * R[b][2] - R[a][2] * R[b][1];
                const double c1 = R[a][2] * R[b][0] - R[a][0] * R[b][2];
                const double c2 = R[a][0] * R[b][1] - R[a][1] * R[b][0];
                if (fabs(c2) > best) { best = fabs(c2); nx = c0; ny = c1; nz = c2; }
            }
        if (!(best > 0.0)) continue;
        const double x = nx / nz, y = ny / nz;
        bool fin = true;
        for (int e = 0; e < 9; e++) {
            const double ev = ((x * N[e] + y * N[9 + e]) + z * N[18 + e]) + N[27 + e];
            Es[9 * ne + e] = ev;
            fin = fin && isfinite(ev);
        }
        if (fin) ne++;
    }
    return ne;
}

__device__ static bool essential_poses(const double *E, double *Rt)
{
    double G[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) G[3 * r + c] = (E[3 * r] * E[3 * c] + E[3 * r + 1] * E[3 * c + 1]) + E[3 * r + 2] * E[3 * c + 2];
    const double h = 0.5 * ((G[0] + G[4]) + G[8]);
    for (int i = 0; i < 9; i++) G[i] = -G[i];
    G[0] += h; G[4] += h; G[8] += h;
    int m = 0;
    if (G[4] > G[0]) m = 1;
    if (G[8] > (m == 1 ? G[4] : G[0])) m = 2;
    const double gm = m == 0 ? G[0] : (m == 1 ? G[4] : G[8]);
    if (!(gm > 0.0)) return false;
    const double s = 1.0 / sqrt(gm);
    double b[3];
    for (int j = 0; j < 3; j++) b[j] = (m == 0 ? G[j] : (m == 1 ? G[3 + j] : G[6 + j])) * s;
    const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    if (!(bb > 0.0) || !isfinite(bb)) return false;
    double C[9], BE[9];
    for (int r = 0; r < 3; r++) {
        const double *p = E + 3 * ((r + 1) % 3), *q = E + 3 * ((r + 2) % 3);
        C[3 * r] = p[1] * q[2] - p[2] * q[1]; C[3 * r + 1] = p[2] * q[0] - p[0] * q[2]; C[3 * r + 2] = p[0] * q[1] - p[1] * q[0];
    }
    for (int c = 0; c < 3; c++) {
        BE[c] = b[1] * E[6 + c] - b[2] * E[3 + c];
        BE[3 + c] = b[2] * E[c] - b[0] * E[6 + c];
        BE[6 + c] = b[0] * E[3 + c] - b[1] * E[c];
    }
    const double ib = 1.0 / bb, in = 1.0 / sqrt(bb);
    const double t[3] = {b[0] * in, b[1] * in, b[2] * in};
    for (int k = 0; k < 4; k++) {
        double *P = Rt + 12 * k;
        const double sg = (k & 1) ? -1.0 : 1.0;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++)
                P[r + 3 * c] = (k < 2 ? C[3 * r + c] - BE[3 * r + c] : C[3 * r + c] + BE[3 * r + c]) * ib;
            P[9 + r] = sg * t[r];
        }
    }
    return true;
}

__device__ static inline bool tri_two_view(const double *k1, const double *k2, const double *Rt, const double *a, const double *b, double *X, double *Y)
{
    double P1[12], P2[12];
    for (int i = 0; i < 12; i++) P1[i] = 0.0;
    P1[0] = k1[0]; P1[2] = k1[2]; P1[5] = k1[1]; P1[6] = k1[3]; P1[10] = 1.0;
    for (int c = 0; c < 4; c++) {
        const double r0 = Rt[3 * c], r1 = Rt[3 * c + 1], r2 = Rt[3 * c + 2];
        P2[c] = k2[0] * r0 + k2[2] * r2; P2[4 + c] = k2[1] * r1 + k2[3] * r2; P2[8 + c] = r2;
    }
    double A[16], S[16], v[4];
    for (int j = 0; j < 4; j++) {
        A[j] = a[0] * P1[8 + j] - P1[j]; A[4 + j] = a[1] * P1[8 + j] - P1[4 + j];
        A[8 + j] = b[0] * P2[8 + j] - P2[j]; A[12 + j] = b[1] * P2[8 + j] - P2[4 + j];
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += A[4 * k + i] * A[4 * k + j];
            S[4 * i + j] = acc;
        }
    sym4_min_eigvec_invit(S, v);
    const double iw = 1.0 / v[3];
    X[0] = v[0] * iw; X[1] = v[1] * iw; X[2] = v[2] * iw;
    for (int r = 0; r < 3; r++) Y[r] = ((Rt[r] * X[0] + Rt[3 + r] * X[1]) + Rt[6 + r] * X[2]) + Rt[9 + r];
    return isfinite(X[0]) && isfinite(X[1]) && isfinite(X[2]);
}

__device__ static inline bool two_view_errors(const double *k1, const double *k2, const double *Rt, const double *a, const double *b, double *e1, double *e2)
{
    double X[3], Y[3];
    if (!tri_two_view(k1, k2, Rt, a, b, X, Y)) return false;
    if (!(X[2] > 0.0) || !(Y[2] > 0.0)) return false;
    const double i1 = 1.0 / X[2], i2 = 1.0 / Y[2];
    const double dx1 = a[0] - (k1[0] * X[0] * i1 + k1[2]), dy1 = a[1] - (k1[1] * X[1] * i1 + k1[3]);
    const double dx2 = b[0] - (k2[0] * Y[0] * i2 + k2[2]), dy2 = b[1] - (k2[1] * Y[1] * i2 + k2[3]);
    *e1 = sqrt(dx1 * dx1 + dy1 * dy1); *e2 = sqrt(dx2 * dx2 + dy2 * dy2);
    return true;
}

// depths along the two rays under x2 = R x1 + t (normalised coordinates), see the oracle's ray_depths
__device__ static inline void ray_depths(const double *Rt, const double *q1, const double *q2, double *l1, double *l2)
{
    const double r0 = (Rt[0] * q1[0] + Rt[3] * q1[1]) + Rt[6], r1 = (Rt[1] * q1[0] + Rt[4] * q1[1]) + Rt[7],
                 r2 = (Rt[2] * q1[0] + Rt[5] * q1[1]) + Rt[8];
    const double a0 = q2[1] * r2 - r1, a1 = r0 - q2[0] * r2, a2 = q2[0] * r1 - q2[1] * r0;
    const double b0 = q2[1] * Rt[11] - Rt[10], b1 = Rt[9] - q2[0] * Rt[11], b2 = q2[0] * Rt[10] - q2[1] * Rt[9];
    const double num = (a0 * b0 + a1 * b1) + a2 * b2, den = (a0 * a0 + a1 * a1) + a2 * a2;
    *l1 = -num / den;
    *l2 = *l1 * r2 + Rt[11];
}

// (three waves per SIMD: left to itself the allocator takes 244 registers -- one or two waves per SIMD for a kernel that is a latency chain of eliminations and
//  root brackets; capped at 168 it spills 191 registers to scratch and is still 21 % faster: 1.02 -> 0.80 ms per 128-stream call; four waves per SIMD, 128 registers: 0.97)
__global__ __launch_bounds__(FP_TPB * FP_TEAM) __attribute__((amdgpu_waves_per_eu(3))) void k_5pt_solve(FPArgs T)
{
    extern __shared__ double s_fp[];
    const int team = threadIdx.x / FP_TEAM, l = threadIdx.x % FP_TEAM, z = blockIdx.y;
    const int it = blockIdx.x * FP_TPB + team;
    const int base = T.cnt ? z * T.stride : T.off[z], n = T.cnt ? T.cnt[z] : T.off[z + 1] - base;
    const double *pd1 = T.pd1 + 2 * (size_t)base, *pd2 = T.pd2 + 2 * (size_t)base;
    const Col L{(fp_lds *)s_fp + team};
    bool ok = it < T.iters;
    int ids[5] = {0, 0, 0, 0, 0};
    if (ok) {
        const int32_t *sm = T.samples + 5 * ((size_t)z * T.iters + it);
        for (int a = 0; a < 5; a++) {
            ids[a] = sm[a];
            if (ids[a] < 0 || ids[a] >= n) ok = false;
            for (int b = 0; b < a; b++) if (ids[a] == ids[b]) ok = false;
        }
    }
    if (__ballot(ok) == 0) {                                      // no valid tuple in this workgroup (one wave): e.g. the whole stream is gated off
        if (l == 0 && it < T.iters) T.ne[(size_t)z * T.iters + it] = 0;
        return;
    }
    double q1[10], q2[10];
    const Col Es = L.at(416);
    for (int a = 0; a < 5; a++) {
        const int id = ok ? ids[a] : 0;
        q1[2 * a] = ok ? pd1[2 * id] : 0.0; q1[2 * a + 1] = ok ? pd1[2 * id + 1] : 0.0;
        q2[2 * a] = ok ? pd2[2 * id] : 0.0; q2[2 * a + 1] = ok ? pd2[2 * id + 1] : 0.0;
    }
    const int ne = five_point_solve(q1, q2, L, l, team, ok);          // all threads: it synchronises
    if (l != 0 || it >= T.iters) return;
    const size_t slot = (size_t)z * T.iters + it;
    int np = 0;
    for (int e = 0; e < ne; e++) {
        double C[48], Ee[9];
        for (int j = 0; j < 9; j++) Ee[j] = Es[9 * e + j];
        if (!essential_poses(Ee, C)) continue;
        int best = -1, bk = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int cnt = 0;
            for (int i = 0; i < 5; i++) {
                double l1, l2;
                ray_depths(C + 12 * k, q1 + 2 * i, q2 + 2 * i, &l1, &l2);
                cnt += (l1 > 0.0 && l2 > 0.0) ? 1 : 0;
            }
            if (cnt > best) { best = cnt; bk = k; }
        }
        double Rb[12];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k == bk) for (int j = 0; j < 12; j++) Rb[j] = C[12 * k + j];
        double *po = T.poses + (slot * FP_MAXE + np) * 12, *eo = T.Es + (slot * FP_MAXE + np) * 9;
        for (int j = 0; j < 12; j++) po[j] = Rb[j];
        for (int j = 0; j < 9; j++) eo[j] = Ee[j];
        np++;
    }
    T.ne[slot] = np;
}

__global__ __launch_bounds__(FP_SCORE_T) void k_5pt_score(FPArgs T)
{
    // one workgroup per 5-tuple scores ALL its poses (a grid of iters x FP_MAXE x S workgroups, most of them for poses that do not
    // exist, each with its own reduction, took 385 us per 32-stream call).  Round 6: a WAVE takes whole poses (pose e on wave e % 4; the 12
    // numbers arrive by scalar loads, the 64 lanes stride over the correspondences, which stay in L1 / L2 between the poses) and counts
    // with ballots, so that the count is wave-uniform and a pose can be DROPPED as soon as it cannot win any more: best[z] (behind the counts:
    // the largest count of a completely scored pose of stream z so far, raised by atomicMax) is an incumbent, and a pose whose count plus the
    // correspondences still to come stays strictly below it is neither the winner nor a tie -- k_5pt_select's winner (most inliers, ties to the
    // earlier tuple, then root), its mask and its summed error are untouched, bit for bit; only the counts of losers are partial.  Nine of a
    // tuple's ten essential matrices are wrong and most tuples are worse than the best one: they now stop after the first 256 correspondences.
    const int it = blockIdx.x, z = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t slot = (size_t)z * T.iters + it;
    const int ne = T.ne[slot];                                  // workgroup-uniform
    int *bestp = T.counts + (size_t)gridDim.y * T.iters * FP_MAXE + z;
    if (tid < FP_MAXE && tid >= ne) T.counts[slot * FP_MAXE + tid] = 0;
    if (ne <= 0) return;
    const int base = T.cnt ? z * T.stride : T.off[z], n = T.cnt ? T.cnt[z] : T.off[z + 1] - base;
    const double *px1 = T.px1 + 2 * (size_t)base, *px2 = T.px2 + 2 * (size_t)base;
    double k1[4], k2[4];
    for (int j = 0; j < 4; j++) { k1[j] = T.ks[8 * z + j]; k2[j] = T.ks[8 * z + 4 + j]; }
    for (int e = wave; e < ne; e += FP_SCORE_T / 64) {
        double Rt[12];
        for (int j = 0; j < 12; j++) Rt[j] = T.poses[(slot * FP_MAXE + e) * 12 + j];
        int best = __hip_atomic_load(bestp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int cnt = 0; bool dropped = false;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            bool ok = false;
            if (i < n) {
                const double a[2] = {px1[2 * i], px1[2 * i + 1]}, b[2] = {px2[2 * i], px2[2 * i + 1]};
                double e1, e2;
                if (two_view_errors(k1, k2, Rt, a, b, &e1, &e2)) ok = e1 < T.thr && e2 < T.thr;
            }
            cnt += __builtin_popcountll(__ballot(ok));
            // can this pose still reach the incumbent?  Tested after every 64 correspondences against the copy in a register (a wrong pose with 5 % inliers
            // falls behind a 70 % incumbent after ~0.32 n of them: at 320, not at 512); the incumbent itself is read again every 256 (an L2 round trip)
            if ((i0 & 192) == 192) best = max(best, __hip_atomic_load(bestp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (cnt + max(n - i0 - 64, 0) < best) { dropped = true; break; }
        }
        if (lane == 0) {
            T.counts[slot * FP_MAXE + e] = cnt;
            if (!dropped) (void)__hip_atomic_fetch_max(bestp, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(FP_SEL_T) void k_5pt_select(FPArgs T)
{
    __shared__ int s_cnt[FP_SEL_T], s_idx[FP_SEL_T];
    __shared__ double s_P[12], s_k[8];
    __shared__ double s_err[FP_ERR_LDS];
    const int tid = threadIdx.x, z = blockIdx.x, ne = FP_MAXE * T.iters;
    const int base = T.cnt ? z * T.stride : T.off[z], n = T.cnt ? T.cnt[z] : T.off[z + 1] - base;
    const double *px1 = T.px1 + 2 * (size_t)base, *px2 = T.px2 + 2 * (size_t)base;
    const int *counts = T.counts + (size_t)z * ne;
    const double *poses = T.poses + (size_t)z * ne * 12, *Es = T.Es + (size_t)z * ne * 9;
    double *errs = T.errs + base, *out = T.out + 32 * (size_t)z;
    uint8_t *inliers = T.inliers + base;
    const bool in_lds = n <= FP_ERR_LDS;
    if (tid < 8) s_k[tid] = T.ks[8 * z + tid];
    int bc = 0, bi = -1;
    for (int e = tid; e < ne; e += FP_SEL_T) {
        const int c = counts[e];
        if (c > bc) { bc = c; bi = e; }
    }
    s_cnt[tid] = bc; s_idx[tid] = bi;
    __syncthreads();
    for (int o = FP_SEL_T / 2; o > 0; o >>= 1) {
        if (tid < o) {
            const int c2 = s_cnt[tid + o], i2 = s_idx[tid + o];
            if (c2 > s_cnt[tid] || (c2 == s_cnt[tid] && c2 > 0 && i2 < s_idx[tid])) { s_cnt[tid] = c2; s_idx[tid] = i2; }
        }
        __syncthreads();
    }
    const int best = s_cnt[0], be = s_idx[0];
    if (tid < 12) s_P[tid] = best > 0 ? poses[(size_t)be * 12 + tid] : 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += FP_SEL_T) {
        double e1 = 0.0, e2 = 0.0;
        bool in = false;
        if (best > 0) {
            const double a[2] = {px1[2 * i], px1[2 * i + 1]}, b[2] = {px2[2 * i], px2[2 * i + 1]};
            in = two_view_errors(s_k, s_k + 4, s_P, a, b, &e1, &e2) && e1 < T.thr && e2 < T.thr;
        }
        inliers[i] = in ? 1 : 0;
        if (in_lds) s_err[i] = in ? e1 + e2 : 0.0; else errs[i] = in ? e1 + e2 : 0.0;   // + 0.0 leaves the sum unchanged
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        double esum = 0.0;
        if (in_lds) {
#pragma unroll 16
            for (int i = 0; i < n; i++) esum += s_err[i];             // index order; the reads pipeline, the adds are the chain
        } else {
#pragma unroll 16
            for (int i = 0; i < n; i++) esum += errs[i];
        }
        out[21] = esum;
        int *oi = (int *)(out + 22);
        oi[0] = best; oi[1] = best > 0 ? be / FP_MAXE : -1;
        for (int j = 0; j < 12; j++) out[j] = s_P[j];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) out[12 + r + 3 * c] = best > 0 ? Es[(size_t)be * 9 + 3 * r + c] : 0.0;
    }
}

// S problems in three launches (grid.y / grid.z / grid.x = problem)
static int fp_run(slam_ctx *ctx, int S, const int32_t *off, const double *px1_xy, const double *px2_xy, const double *pd1_xy,
                  const double *pd2_xy, const double *K1, const double *K2, double max_repr_error, const int32_t *samples, int iters,
                  double *E, double *P, uint8_t *inliers, int *n_inliers, double *error, int *best_iter)
{
    const int ntot = off[S];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t pb = up((size_t)ntot * 16);
    const size_t o_off = 4 * pb, o_k = o_off + up((size_t)(S + 1) * 4), o_smp = o_k + up((size_t)S * 64);
    const size_t o_out = o_smp + up((size_t)S * iters * 20), o_inl = o_out + (size_t)S * 256, total = o_inl + up((size_t)ntot);
    char *h, *d;
    int rc = slam_pinned(ctx, total, (void **)&h);
    if (rc) return rc;
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&d, h, 0));
    memcpy(h, px1_xy, (size_t)ntot * 16); memcpy(h + pb, px2_xy, (size_t)ntot * 16);
    memcpy(h + 2 * pb, pd1_xy, (size_t)ntot * 16); memcpy(h + 3 * pb, pd2_xy, (size_t)ntot * 16);
    memcpy(h + o_off, off, (size_t)(S + 1) * 4);
    double *ks = (double *)(h + o_k);
    for (int z = 0; z < S; z++) {
        const double *a = K1 + 9 * z, *b = K2 + 9 * z;
        ks[8 * z] = a[0]; ks[8 * z + 1] = a[4]; ks[8 * z + 2] = a[6]; ks[8 * z + 3] = a[7];
        ks[8 * z + 4] = b[0]; ks[8 * z + 5] = b[4]; ks[8 * z + 6] = b[6]; ks[8 * z + 7] = b[7];
    }
    memcpy(h + o_smp, samples, (size_t)S * iters * 20);
    const size_t slots = (size_t)S * iters;
    const size_t s_ne = up(slots * 4), s_es = up(slots * FP_MAXE * 72), s_po = up(slots * FP_MAXE * 96);
    const size_t s_cn = up(slots * FP_MAXE * 4 + (size_t)S * 4), s_er = up((size_t)ntot * 8);
    char *scr;
    rc = slam_scratch(ctx, s_ne + s_es + s_po + s_cn + s_er, (void **)&scr);
    if (rc) return rc;
    FPArgs T;
    T.px1 = (const double *)d; T.px2 = (const double *)(d + pb); T.pd1 = (const double *)(d + 2 * pb); T.pd2 = (const double *)(d + 3 * pb);
    T.samples = (const int32_t *)(d + o_smp); T.off = (const int *)(d + o_off); T.ks = (const double *)(d + o_k);
    T.iters = iters; T.thr = max_repr_error; T.cnt = nullptr; T.stride = 0;
    T.ne = (int *)scr; T.Es = (double *)(scr + s_ne); T.poses = (double *)(scr + s_ne + s_es);
    T.counts = (int *)(scr + s_ne + s_es + s_po); T.errs = (double *)(scr + s_ne + s_es + s_po + s_cn);
    T.out = (double *)(d + o_out); T.inliers = (uint8_t *)(d + o_inl);
    const size_t lds = (size_t)FP_LDS_PER_THREAD * FP_TPB * sizeof(double);
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_5pt_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    { ProfScope span(ctx, "five_point_ransac");
      hipLaunchKernelGGL(k_5pt_solve, dim3((iters + FP_TPB - 1) / FP_TPB, S), dim3(FP_TPB * FP_TEAM), lds, ctx->stream, T);
      (void)hipMemsetAsync(T.counts + (size_t)S * iters * FP_MAXE, 0, (size_t)S * 4, ctx->stream);      // the incumbent counts
      hipLaunchKernelGGL(k_5pt_score, dim3(iters, S), dim3(FP_SCORE_T), 0, ctx->stream, T);
      hipLaunchKernelGGL(k_5pt_select, dim3(S), dim3(FP_SEL_T), 0, ctx->stream, T); }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    for (int z = 0; z < S; z++) {
        const char *o = h + o_out + (size_t)z * 256;      // P [0,96) E [96,168) error [168,176) n_inliers [176,180) best_iter [180,184)
        memcpy(P + 12 * z, o, 96);
        if (E) memcpy(E + 9 * z, o + 96, 72);
        if (error) memcpy(error + z, o + 168, 8);
        memcpy(n_inliers + z, o + 176, 4);
        if (best_iter) memcpy(best_iter + z, o + 180, 4);
    }
    memcpy(inliers, h + o_inl, (size_t)ntot);
    return SLAM_OK;
}

extern "C" int slam_five_point_ransac(slam_ctx *ctx, const double *px1_xy, const double *px2_xy, const double *pd1_xy,
                                      const double *pd2_xy, int n, const double *K1, const double *K2, double max_repr_error,
                                      const int32_t *samples, int iters, double *E, double *P, uint8_t *inliers,
                                      int *n_inliers, double *error, int *best_iter)
{
    ARG_TRY(ctx, ctx != nullptr && n >= 0 && iters >= 0);
    ARG_TRY(ctx, K1 && K2 && P && n_inliers);
    ARG_TRY(ctx, n == 0 || (px1_xy && px2_xy && pd1_xy && pd2_xy && inliers));
    ARG_TRY(ctx, iters == 0 || samples);
    if (n < 5 || iters == 0) {
        *n_inliers = 0;
        for (int j = 0; j < 12; j++) P[j] = 0.0;
        if (E) for (int j = 0; j < 9; j++) E[j] = 0.0;
        for (int i = 0; i < n; i++) inliers[i] = 0;
        if (error) *error = 0.0;
        if (best_iter) *best_iter = -1;
        return SLAM_OK;
    }
    const int32_t off[2] = {0, n};
    return fp_run(ctx, 1, off, px1_xy, px2_xy, pd1_xy, pd2_xy, K1, K2, max_repr_error, samples, iters, E, P, inliers, n_inliers, error, best_iter);
}

extern "C" int slam_five_point_ransac_batch(slam_ctx *ctx, int S, const int32_t *offsets, const double *px1_xy, const double *px2_xy,
                                            const double *pd1_xy, const double *pd2_xy, const double *K1, const double *K2,
                                            double max_repr_error, const int32_t *samples, int iters, double *E, double *P,
                                            uint8_t *inliers, int *n_inliers, double *error, int *best_iter)
{
    ARG_TRY(ctx, ctx != nullptr && S >= 0 && iters >= 0);
    if (S == 0) return SLAM_OK;
    ARG_TRY(ctx, offsets && K1 && K2 && P && n_inliers && offsets[0] == 0);
    for (int z = 0; z < S; z++) ARG_TRY(ctx, offsets[z + 1] >= offsets[z]);
    const int ntot = offsets[S];
    ARG_TRY(ctx, ntot == 0 || (px1_xy && px2_xy && pd1_xy && pd2_xy && inliers));
    ARG_TRY(ctx, iters == 0 || samples);
    if (ntot == 0 || iters == 0) {
        for (int z = 0; z < S; z++) {
            n_inliers[z] = 0;
            for (int j = 0; j < 12; j++) P[12 * z + j] = 0.0;
            if (E) for (int j = 0; j < 9; j++) E[9 * z + j] = 0.0;
            if (error) error[z] = 0.0;
            if (best_iter) best_iter[z] = -1;
        }
        for (int i = 0; i < ntot; i++) inliers[i] = 0;
        return SLAM_OK;
    }
    return fp_run(ctx, S, offsets, px1_xy, px2_xy, pd1_xy, pd2_xy, K1, K2, max_repr_error, samples, iters, E, P, inliers, n_inliers, error, best_iter);
}


// =====================================================================================================================
// compute_pose_5pt! on the device-resident keypoint set (src/front_end.jl:242-332; called every frame at :105): the keypoints the
// previous key-frame also observes (slam_kpset_keyframe keeps that observation beside every keypoint) -> undistorted pixels and
// normalised coordinates of both views, the rotation-compensated average parallax (:277-281) -> five-point RANSAC (the kernels
// above, tuples from the counter-based generator of pose.hip) -> its outliers leave the list (:314-318).  The pose composition
// with the motion-model scale (:320-330) needs the key-frame's and the frame's poses: that stays with the caller
// (keypoint_set.pose_5pt_compose), which receives [R | t] of the essential-matrix decomposition.
// =====================================================================================================================
struct KFiveArgs {
    const double *yx, *kyx; const uint8_t *haskf; const int *count; int cap;
    const double *par;                 // S x 32: [0..8] R_compensation (column-major 3 x 3), [16..19] fx fy cx cy, [20..23] k1 k2 p1 p2
    double *px1, *px2, *pd1, *pd2; int *slot; int *n5; double *psum; double *ks;      // gathered pairs, stride cap; S x 8 intrinsics
    int32_t *samples; int iters; unsigned long long seed; double min_parallax;
    const double *fp_out; const uint8_t *inl;                                          // k_5pt_select's outputs
    uint8_t *flags; double *P; int *status, *ninl; double *parallax;                   // results
};

__device__ __forceinline__ unsigned long long fp_splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ void fp_undistort(double y, double x, double fx, double fy, double cx, double cy, double k1, double k2, double p1, double p2,
                                             double &uy, double &ux)
{   // undistort_point, camera.jl:98-125 (the arithmetic of k_kpose_gather)
    const double ny = (y - cy) / fy, nx = (x - cx) / fx;
    const double s0 = ny * ny, s1 = nx * nx, r2 = s0 + s1;
    const double rd = (1.0 + k1 * r2) + k2 * (r2 * r2);
    const double pp = ny * nx;
    const double dtx = 2 * p1 * pp + p2 * (r2 + 2 * s0), dty = p1 * (r2 + 2 * s1) + 2 * p2 * pp;
    uy = (rd * ny + dty) * fy + cy; ux = (rd * nx + dtx) * fx + cx;
}

__global__ __launch_bounds__(256) void k_kfive_gather(KFiveArgs A)
{
    __shared__ int s_w[4], s_base;
    __shared__ double s_par[4];
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = A.count[z];
    const size_t b = (size_t)z * A.cap;
    const double *par = A.par + 32 * (size_t)z;
    const double fx = par[16], fy = par[17], cx = par[18], cy = par[19], k1 = par[20], k2 = par[21], p1 = par[22], p2 = par[23];
    if (tid == 0) s_base = 0;
    if (tid < 8) A.ks[8 * z + tid] = par[16 + (tid & 3)];        // both views through the same camera
    __syncthreads();
    double psum = 0.0;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int j = c0 + tid;
        const bool take = j < n && A.haskf[b + j] != 0;
        const unsigned long long m = __ballot(take);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wv] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; w++) off += s_w[w];
        if (take) {
            const size_t q = b + j, o = b + off + before;
            double uy, ux, vy, vx;
            fp_undistort(A.yx[2 * q], A.yx[2 * q + 1], fx, fy, cx, cy, k1, k2, p1, p2, uy, ux);
            fp_undistort(A.kyx[2 * q], A.kyx[2 * q + 1], fx, fy, cx, cy, k1, k2, p1, p2, vy, vx);
            const double bx = (ux - cx) / fx, by = (uy - cy) / fy, ax = (vx - cx) / fx, ay = (vy - cy) / fy;
            A.px1[2 * o] = vx; A.px1[2 * o + 1] = vy; A.px2[2 * o] = ux; A.px2[2 * o + 1] = uy;      // (x, y), :266-267
            A.pd1[2 * o] = ax; A.pd1[2 * o + 1] = ay; A.pd2[2 * o] = bx; A.pd2[2 * o + 1] = by;      // position[[1, 2]], :268-269
            A.slot[o] = j;
            // rotation-compensated parallax, :277-279: project(camera, R_compensation * position) - previous undistorted pixel
            const double rx = (par[0] * bx + par[3] * by) + par[6] * 1.0, ry = (par[1] * bx + par[4] * by) + par[7] * 1.0,
                         rz = (par[2] * bx + par[5] * by) + par[8] * 1.0;
            const double qy = fy * ry / rz + cy, qx = fx * rx / rz + cx;
            const double dy = qy - vy, dx = qx - vx;
            psum += sqrt(dy * dy + dx * dx);
        }
        __syncthreads();
        if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    // the reference adds the terms in Dict iteration order (unspecified); here: per lane, then a fixed butterfly, then the four waves
    for (int o = 32; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
    if (lane == 0) s_par[wv] = psum;
    __syncthreads();
    if (tid == 0) { A.n5[z] = s_base; A.psum[z] = (s_par[0] + s_par[1]) + (s_par[2] + s_par[3]); }
}

__global__ __launch_bounds__(256) void k_kfive_samples(KFiveArgs A)
{
    const int z = blockIdx.y, it = blockIdx.x * 256 + threadIdx.x;
    if (it >= A.iters) return;
    const int n = A.n5[z];
    int32_t *sm = A.samples + 5 * ((size_t)z * A.iters + it);
    // :243 fewer than 8 keypoints in the frame, :283 fewer than 8 in the key-frame, :290 not enough parallax -> no RANSAC
    const bool run = A.count[z] >= 8 && n >= 8 && !(A.psum[z] / (double)n < A.min_parallax);
    if (!run) { for (int k = 0; k < 5; k++) sm[k] = -1; return; }
    int idx[5]; unsigned att = 0;
    for (int k = 0; k < 5; k++) {
        for (;;) {
            const unsigned long long h = fp_splitmix64(A.seed ^ ((unsigned long long)z << 48) ^ ((unsigned long long)it << 16) ^ (unsigned long long)att);
            att++;
            const int c = (int)(h % (unsigned long long)n);
            bool dup = false;
            for (int m = 0; m < k; m++) dup = dup || idx[m] == c;
            if (!dup) { idx[k] = c; break; }
        }
    }
    for (int k = 0; k < 5; k++) sm[k] = idx[k];
}

__global__ __launch_bounds__(256) void k_kfive_finish(KFiveArgs A)
{
    const int z = blockIdx.x, tid = threadIdx.x, n = A.n5[z];
    const size_t b = (size_t)z * A.cap;
    const double *out = A.fp_out + 32 * (size_t)z;
    const int best = ((const int *)(out + 22))[0];
    const bool ran = A.count[z] >= 8 && n >= 8 && !(A.psum[z] / (double)n < A.min_parallax);
    const bool ok = ran && best >= 5;                                        // :305
    if (ok && best != n)                                                     // :310-318
        for (int i = tid; i < n; i += 256)
            if (!A.inl[b + i]) A.flags[b + A.slot[b + i]] = 1;
    if (tid == 0) {
        for (int j = 0; j < 12; j++) A.P[12 * z + j] = ok ? out[j] : 0.0;
        A.status[z] = ok ? 1 : 0; A.ninl[z] = ran ? best : 0;
        A.parallax[z] = n > 0 ? A.psum[z] / (double)n : 0.0;
    }
}

int kpset_compact(slam_ctx *ctx, slam_kpset *ks, int mode, const uint8_t *flags_dev);

extern "C" int slam_kpset_compute_pose_5pt(slam_ctx *ctx, slam_kpset *ks, const double *params, double min_parallax, double max_repr_error,
                                           int iters, uint64_t seed, double *P, int32_t *status, int32_t *n_inliers, double *parallax,
                                           int32_t *counts)
{
    ARG_TRY(ctx, ctx != nullptr && ks != nullptr && params != nullptr && iters > 0 && ((P != nullptr) == (status != nullptr)));
    const bool fetch = P != nullptr;                             // P == status == NULL: enqueue only (the filter's effect is on the lists)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int S = ks->S, cap = ks->cap;
    const size_t nc = (size_t)S * cap, slots = (size_t)S * iters;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += up(bytes); return at; };
    const size_t o_px1 = take(nc * 16), o_px2 = take(nc * 16), o_pd1 = take(nc * 16), o_pd2 = take(nc * 16), o_slot = take(nc * 4);
    const size_t o_n5 = take((size_t)S * 4), o_ps = take((size_t)S * 8), o_ks = take((size_t)S * 64), o_smp = take(slots * 20);
    const size_t o_ne = take(slots * 4), o_es = take(slots * FP_MAXE * 72), o_po = take(slots * FP_MAXE * 96), o_cn = take(slots * FP_MAXE * 4 + (size_t)S * 4);      // (+ the incumbent count per stream: k_5pt_score)
    const size_t o_er = take(nc * 8), o_out = take((size_t)S * 256), o_inl = take(nc), o_fl = take(nc);
    const size_t o_P = take((size_t)S * 96), o_st = take((size_t)S * 4), o_ni = take((size_t)S * 4), o_pa = take((size_t)S * 8);
    char *scr;
    int rc = slam_scratch2(ctx, o, (void **)&scr);
    if (rc) return rc;
    const double *par_dev;
    rc = kpset_stage_params(ctx, ks, params, (size_t)S * 32, &par_dev);
    if (rc) return rc;
    KFiveArgs A;
    A.yx = ks->yx; A.kyx = ks->kyx; A.haskf = ks->haskf; A.count = ks->count; A.cap = cap; A.par = par_dev;
    A.px1 = (double *)(scr + o_px1); A.px2 = (double *)(scr + o_px2); A.pd1 = (double *)(scr + o_pd1); A.pd2 = (double *)(scr + o_pd2);
    A.slot = (int *)(scr + o_slot); A.n5 = (int *)(scr + o_n5); A.psum = (double *)(scr + o_ps); A.ks = (double *)(scr + o_ks);
    A.samples = (int32_t *)(scr + o_smp); A.iters = iters; A.seed = seed; A.min_parallax = min_parallax;
    A.fp_out = (const double *)(scr + o_out); A.inl = (const uint8_t *)(scr + o_inl);
    A.flags = (uint8_t *)(scr + o_fl); A.P = (double *)(scr + o_P); A.status = (int *)(scr + o_st); A.ninl = (int *)(scr + o_ni);
    A.parallax = (double *)(scr + o_pa);
    FPArgs T;
    T.px1 = A.px1; T.px2 = A.px2; T.pd1 = A.pd1; T.pd2 = A.pd2; T.samples = A.samples; T.off = nullptr; T.cnt = A.n5; T.stride = cap;
    T.ks = A.ks; T.iters = iters; T.thr = max_repr_error;
    T.ne = (int *)(scr + o_ne); T.Es = (double *)(scr + o_es); T.poses = (double *)(scr + o_po); T.counts = (int *)(scr + o_cn);
    T.errs = (double *)(scr + o_er); T.out = (double *)(scr + o_out); T.inliers = (uint8_t *)(scr + o_inl);
    const size_t lds = (size_t)FP_LDS_PER_THREAD * FP_TPB * sizeof(double);
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_5pt_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(ctx, hipMemsetAsync(A.flags, 0, nc, ctx->stream));
    { ProfScope span(ctx, "kpset_compute_pose_5pt");
      hipLaunchKernelGGL(k_kfive_gather, dim3(S), dim3(256), 0, ctx->stream, A);
      hipLaunchKernelGGL(k_kfive_samples, dim3((iters + 255) / 256, S), dim3(256), 0, ctx->stream, A);
      hipLaunchKernelGGL(k_5pt_solve, dim3((iters + FP_TPB - 1) / FP_TPB, S), dim3(FP_TPB * FP_TEAM), lds, ctx->stream, T);
      (void)hipMemsetAsync(T.counts + (size_t)S * iters * FP_MAXE, 0, (size_t)S * 4, ctx->stream);      // the incumbent counts
      hipLaunchKernelGGL(k_5pt_score, dim3(iters, S), dim3(FP_SCORE_T), 0, ctx->stream, T);
      hipLaunchKernelGGL(k_5pt_select, dim3(S), dim3(FP_SEL_T), 0, ctx->stream, T);
      hipLaunchKernelGGL(k_kfive_finish, dim3(S), dim3(256), 0, ctx->stream, A); }
    HIP_TRY(ctx, hipGetLastError());
    rc = kpset_compact(ctx, ks, 1, A.flags);
    if (rc) return rc;
    if (!fetch) return SLAM_OK;
    void *hv;
    rc = slam_pinned(ctx, up((size_t)S * 96) + 4 * up((size_t)S * 8), &hv);
    if (rc) return rc;
    char *h = (char *)hv;
    const size_t h_st = up((size_t)S * 96), h_ni = h_st + up((size_t)S * 8), h_pa = h_ni + up((size_t)S * 8), h_cn = h_pa + up((size_t)S * 8);
    HIP_TRY(ctx, hipMemcpyAsync(h, A.P, (size_t)S * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + h_st, A.status, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + h_ni, A.ninl, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + h_pa, A.parallax, (size_t)S * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + h_cn, ks->count, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    memcpy(P, h, (size_t)S * 96);
    memcpy(status, h + h_st, (size_t)S * 4);
    if (n_inliers) memcpy(n_inliers, h + h_ni, (size_t)S * 4);
    if (parallax) memcpy(parallax, h + h_pa, (size_t)S * 8);
    if (counts) memcpy(counts, h + h_cn, (size_t)S * 4);
    return SLAM_OK;
}
