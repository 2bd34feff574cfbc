// ba_single.hip (+ ba_device.hpp, ba_batch.hip, ba_window.hip, ba_host.hip) -- local bundle adjustment and single-pose refinement on gfx950.
//
// Replaces bundle_adjustment! / _ba_detect_outliers! / pnp_bundle_adjustment of
// the reference (src/bundle_adjustment.jl:1-171) on the flat LocalBACache
// arrays (src/estimator.jl:16-40).  The Levenberg-Marquardt outer loop is
// LeastSquaresOptim's (trust-region radius update, step-quality test,
// diagonal clamping); the step itself is the EXACT solution of the damped
// normal equations, obtained by eliminating the map points (Schur complement)
// and factorising the 6P x 6P reduced camera system, instead of the
// reference's inexact LSMR solve on the full sparse system.
//
// Per LM iteration (all on device, fixed launch sequence, no host sync; every
// kernel early-outs once the device-side state says "converged"):
//   k_schur_groups   one workgroup per group of map points with the same first free observer: residuals, analytic
//                    2x6 / 2x3 Jacobians, V = Jl'Jl + D, V^-1, bl, W = Jp'Jl and the group's window of pose blocks
//                    -(W V^-1) W' (+ Jp'Jp, gradient, diag U), all in LDS
//   k_schur_reduce   S, g, diag(U) = fixed-order sums of the window partials (deterministic, no atomics)
//   k_band_solve     damped banded Cholesky of S in one workgroup, dp (wide systems: the tiled k_chol_* chain)
//   k_update_groups  per group: dl = V^-1 (bl - W' dp), trial parameters, trial and predicted residuals
//   k_control        rho, accept/reject, radius update, convergence (LeastSquaresOptim's rules)
//   (commit)         accept: the committed and the trial parameter buffers swap roles (LMState::cur, flipped by lm_decide); k_commit is
//                    the host-paced protocol's flip
// Fallback for systems the groups do not cover (block half-bandwidth > 20, a point with > 448 observations):
//   k_linearize, k_points, k_obs_factors, k_blocks (pair lists sorted by pose block), k_backsub, k_trial.
// Observations are re-ordered by map point at upload (map points by first free observer) so a point's observations
// and a group's points are contiguous.
// (this file: the single-window kernels, slam_local_ba, the sharded slam_ba_* protocol and pnp_bundle_adjustment)
#include "ba_device.hpp"

__global__ __launch_bounds__(256) void k_linearize(BADev d, int ignore_outliers, int respect_done) { linearize_body(d, ignore_outliers, respect_done); }
__global__ __launch_bounds__(256) void k_points(BADev d, double inv_delta_host, int use_state)
{
    if (use_state && d.st->converged) return;
    const int k = blockIdx.x * 256 + threadIdx.x, M = d.M;
    if (k >= M) return;
    const int j = d.pt_id[k];
    const double inv_delta = use_state ? 1.0 / d.st->delta : inv_delta_host;
    double V[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    const int t0 = d.pt_start[k], t1 = d.pt_start[k + 1];
    for (int i = t0; i < t1; i++) {
        double jl[6], ff[2];
        ld_rec<6>(d.Jl + (size_t)i * 6, jl); ld_rec<2>(d.f + 2 * (size_t)i, ff);
        const double f0 = ff[0], f1 = ff[1];
        V[0] += jl[0] * jl[0] + jl[3] * jl[3]; V[1] += jl[0] * jl[1] + jl[3] * jl[4]; V[2] += jl[0] * jl[2] + jl[3] * jl[5];
        V[3] += jl[1] * jl[1] + jl[4] * jl[4]; V[4] += jl[1] * jl[2] + jl[4] * jl[5]; V[5] += jl[2] * jl[2] + jl[5] * jl[5];
#pragma unroll
        for (int k = 0; k < 3; k++) bl[k] += jl[k] * f0 + jl[3 + k] * f1;
    }
    V[0] += fmin(fmax(V[0], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    V[3] += fmin(fmax(V[3], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    V[5] += fmin(fmax(V[5], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    double Vi[6];
    inv3_sym(V, Vi);
#pragma unroll
    for (int k = 0; k < 6; k++) d.Vinv[(size_t)k * M + j] = Vi[k];
#pragma unroll
    for (int k = 0; k < 3; k++) d.bl[(size_t)k * M + j] = bl[k];
}

__global__ __launch_bounds__(256) void k_obs_factors(BADev d, int use_state)
{
    if (use_state && d.st->converged) return;
    const int i = blockIdx.x * 256 + threadIdx.x, M = d.M;
    if (i >= d.O) return;
    if (d.pconst[d.opose[i]]) return;                 // never referenced by a pair list
    double *Wo = d.Wm + (size_t)i * 18, *To = d.T + (size_t)i * 18;
    if (!d.hasp[i]) {                                  // ignored outlier: its pair-list entries must contribute nothing
#pragma unroll
        for (int k = 0; k < 18; k++) { Wo[k] = 0.0; To[k] = 0.0; }
        return;
    }
    const int j = d.opoint[i];
    double jp[12], jl[6], Vi[6], wv[18], tv[18];
    ld_rec<12>(d.Jp + (size_t)i * 12, jp); ld_rec<6>(d.Jl + (size_t)i * 6, jl);
#pragma unroll
    for (int k = 0; k < 6; k++) Vi[k] = d.Vinv[(size_t)k * M + j];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const double w0 = jp[a] * jl[0] + jp[6 + a] * jl[3];
        const double w1 = jp[a] * jl[1] + jp[6 + a] * jl[4];
        const double w2 = jp[a] * jl[2] + jp[6 + a] * jl[5];
        wv[3 * a] = w0; wv[3 * a + 1] = w1; wv[3 * a + 2] = w2;
        tv[3 * a] = w0 * Vi[0] + w1 * Vi[1] + w2 * Vi[2];
        tv[3 * a + 1] = w0 * Vi[1] + w1 * Vi[3] + w2 * Vi[4];
        tv[3 * a + 2] = w0 * Vi[2] + w1 * Vi[4] + w2 * Vi[5];
    }
    st_rec<18>(Wo, wv); st_rec<18>(To, tv);
}

__global__ __launch_bounds__(256) void k_blocks(BADev d, int use_state)
{
    __shared__ double s_red[4][48];
    if (use_state && d.st->converged) return;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The blocks are sorted by (p, q) and the blocks of
    // one pose row p read the same T records (and neighbouring rows the same W records): give every XCD one contiguous
    // eighth of the block list, so that those re-reads are L2 hits instead of 8 separate fetches of the T / W arrays.
    const int per = (d.nblk + 7) / 8;
    const int b = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (b >= d.nblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, M = d.M, n = d.n;
    const int2 pq = d.blk_pq[b];
    const int e0 = d.blk_start[b], e1 = d.blk_start[b + 1];
    double acc[36], gg[6], ud[6];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) { gg[k] = 0.0; ud[k] = 0.0; }
    for (int e = e0 + tid; e < e1; e += 256) {
        const int2 tt = d.pairs[e];
        double T[18], W2[18];
        ld_rec<18>(d.T + (size_t)tt.x * 18, T); ld_rec<18>(d.Wm + (size_t)tt.y * 18, W2);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++)
                acc[a + 6 * c] -= T[3 * a] * W2[3 * c] + T[3 * a + 1] * W2[3 * c + 1] + T[3 * a + 2] * W2[3 * c + 2];
        if (tt.x == tt.y) {
            const int i = tt.x, j = d.opoint[i];
            double jp[12];
#pragma unroll
            for (int k = 0; k < 12; k++) jp[k] = d.Jp[(size_t)i * 12 + k];
            const double f0 = d.f[2 * (size_t)i], f1 = d.f[2 * (size_t)i + 1];
            const double b0 = d.bl[j], b1 = d.bl[(size_t)M + j], b2 = d.bl[(size_t)2 * M + j];
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int c = 0; c < 6; c++) acc[a + 6 * c] += jp[a] * jp[c] + jp[6 + a] * jp[6 + c];
                gg[a] += (jp[a] * f0 + jp[6 + a] * f1) - (T[3 * a] * b0 + T[3 * a + 1] * b1 + T[3 * a + 2] * b2);
                ud[a] += jp[a] * jp[a] + jp[6 + a] * jp[6 + a];
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] += __shfl_xor(acc[k], m);
        if (pq.x == pq.y) {
#pragma unroll
            for (int k = 0; k < 6; k++) { gg[k] += __shfl_xor(gg[k], m); ud[k] += __shfl_xor(ud[k], m); }
        }
    }
    // fold the 4 waves in a fixed order through LDS (deterministic)
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 36; k++) s_red[wv][k] = acc[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { s_red[wv][36 + k] = gg[k]; s_red[wv][42 + k] = ud[k]; }
    }
    __syncthreads();
    if (tid < 36) {
        const int a = tid % 6, c = tid / 6;
        const double v = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
        d.S[(size_t)(6 * pq.x + a) + (size_t)(6 * pq.y + c) * n] = v;
        if (pq.x != pq.y) d.S[(size_t)(6 * pq.y + c) + (size_t)(6 * pq.x + a) * n] = v;
    }
    if (pq.x == pq.y && tid >= 64 && tid < 70) {
        const int a = tid - 64;
        d.g[6 * pq.x + a] = ((s_red[0][36 + a] + s_red[1][36 + a]) + s_red[2][36 + a]) + s_red[3][36 + a];
        d.udiag[6 * pq.x + a] = ((s_red[0][42 + a] + s_red[1][42 + a]) + s_red[2][42 + a]) + s_red[3][42 + a];
    }
}

__global__ __launch_bounds__(SG_T) void k_schur_groups(BADev d, double inv_delta_host, int ignore_outliers, int use_state) { schur_groups_body<SG_T>(d, inv_delta_host, ignore_outliers, use_state); }
__global__ __launch_bounds__(256) void k_schur_reduce(BADev d, int use_state) { schur_reduce_body(d, use_state); }
__global__ __launch_bounds__(256) void k_chol_prepare(BADev d, const double *Sin, const double *gin, const double *udin,
                                                      double inv_delta_host, int use_state)
{
    if (use_state && d.st->converged) return;
    const int n = d.n, ld = n + 1;
    const double inv_delta = use_state ? 1.0 / d.st->delta : inv_delta_host;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)ld * n) return;
    const int i = (int)(idx % ld), j = (int)(idx / ld);
    double v;
    if (i == n) v = gin[j];
    else {
        v = Sin[(size_t)i + (size_t)j * n];
        if (i == j) v += fmin(fmax(udin[j], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
    }
    d.Swork[idx] = v;
}

__global__ __launch_bounds__(256) void k_chol_step(BADev d, CholArgs C, double *Linv, int k, int nbr, int use_state)
{
    if (use_state && d.st->converged) return;
    __shared__ double Li[CT][CT + 1], Lr[CT][CT + 1], Lc[CT][CT + 1], Arc[CT][CT + 1], Tmp[CT][CT + 1];
    const int n = C.n, ld = C.ld, tid = threadIdx.x;
    int r, c;
    {
        int b = blockIdx.x, cc = k;
        const int nbc = (n + CT - 1) / CT;
        while (true) { const int cnt = nbr - cc; if (b < cnt || cc == nbc - 1) { r = cc + b; c = cc; break; } b -= cnt; cc++; }
    }
    if (r == k && c == k) return;                                   // the panel's diagonal tile is already factored
    const int wk = min(CT, n - CT * k);
    const int hr = min(CT, n + 1 - CT * r), hc = min(CT, n + 1 - CT * c);
    const double *Lik = Linv + (size_t)k * CT * CT;
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        const int gj = CT * k + j;
        Li[i][j] = Lik[i + CT * j];
        const int ri = CT * r + i;
        Lr[i][j] = (i < hr && j < wk) ? C.A[(size_t)ri + (size_t)gj * ld] : 0.0;
        if (c > k) {
            const int ci = CT * c + i;
            Lc[i][j] = (c != r && i < hc && j < wk) ? C.A[(size_t)ci + (size_t)gj * ld] : 0.0;
            const int aj = CT * c + j;
            Arc[i][j] = (i < hr && aj < n && ri >= aj) ? C.A[(size_t)ri + (size_t)aj * ld] : 0.0;
        }
    }
    __syncthreads();
    // panel solves as small GEMMs: X[i][j] = sum_{m<=j} B[i][m] * Linv[j][m]
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        double s0 = 0.0;
        for (int m = 0; m <= j; m++) s0 += Lr[i][m] * Li[j][m];
        Tmp[i][j] = s0;
    }
    __syncthreads();
    for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; Lr[i][j] = Tmp[i][j]; }
    if (c > k && c != r) {
        __syncthreads();
        for (int e = tid; e < CT * CT; e += 256) {
            const int i = e % CT, j = e / CT;
            double s0 = 0.0;
            for (int m = 0; m <= j; m++) s0 += Lc[i][m] * Li[j][m];
            Tmp[i][j] = s0;
        }
        __syncthreads();
        for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; Lc[i][j] = Tmp[i][j]; }
    }
    __syncthreads();
    if (c == k) {                                                   // panel tile: store L_rk
        for (int e = tid; e < CT * CT; e += 256) {
            const int i = e % CT, j = e / CT;
            // NOT in place: other workgroups of this launch still read the un-solved panel from A
            if (i < hr && j < wk) C.Lf[(size_t)(CT * r + i) + (size_t)(CT * k + j) * ld] = Lr[i][j];
        }
        return;
    }
    const double (*Lcc)[CT + 1] = (c == r) ? Lr : Lc;
    const int wc = min(CT, n - CT * c);
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        if (i < hr && j < wc && CT * r + i >= CT * c + j) {
            double s0 = Arc[i][j];
            for (int m = 0; m < wk; m++) s0 -= Lr[i][m] * Lcc[j][m];
            Arc[i][j] = s0;
        }
    }
    __syncthreads();
    if (r == k + 1 && c == k + 1) {                                 // next panel's diagonal tile is final now
        tile_potrf_inv(Arc, Tmp, hr, wc, C.fail);
        __syncthreads();
        tile_mask_lower(Arc, hr, wc);
        __syncthreads();
        double *Lo = Linv + (size_t)(k + 1) * CT * CT;
        for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; Lo[i + CT * j] = Tmp[i][j]; }
    }
    double *dstm = (r == k + 1 && c == k + 1) ? C.Lf : C.A;        // a factored diagonal tile is final
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        if (i < hr && j < wc && CT * r + i >= CT * c + j) dstm[(size_t)(CT * r + i) + (size_t)(CT * c + j) * ld] = Arc[i][j];
    }
}

__global__ __launch_bounds__(256) void k_chol_first(BADev d, CholArgs C, double *Linv, int use_state)
{
    if (use_state && d.st->converged) return;
    __shared__ double t[CT][CT + 1], inv[CT][CT + 1];
    const int n = C.n, ld = C.ld, tid = threadIdx.x;
    const int h = min(CT, n + 1), w = min(CT, n);
    for (int e = tid; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; t[i][j] = (i < h && j < w && i >= j) ? C.A[(size_t)i + (size_t)j * ld] : 0.0; }
    if (tid == 0) *C.fail = 0;
    __syncthreads();
    tile_potrf_inv(t, inv, h, w, C.fail);
    __syncthreads();
    tile_mask_lower(t, h, w);
    __syncthreads();
    for (int e = tid; e < CT * CT; e += 256) {
        const int i = e % CT, j = e / CT;
        if (i < h && j < w && i >= j) C.Lf[(size_t)i + (size_t)j * ld] = t[i][j];
        Linv[i + CT * j] = inv[i][j];
    }
}

__global__ __launch_bounds__(256) void k_chol_backsolve(BADev d, CholArgs C, const double *Linv, int use_state)
{
    if (use_state && d.st->converged) return;
    __shared__ double x[SOLVE_MAX_N];
    __shared__ double xb[CT];
    __shared__ double Lis[CT][CT + 1];
    const int n = C.n, ld = C.ld, tid = threadIdx.x;
    for (int a = tid; a < n; a += 256) x[a] = C.Lf[(size_t)n + (size_t)a * ld];
    const int nbc = (n + CT - 1) / CT;
    // The 10 block steps are a dependent chain through x, but what they read from HBM (the tile inverse and the block
    // row of L) does not depend on x: the data of step kb-1 is requested before step kb is computed, so the chain
    // only pays LDS latency and barriers instead of two global round trips per step.
    constexpr int RPT = 2;                                       // prefetched rows per thread (a < 512); larger systems read the rest directly
    double lf[RPT][CT], lfn[RPT][CT], li[4], lin[4];
    auto fetch = [&](int kb, double (&rf)[RPT][CT], double (&ri)[4]) {
        const int j0 = CT * kb, w = min(CT, n - j0);
        const double *Li = Linv + (size_t)kb * CT * CT;
#pragma unroll
        for (int q = 0; q < 4; q++) ri[q] = Li[tid + 256 * q];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int a = tid + 256 * r;
#pragma unroll
            for (int j = 0; j < CT; j++) rf[r][j] = (a < j0 && j < w) ? C.Lf[(size_t)(j0 + j) + (size_t)a * ld] : 0.0;
        }
    };
    fetch(nbc - 1, lf, li);
    __syncthreads();
    for (int kb = nbc - 1; kb >= 0; kb--) {
        const int j0 = CT * kb, w = min(CT, n - j0);
#pragma unroll
        for (int q = 0; q < 4; q++) { const int e = tid + 256 * q; Lis[e % CT][e / CT] = li[q]; }      // Linv tile, [i][j] = Li[i + CT j]
        if (kb > 0) fetch(kb - 1, lfn, lin);
        __syncthreads();
        if (tid < w) {                                              // x_k = Linv' y_k : x[a] = sum_{i>=a} Linv[i][a] y[i]
            double s0 = 0.0;
            for (int i = tid; i < w; i++) s0 += Lis[i][tid] * x[j0 + i];
            xb[tid] = s0;
        }
        __syncthreads();
        if (tid < w) x[j0 + tid] = xb[tid];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; r++) {                             // y_a -= sum_j L[j0+j][a] x[j0+j]
            const int a = tid + 256 * r;
            if (a < j0) {
                double s0 = 0.0;
#pragma unroll
                for (int j = 0; j < CT; j++) s0 += lf[r][j] * x[j0 + j < n ? j0 + j : n - 1];
                x[a] -= s0;
            }
        }
        for (int a = tid + 256 * RPT; a < j0; a += 256) {            // rows beyond the prefetched ones (n > 512)
            double s0 = 0.0;
            for (int j = 0; j < w; j++) s0 += C.Lf[(size_t)(j0 + j) + (size_t)a * ld] * x[j0 + j];
            x[a] -= s0;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; r++)
#pragma unroll
            for (int j = 0; j < CT; j++) lf[r][j] = lfn[r][j];
#pragma unroll
        for (int q = 0; q < 4; q++) li[q] = lin[q];
    }
    for (int a = tid; a < n; a += 256) d.dp[a] = x[a];
    if (tid == 0 && *C.fail) d.st->chol_fail = 1;
}

__global__ __launch_bounds__(BS_T) void k_band_solve(BADev d, BandArgs B, int use_state) { band_solve_body(d, B, use_state); }
__global__ __launch_bounds__(DS_T) void k_dense_solve(BADev d, BandArgs B, int use_state)
{
    if (use_state && d.st->converged) return;
    extern __shared__ __attribute__((aligned(16))) double ds_sm[];
    __shared__ int s_bad;
    const int F = B.nb, n = d.n, ns = 6 * F, tid = threadIdx.x, nblk = F * (F + 1) / 2;
    double *A = ds_sm;                                   // [nblk][36]
    double *y = A + (size_t)nblk * 36;                   // [ns]: g -> y -> dp
    double *idg = y + ns;                                // [ns]: 1 / L_jj
    const double inv_delta = use_state ? 1.0 / d.st->delta : B.inv_delta_host;
    if (tid == 0) s_bad = 0;
    // ---- load: EVERY element is requested before the first one is stored (S was written by other kernels from all eight XCDs: first
    //      touches are trips to memory, and a load-store loop pays one after the other -- 25 round trips were half of the kernel).  Block
    //      (i, j) entry (r, c) is read through its symmetric twin S[6 j + c, 6 i + r] so that consecutive lanes read consecutive addresses
    constexpr int NL = (DS_MAXF * (DS_MAXF + 1) / 2 * 36 + DS_T - 1) / DS_T;
    {
        double v[NL];
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int e = tid + u * DS_T;
            v[u] = 0.0;
            if (e < nblk * 36) {
                const int blk = e / 36, q = e - 36 * blk, r = q / 6, c = q - 6 * r;
                int i = (int)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
                while (i * (i + 1) / 2 > blk) i--;
                while ((i + 1) * (i + 2) / 2 <= blk) i++;
                const int j = blk - i * (i + 1) / 2;
                v[u] = B.S[(size_t)(6 * j + c) + (size_t)(6 * i + r) * n];
            }
        }
        const double udv = tid < ns ? B.ud[tid] : 0.0, gv = tid < ns ? B.g[tid] : 0.0;
#pragma unroll
        for (int u = 0; u < NL; u++) { const int e = tid + u * DS_T; if (e < nblk * 36) A[e] = v[u]; }
        if (tid < ns) y[tid] = gv;
        ds_barrier();
        if (tid < ns) { const int k = tid / 6, r = tid - 6 * k; A[(size_t)(k * (k + 1) / 2 + k) * 36 + 7 * r] += fmin(fmax(udv, LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta; }
    }
    ds_barrier();
    // Cholesky of the diagonal block (k, k) by wave 0.  upd: the block first loses L_{k,k-1} L_{k,k-1}^T (the previous column's trailing update
    // for this one block -- the look-ahead: the wave factors block k while the other waves update the rest), lane q < 21 forms lower entry q.
    // Every lane then holds the whole triangle (42 v_readlane) and eliminates it division-free on scaled entries, k_band_solve's scheme:
    // m_ik <- m_ik p_j - m_ij m_kj (two dependent operations per pivot; every second pivot a power-of-two rescale), the six 1 / sqrt from
    // v_rsq_f64 + one Newton step, independent of each other: ~1.5 k cycles per block instead of ~5 k for the textbook loop on one wave
    // (a dependent Float64 operation of a lone wave costs 36 cycles).
    auto factor_diag = [&](int k, bool upd) {
#pragma clang fp contract(fast)
        double *D = A + (size_t)(k * (k + 1) / 2 + k) * 36;
        const int q = tid < 21 ? tid : 20;
        int r = 0; while ((r + 1) * (r + 2) / 2 <= q) r++;
        const int c = q - r * (r + 1) / 2;
        double e = D[6 * r + c];
        if (upd) {
            const double *Lp = A + (size_t)(k * (k + 1) / 2 + k - 1) * 36;
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 6; m++) acc += Lp[6 * r + m] * Lp[6 * c + m];
            e -= acc;
        }
        double M[21], ps[6], sj[6], Lr[21];
#pragma unroll
        for (int t = 0; t < 21; t++) M[t] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(e), t), __builtin_amdgcn_readlane(__double2loint(e), t));
        double sc = 1.0; bool bad = false;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double pj = M[j * (j + 1) / 2 + j];
            bad = bad || !(pj > 0.0 && pj < 1e300);
            pj = (pj > 0.0 && pj < 1e300) ? pj : 1.0;
            ps[j] = pj * sc; sj[j] = sc;
            Lr[j * (j + 1) / 2 + j] = pj;
#pragma unroll
            for (int i = j + 1; i < 6; i++) Lr[i * (i + 1) / 2 + j] = M[i * (i + 1) / 2 + j];      // (unscaled column: times rsqrt(p_j s_j) below)
            if ((j & 1) == 0) {
                const int ex = -__builtin_amdgcn_frexp_exp(pj);
                sc *= __builtin_amdgcn_frexp_mant(pj);
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int k2 = j + 1; k2 <= i; k2++)
                        M[i * (i + 1) / 2 + k2] = __builtin_amdgcn_ldexp(M[i * (i + 1) / 2 + k2] * pj - M[i * (i + 1) / 2 + j] * M[k2 * (k2 + 1) / 2 + j], ex);
            } else {
                sc *= pj;
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int k2 = j + 1; k2 <= i; k2++)
                        M[i * (i + 1) / 2 + k2] = M[i * (i + 1) / 2 + k2] * pj - M[i * (i + 1) / 2 + j] * M[k2 * (k2 + 1) / 2 + j];
            }
        }
        double out = 0.0, iq = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double y0 = __builtin_amdgcn_rsq(ps[j]);
            const double r0 = fma(-(ps[j] * y0), y0, 1.0), rd = fma(y0 * 0.5, r0, y0);
            if (tid == j) iq = sj[j] * rd;                        // 1 / L_jj = s_j rsqrt(p_j s_j)
#pragma unroll
            for (int i = j; i < 6; i++) if (q == i * (i + 1) / 2 + j) out = Lr[i * (i + 1) / 2 + j] * rd;
        }
        if (tid < 21) D[6 * r + c] = out;
        if (tid < 6) idg[6 * k + tid] = iq;
        if (bad && tid == 0) s_bad = 1;
    };
    if (tid < 64) factor_diag(0, false);
    ds_barrier();
    for (int k = 0; k < F; k++) {
        const double *D = A + (size_t)(k * (k + 1) / 2 + k) * 36;
        // (1) panel rows: x L_kk^T = a (forward substitution), item = (block row i > k, row r) or the right-hand side's block k
        const int npan = (F - 1 - k) * 6 + 1;
        for (int it = tid; it < npan; it += DS_T) {
            double *row = it < npan - 1 ? A + (size_t)((k + 1 + it / 6) * (k + 2 + it / 6) / 2 + k) * 36 + 6 * (it % 6) : y + 6 * k;
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double a = row[c];
#pragma unroll
                for (int m = 0; m < 6; m++) if (m < c) a -= x[m] * D[6 * c + m];
                x[c] = a * idg[6 * k + c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) row[c] = x[c];
        }
        ds_barrier();
        // (2) wave 0: the next diagonal block, updated and factored (look-ahead); waves 1-7: the trailing update A_ij -= L_ik L_jk^T of every
        //     other pair k < j <= i, item = (pair, row), and the right-hand side y_j -= L_jk y_k, item = block row j
        if (tid < 64) { if (k + 1 < F) factor_diag(k + 1, true); }
        else {
            // (measured: items of one whole block pair -- L_jk in registers for its six rows, 144 instead of 324 LDS accesses per block -- are
            //  slower: 126 k vs 104 k cycles per solve; past the first columns there are fewer pairs than lanes and a thread's six rows are a chain)
            const int m1 = F - 1 - k, npair = m1 * (m1 + 1) / 2, nupd = npair * 6 + m1;
            for (int it = tid - 64 + 6; it < nupd; it += DS_T - 64) {      // (items 0 .. 5 are the rows of pair (k + 1, k + 1): wave 0's)
                if (it < npair * 6) {
                    const int pr = it / 6, r = it - 6 * pr;
                    int a = (int)((sqrtf(8.0f * (float)pr + 1.0f) - 1.0f) * 0.5f);
                    while (a * (a + 1) / 2 > pr) a--;
                    while ((a + 1) * (a + 2) / 2 <= pr) a++;
                    const int b = pr - a * (a + 1) / 2, i = k + 1 + a, j = k + 1 + b;          // b <= a: j <= i
                    const double *Li = A + (size_t)(i * (i + 1) / 2 + k) * 36 + 6 * r, *Lj = A + (size_t)(j * (j + 1) / 2 + k) * 36;
                    double *T = A + (size_t)(i * (i + 1) / 2 + j) * 36 + 6 * r;
                    double li[6];
#pragma unroll
                    for (int m = 0; m < 6; m++) li[m] = Li[m];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double acc = 0.0;
#pragma unroll
                        for (int m = 0; m < 6; m++) acc += li[m] * Lj[6 * c + m];
                        T[c] -= acc;
                    }
                } else {
                    const int j = k + 1 + (it - npair * 6);
                    const double *Lj = A + (size_t)(j * (j + 1) / 2 + k) * 36;
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double acc = 0.0;
#pragma unroll
                        for (int m = 0; m < 6; m++) acc += Lj[6 * c + m] * y[6 * k + m];
                        y[6 * j + c] -= acc;
                    }
                }
            }
        }
        ds_barrier();
    }
    // ---- L^T dp = y, bottom up: dp_k = L_kk^-T (y_k - sum_{i > k} L_ik^T dp_i).  The diagonal blocks are inverted all at once first
    //      (thread = (block, column of the inverse): six forward substitutions each, in parallel), then wave 0 alone walks the block rows --
    //      lane c forms dp_k[c] from the inverse, the lanes share out the 6 k entries of y above it: no workgroup barrier in the chain
    double *Linv = idg + ns;                             // [F][36] L_kk^-1 (lower, row-major)
    for (int it = tid; it < 6 * F; it += DS_T) {
        const int k = it / 6, c = it - 6 * k;
        const double *D = A + (size_t)(k * (k + 1) / 2 + k) * 36;
        double x[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double a = r == c ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < 6; m++) if (m < r) a -= D[6 * r + m] * x[m];
            x[r] = r < c ? 0.0 : a * idg[6 * k + r];
        }
#pragma unroll
        for (int r = 0; r < 6; r++) Linv[36 * k + 6 * r + c] = x[r];
    }
    ds_barrier();
    if (tid < 64) {
        for (int k = F - 1; k >= 0; k--) {
            if (tid < 6) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) acc += Linv[36 * k + 6 * m + tid] * y[6 * k + m];      // (L^-T y)[c] = sum_m Linv[m][c] y[m]
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                y[6 * k + tid] = acc;
            } else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int it = tid; it < 6 * k; it += 64) {        // y_j[c] -= sum_m L_kj[m][c] dp_k[m], j < k
                const int j = it / 6, c = it - 6 * j;
                const double *Lk = A + (size_t)(k * (k + 1) / 2 + j) * 36;
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) acc += Lk[6 * m + c] * y[6 * k + m];
                y[it] -= acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    ds_barrier();
    double *const dpo = d.dp + 6 * B.p0;
    for (int a = tid; a < ns; a += DS_T) dpo[a] = y[a];
    if (tid == 0) { *B.fail = s_bad; if (s_bad) d.st->chol_fail = 1; }
}

__global__ __launch_bounds__(256) void k_backsub(BADev d, int use_state)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    __shared__ double sh[4];
    if (use_state && d.st->converged) return;
    const int kk = blockIdx.x * 256 + threadIdx.x, M = d.M;
    double mx = 0.0;
    if (kk < M) {
        const int j = d.pt_id[kk];
        double bl[3] = {d.bl[j], d.bl[(size_t)M + j], d.bl[(size_t)2 * M + j]};
        for (int i = d.pt_start[kk]; i < d.pt_start[kk + 1]; i++) {
            if (!d.hasp[i]) continue;
            const double *dp = d.dp + 6 * d.opose[i];
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) { a += d.Jp[(size_t)i * 12 + k] * dp[k]; b += d.Jp[(size_t)i * 12 + 6 + k] * dp[k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) bl[k] -= d.Jl[(size_t)i * 6 + k] * a + d.Jl[(size_t)i * 6 + 3 + k] * b;
        }
        double Vi[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Vi[k] = d.Vinv[(size_t)k * M + j];
        const double l0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
        const double l1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
        const double l2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
        d.dl[3 * j] = l0; d.dl[3 * j + 1] = l1; d.dl[3 * j + 2] = l2;
        pb.pts_t[3 * j] = pb.pts[3 * j] - l0; pb.pts_t[3 * j + 1] = pb.pts[3 * j + 1] - l1; pb.pts_t[3 * j + 2] = pb.pts[3 * j + 2] - l2;
        mx = fmax(fabs(l0), fmax(fabs(l1), fabs(l2)));
    }
    if (kk < d.n) { pb.pose_t[kk] = pb.pose[kk] - d.dp[kk]; mx = fmax(mx, fabs(d.dp[kk])); }
    const double t = block_max(mx, sh);
    if (threadIdx.x == 0) d.part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_trial(BADev d, int ignore_outliers, int use_state, int nb_pts)
{
    const ParamBufs pb = param_bufs(d);                      // committed / trial parameters (LMState::cur)
    __shared__ double sh[4];
    if (use_state && d.st->converged) return;
    const int i = blockIdx.x * 256 + threadIdx.x, O = d.O;
    double st = 0.0, sp = 0.0;
    if (i < O) {
        const int p = d.opose[i], j = d.opoint[i];
        double r[2] = {0.0, 0.0};
        if (!(ignore_outliers && d.outl[i])) {
            const double X[3] = {pb.pts_t[3 * j], pb.pts_t[3 * j + 1], pb.pts_t[3 * j + 2]};
            double pose[6];
#pragma unroll
            for (int k = 0; k < 6; k++) pose[k] = pb.pose_t[6 * p + k];
            obs_eval(pose, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
        }
        double a = 0.0, b = 0.0;
        const double *dp = d.dp + 6 * p, *dl = d.dl + 3 * j;
#pragma unroll
        for (int k = 0; k < 6; k++) { a += d.Jp[(size_t)i * 12 + k] * dp[k]; b += d.Jp[(size_t)i * 12 + 6 + k] * dp[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { a += d.Jl[(size_t)i * 6 + k] * dl[k]; b += d.Jl[(size_t)i * 6 + 3 + k] * dl[k]; }
        a -= d.f[2 * (size_t)i]; b -= d.f[2 * (size_t)i + 1];
        st = r[0] * r[0] + r[1] * r[1];
        sp = a * a + b * b;
    }
    const double t1 = block_sum(st, sh);
    const double t2 = block_sum(sp, sh);
    if (threadIdx.x == 0) { d.part[nb_pts + 2 * blockIdx.x] = t1; d.part[nb_pts + 2 * blockIdx.x + 1] = t2; }
}

__global__ __launch_bounds__(SG_T) void k_update_groups(BADev d, int ignore_outliers, int use_state)
{
    __shared__ double s_dp[SOLVE_MAX_N], s_u[SG_OB * 3], s_dl[SG_SB * 6], s_red[8], s_sct[SOLVE_MAX_N];
    update_groups_body<SG_T>(d, ignore_outliers, use_state, s_dp, s_u, s_dl, s_red, s_sct, false);
}

__global__ __launch_bounds__(256) void k_control(BADev d, int mode, int nb_obs, int nb_pts, int lm, double *out4) { control_body(d, mode, nb_obs, nb_pts, lm, out4); }
__global__ void k_control_gathered(BADev d, const double *g, int nranks)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LMState *s = d.st;
    if (s->converged) return;
    double t = 0.0, p = 0.0, mx = 0.0, cf = 0.0;
    for (int r = 0; r < nranks; r++) { t += g[4 * r]; p += g[4 * r + 1]; mx = fmax(mx, g[4 * r + 2]); cf = fmax(cf, g[4 * r + 3]); }
    s->trial_ssr = t; s->pred_ssr = p; s->maxdx = mx;
    if (cf != 0.0) s->chol_fail = 1;
    lm_decide(s, t, p, mx);
}

__global__ void k_lm_start(BADev d, const double *ssr_slot, int first_pass)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LMState *s = d.st;
    s->ssr = *ssr_slot;
    if (first_pass) { s->ssr_init = s->ssr; s->chol_fail = 0; }
    s->delta = LM_DELTA0; s->decrease_factor = 2.0; s->converged = 0; s->accept = 0; s->iters = 0;
}

__global__ void k_commit(BADev d, int accept_host)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && accept_host) d.st->cur ^= 1;
}

__global__ void k_lm_reset(BADev d, int pass)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LMState *s = d.st;
    if (pass == 0) { s->ssr_init = s->ssr; s->chol_fail = 0; s->n_outliers = 0; }
    if (pass == 3) { s->ssr_pass1 = s->ssr; s->iters_pass1 = s->iters; return; }   // record the end of pass 1
    if (pass == 2) { s->ssr_final = s->ssr; s->iters_pass2 = s->iters; return; }   // record the end of pass 2
    s->delta = LM_DELTA0; s->decrease_factor = 2.0; s->converged = 0; s->accept = 0; s->iters = 0;
}

__global__ __launch_bounds__(256) void k_outliers(BADev d, double repr_eps, double depth_eps) { outliers_body(d, repr_eps, depth_eps); }
__global__ __launch_bounds__(256) void k_outlier_count(BADev d, int nb_obs) { outlier_count_body(d, nb_obs); }

static int ba_setup(slam_ctx *ctx, double fx, double fy, double cx, double cy, int P, int M, int O,
                    const double *theta, const uint8_t *theta_const_in, const double *pixels_yx,
                    const int64_t *pose_ids, const int64_t *point_ids, slam_ba **out, bool ctx_mem = false, bool may_reorder = false)
{
    BAPlan pl;
    pl.fx = fx; pl.fy = fy; pl.cx = cx; pl.cy = cy; pl.P = P; pl.M = M; pl.O = O; pl.theta = theta; pl.theta_const_in = theta_const_in;
    pl.pixels_yx = pixels_yx; pl.pose_ids = pose_ids; pl.point_ids = point_ids; pl.may_reorder = may_reorder;
    pl.nthreads = ba_pool_threads();                          // a large window splits its passes over the observations (BAPlan::chunks)
    if (ba_plan(pl)) return slam_fail(ctx, pl.err, "%s", pl.msg);
    slam_ba *ba = pl.ba;
    ba->device = ctx->device;
    const size_t up_end = pl.up_bytes, zero_end = up_end + pl.zero_bytes, total = zero_end + pl.work_bytes;
    char *A = nullptr;
    if (ctx_mem) {                                             // slam_local_ba: the context's grow-only scratch, no hipMalloc / hipFree per call
        const int rcs = slam_scratch(ctx, total, (void **)&A);
        if (rcs) return rcs;
        ba->owns_arena = false;
    } else {
        hipError_t e = hipMalloc((void **)&A, total);
        if (e != hipSuccess) return slam_fail(ctx, SLAM_ERR_HIP, "slam_ba: hipMalloc(%zu): %s", total, hipGetErrorString(e));
    }
    ba->arena = A;
    struct Guard { slam_ba *b; ~Guard() { if (b && b->arena && b->owns_arena) (void)hipFree(b->arena); } } guard{ba};   // a failing upload frees the arena (the plan owns the object)
    hipStream_t st = ctx->stream;
    char *stage = nullptr;
    std::vector<char> pageable;
    if (ctx_mem) { const int rcs = slam_pinned(ctx, up_end, (void **)&stage); if (rcs) return rcs; }
    else { pageable.resize(up_end); stage = pageable.data(); }
    if (ba_emit(pl, A, A + up_end, A + zero_end, stage)) return slam_fail(ctx, pl.err, "%s", pl.msg);
    HIP_TRY(ctx, hipMemcpyAsync(A, stage, up_end, hipMemcpyHostToDevice, st));                  // pinned -> device: one DMA, nothing to wait for
    HIP_TRY(ctx, hipMemsetAsync(A + up_end, 0, zero_end - up_end, st));                         // LM state, flags, outlier marks, dp, the reduce buffer
    if (!ctx_mem) HIP_TRY(ctx, slam_stream_wait(st));        // the pageable staging block goes out of scope
    guard.b = nullptr;
    pl.ba = nullptr;                                           // ownership passes to the caller
    *out = ba;
    return SLAM_OK;
}


// linearise at the current parameters and build [S; g; udiag] into `red`
static int ba_enqueue_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double inv_delta, int use_state, double *red)
{
    BADev d = ba->d;
    const int n = d.n;
    d.S = red; d.g = red + (size_t)n * n; d.udiag = d.g + n;
    hipStream_t st = ctx->stream;
    // the grouped build rewrites every block inside THIS problem's band (d.whb), g and diag(U) each time: the rest of the buffer only
    // needs zeroing once -- as long as nobody else writes to it.  A caller-owned buffer (the sharded path all-reduces it in place, and a
    // peer's band may be wider than ours: those blocks would keep the previous iteration's SUM and be summed again) is zeroed every time.
    const bool private_red = red == ba->reduce;
    if (!ba->grouped || !private_red || ba->zeroed != red) { HIP_TRY(ctx, hipMemsetAsync(red, 0, ((size_t)n * n + 2 * n + 8) * 8, st)); ba->zeroed = private_red ? red : nullptr; }
    if (ba->grouped) {
        // the attribute belongs to the function object of the CURRENT device: once per device, result checked
        // (the reference's three tasks call the library concurrently, SLAM.jl:166: the flag is atomic; setting the attribute twice is harmless)
        static std::atomic<bool> attr_set[64];
        const int dv = ctx->device & 63;
        if (!attr_set[dv].load(std::memory_order_acquire)) { HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_schur_groups, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(sg_lds_bytes(BS_MAXHB, SOLVE_MAX_N / 6), (size_t)150 * 1024))); attr_set[dv].store(true, std::memory_order_release); }
        hipLaunchKernelGGL(k_schur_groups, dim3(d.ngrp), dim3(SG_T), sg_lds_bytes(d.whb, d.P, d.sg_ob, d.sg_sb, 512, d.sg_hp), st, d, inv_delta, ignore_outliers, use_state);
        if (!use_state) hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 0, d.ngrp, ba->nblocks_pts, 0, red + (size_t)n * n + 2 * n);
        const int nthr = d.P * (d.whb + 1) * 36 + d.P * 12;
        hipLaunchKernelGGL(k_schur_reduce, dim3((nthr + 255) / 256), dim3(256), 0, st, d, use_state);
        return SLAM_OK;
    }
    hipLaunchKernelGGL(k_linearize, dim3(ba->nblocks_obs), dim3(256), 0, st, d, ignore_outliers, use_state);
    if (!use_state) hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 0, ba->nblocks_obs, ba->nblocks_pts, 0, red + (size_t)n * n + 2 * n);
    if (d.M > 0) hipLaunchKernelGGL(k_points, dim3((d.M + 255) / 256), dim3(256), 0, st, d, inv_delta, use_state);
    if (d.O > 0) hipLaunchKernelGGL(k_obs_factors, dim3((d.O + 255) / 256), dim3(256), 0, st, d, use_state);
    if (d.nblk > 0) hipLaunchKernelGGL(k_blocks, dim3(((d.nblk + 7) / 8) * 8), dim3(256), 0, st, d, use_state);
    return SLAM_OK;
}

static int ba_enqueue_solve(slam_ctx *ctx, slam_ba *ba, const double *red, int ignore_outliers, double inv_delta, int use_state,
                            int lm, double *out4)
{
    BADev d = ba->d;
    const int n = d.n;
    hipStream_t st = ctx->stream;
    static const bool no_band = getenv("SLAMHIP_NO_BAND") != nullptr;
    const int Ps = ba->pspan > 0 ? ba->pspan : d.P, p0 = ba->pspan > 0 ? ba->p0 : 0;       // the poses the banded solve covers: first .. last free pose
    const int hb = std::min(std::max(ba->hb, 1), Ps - 1);      // >= 1: the factor wave reads block row k + 1 while row k + 1 + hb enters the ring
    const size_t band_lds = band_lds_bytes(n, Ps, hb);
    if (ba->grouped && hb > BS_MAXHB && Ps <= DS_MAXF) {       // not banded, small: dense one-workgroup solve (ba_plan admitted the groups for exactly this case)
        BandArgs B = {}; B.S = red + (size_t)6 * p0 * (n + 1); B.g = red + (size_t)n * n + 6 * p0; B.ud = red + (size_t)n * n + n + 6 * p0; B.Lg = nullptr; B.nb = Ps; B.hb = hb; B.p0 = p0;
        B.inv_delta_host = inv_delta; B.fail = ba->chol_flag;
        static std::atomic<bool> ds_attr[64];
        const int dv = ctx->device & 63;
        if (!ds_attr[dv].load(std::memory_order_acquire)) { HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dense_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dense_lds_bytes(DS_MAXF))); ds_attr[dv].store(true, std::memory_order_release); }
        hipLaunchKernelGGL(k_dense_solve, dim3(1), dim3(DS_T), dense_lds_bytes(Ps), st, d, B, use_state);
    } else
    if (!no_band && hb <= BS_MAXHB && band_lds <= 150 * 1024) {
        BandArgs B; B.S = red + (size_t)6 * p0 * (n + 1); B.g = red + (size_t)n * n + 6 * p0; B.ud = red + (size_t)n * n + n + 6 * p0; B.Lg = ba->band; B.nb = Ps; B.hb = hb; B.p0 = p0;
        B.inv_delta_host = inv_delta; B.fail = ba->chol_flag; B.lds_bytes = (int)band_lds;
        B.xchg = ba->xchg; B.epoch = ++ba->epoch;
        static const int twist_shift = [] { const char *v = getenv("SLAMHIP_TWIST_SHIFT"); return v ? atoi(v) : 0; }();    // (measurement knob: side 0 takes 2 x shift columns more than side 1; +1 paid while the hand-over cost 13 k cycles, with 7 k an even split is 1 % ahead)
        B.shift = twist_shift;
        static const bool no_twist = getenv("SLAMHIP_NO_TWIST") != nullptr;
        // (the two workgroups wait for each other: both must be resident, which a stream confined to one compute unit cannot promise)
        static const int twist_min = [] { const char *v = getenv("SLAMHIP_TWIST_MIN"); return v ? atoi(v) : 0; }();    // (measurement knob)
        const bool twist = !no_twist && hb * 6 <= 58 && Ps >= (twist_min > 0 ? std::max(twist_min, hb + 8) : std::max(2 * (hb + 1) - 1, hb + 8))      /* measured: pays from 19 free poses at hb = 9 (19: 92.1 -> 89.3 us per iteration, 18: equal) since the hand-overs stay in one L2 (24 before) */ && ctx->xwg_ok;
        static long long *trace_dev = nullptr; static int trace_n = 0;
        static const bool trace_on = getenv("SLAMHIP_BAND_TRACE") != nullptr;
        if (trace_on && !trace_dev) (void)hipHostMalloc((void **)&trace_dev, 1024);
        B.trace = trace_dev;
        if (trace_on && trace_n++ == 8) {      // side 0 of the ninth launch, shader cycles
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "band trace (cycles): factor wave: panel %lld factor %lld barrier %lld, then middle + back-substitution %lld | update wave 1: flag %lld update %lld barrier %lld | prefetch wave: flag %lld put/fetch %lld | backsub: chat %lld G %lld recurrence %lld\n",
                    trace_dev[5], trace_dev[4], trace_dev[1], trace_dev[3], trace_dev[2], trace_dev[6], trace_dev[7], trace_dev[8], trace_dev[9], trace_dev[10], trace_dev[11], trace_dev[12]);
            fprintf(stderr, "  side 0 timeline (cycles): set-up (damping, first window, tables, L2 warm-up) %lld; then, since the end of the set-up: column loop starts %lld, own columns done %lld, middle assembled + factored %lld, forward done %lld\n", trace_dev[26], trace_dev[16], trace_dev[17], trace_dev[18], trace_dev[19]);
            fprintf(stderr, "  set-up: pair table %lld, window requested %lld, tables %lld, warm-up requested %lld, damping in LDS %lld, window in the ring %lld\n", trace_dev[103], trace_dev[104], trace_dev[105], trace_dev[106], trace_dev[107], trace_dev[26]);
            fprintf(stderr, "  middle: entries prepared %lld, flag seen %lld, fence done %lld, assembled %lld\n", trace_dev[22], trace_dev[23], trace_dev[24], trace_dev[25]);
            fprintf(stderr, "  XCC_ID of side 0 / side 1: %lld / %lld; of the idle workgroups 1-7:", trace_dev[20] & 15, trace_dev[21] & 15);
            for (int w = 1; w < 8; w++) fprintf(stderr, " %lld", trace_dev[80 + w] & 15);
            fprintf(stderr, "\n");
            const long long t0 = trace_dev[32];      // step 10 per wave, relative to wave 0's start of the step: start, (wave 0: flag published), arrival at the barrier, release
            fprintf(stderr, "  factor wave, step 10: row loaded %lld, substituted %lld, flag %lld, D entry %lld, D in every lane %lld\n",
                    trace_dev[96] - t0, trace_dev[97] - t0, trace_dev[40] - t0, trace_dev[98] - t0, trace_dev[99] - t0);
            fprintf(stderr, "  wave 3: registers copied %lld, blocks in the ring %lld, next row requested %lld\n", trace_dev[72] - t0, trace_dev[73] - t0, trace_dev[74] - t0);
            for (int w = 0; w < 8; w++) fprintf(stderr, "  wave %d (simd %lld): start %lld%s arrive %lld release %lld\n", w, (trace_dev[64 + w] >> 4) & 3, trace_dev[32 + w] - t0,
                                                w == 0 ? (" flag " + std::to_string(trace_dev[40] - t0)).c_str() : "", trace_dev[48 + w] - t0, trace_dev[56 + w] - t0);
        }
        static std::atomic<bool> attr_set[64];
        const int dv = ctx->device & 63;
        if (!attr_set[dv].load(std::memory_order_acquire)) { HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_band_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr_set[dv].store(true, std::memory_order_release); }
        static const bool twist_spread = getenv("SLAMHIP_TWIST_SPREAD") != nullptr;     // (test knob: the two sides on different XCDs)
        hipLaunchKernelGGL(k_band_solve, dim3(twist ? (twist_spread ? 2 : 9) : 1), dim3(BS_T), band_lds, st, d, B, use_state);
    } else {
        CholArgs C; C.A = d.Swork; C.Lf = ba->lfac; C.n = n; C.ld = n + 1; C.fail = ba->chol_flag;
        const size_t tot = (size_t)(n + 1) * n;
        hipLaunchKernelGGL(k_chol_prepare, dim3((tot + 255) / 256), dim3(256), 0, st, d, red, red + (size_t)n * n, red + (size_t)n * n + n, inv_delta, use_state);
        hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(256), 0, st, d, C, ba->linv, use_state);
        const int nbc = (n + CT - 1) / CT, nbr = (n + 1 + CT - 1) / CT;
        for (int k = 0; k < nbc; k++) {
            int tiles = 0;
            for (int c = k; c < nbc; c++) tiles += nbr - c;
            if (tiles > 1) hipLaunchKernelGGL(k_chol_step, dim3(tiles), dim3(256), 0, st, d, C, ba->linv, k, nbr, use_state);
        }
        hipLaunchKernelGGL(k_chol_backsolve, dim3(1), dim3(256), 0, st, d, C, (const double *)ba->linv, use_state);
    }
    if (ba->grouped) {
        hipLaunchKernelGGL(k_update_groups, dim3(d.ngrp), dim3(SG_T), 0, st, d, ignore_outliers, use_state);
        hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 1, d.ngrp, d.ngrp, lm | (use_state ? 2 : 0), out4);
        return SLAM_OK;
    }
    hipLaunchKernelGGL(k_backsub, dim3(ba->nblocks_pts), dim3(256), 0, st, d, use_state);
    hipLaunchKernelGGL(k_trial, dim3(ba->nblocks_obs), dim3(256), 0, st, d, ignore_outliers, use_state, ba->nblocks_pts);
    hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 1, ba->nblocks_obs, ba->nblocks_pts, lm | (use_state ? 2 : 0), out4);
    return SLAM_OK;
}

static int ba_enqueue_commit(slam_ctx *ctx, slam_ba *ba, int accept, int use_state, int iter_tag)
{
    (void)iter_tag;
    if (use_state) return SLAM_OK;                           // device-paced: lm_decide has swapped the buffers already
    hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, ctx->stream, ba->d, accept);
    return SLAM_OK;
}

extern "C" {

int slam_ba_destroy(slam_ba *ba)
{
    if (!ba) return SLAM_OK;
    if (ba->owns_arena) {
        (void)hipSetDevice(ba->device);
        (void)hipDeviceSynchronize();
        if (ba->arena) (void)hipFree(ba->arena);
    }
    delete ba;
    return SLAM_OK;
}

int64_t slam_ba_reduce_len(int P) { const int64_t n = 6 * (int64_t)P; return n * n + 2 * n + 8; }

int slam_ba_create(slam_ctx *ctx, double fx, double fy, double cx, double cy, int P, int M_local, int O_local,
                   const double *theta, const uint8_t *theta_const, const double *pixels_yx,
                   const int64_t *pose_ids, const int64_t *point_ids_local, slam_ba **out)
{
    ARG_TRY(ctx, ctx != nullptr && out != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ba_setup(ctx, fx, fy, cx, cy, P, M_local, O_local, theta, theta_const, pixels_yx, pose_ids, point_ids_local, out);
}

int slam_ba_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double inv_delta, double *reduce_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_build(ctx, ba, ignore_outliers, inv_delta, 0, reduce_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_ba_solve(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, double inv_delta, double *trial_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr && trial_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ignore_outliers for the trial residual follows the flags: outliers are only ever set by slam_ba_flag_outliers
    int rc = ba_enqueue_solve(ctx, ba, reduce_dev, 1, inv_delta, 0, 0, trial_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_ba_commit(slam_ctx *ctx, slam_ba *ba, int accept)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ba_enqueue_commit(ctx, ba, accept, 0, 0);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

// ---- device-paced LM for the sharded path: every call returns after enqueueing on ctx's stream; the accept / reject decision
// is taken on the device from the gathered trial costs, so a pass needs no host synchronisation (slam.h has the protocol) ----
int slam_ba_lm_begin(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double *reduce_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_build(ctx, ba, ignore_outliers, 1.0 / LM_DELTA0, 0, reduce_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_start(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, int first_pass)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n = ba->d.n;
    hipLaunchKernelGGL(k_lm_start, dim3(1), dim3(1), 0, ctx->stream, ba->d, reduce_dev + (size_t)n * n + 2 * n, first_pass);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_build(slam_ctx *ctx, slam_ba *ba, int ignore_outliers, double *reduce_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_build(ctx, ba, ignore_outliers, 0.0, 1, reduce_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_solve(slam_ctx *ctx, slam_ba *ba, const double *reduce_dev, int ignore_outliers, double *trial_dev)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && reduce_dev != nullptr && trial_dev != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ba_enqueue_solve(ctx, ba, reduce_dev, ignore_outliers, 0.0, 1, 0, trial_dev);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
int slam_ba_lm_step(slam_ctx *ctx, slam_ba *ba, const double *gathered_dev, int nranks, int iter_tag)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && gathered_dev != nullptr && nranks >= 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_control_gathered, dim3(1), dim3(1), 0, ctx->stream, ba->d, gathered_dev, nranks);
    ba_enqueue_commit(ctx, ba, 0, 1, iter_tag);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}
// synchronises; out8 = {ssr, iters, converged, delta, chol_fail, ssr_init, trial_ssr, max|dx|}
int slam_ba_lm_state(slam_ctx *ctx, slam_ba *ba, double *out8)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr && out8 != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LMState h;
    HIP_TRY(ctx, hipMemcpyAsync(&h, ba->d.st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    out8[0] = h.ssr; out8[1] = h.iters; out8[2] = h.converged; out8[3] = h.delta; out8[4] = h.chol_fail; out8[5] = h.ssr_init;
    out8[6] = h.trial_ssr; out8[7] = h.maxdx;
    return SLAM_OK;
}
// block half-bandwidth of this shard's reduced system (S_pq = 0 for |p - q| > hb); the all-reduced system has the maximum over
// the ranks, which the driver sets on every rank before the first solve
// host only: the pose order slam_local_ba solves in, and the block half-bandwidth of the reduced camera system in that order
int slam_ba_plan_order(int P, int M, int O, const uint8_t *theta_const, const int64_t *pose_ids, const int64_t *point_ids, int32_t *order_out, int *hb_out)
{
    if (P <= 0 || M < 0 || O < 0 || !theta_const || (O > 0 && (!pose_ids || !point_ids))) return SLAM_ERR_ARG;
    for (int i = 0; i < O; i++) if (pose_ids[i] < 1 || pose_ids[i] > P || point_ids[i] < 1 || point_ids[i] > M) return SLAM_ERR_ARG;
    std::vector<int> new_of(P), order(P);
    for (int p = 0; p < P; p++) new_of[p] = order[p] = p;
    auto halfband = [&]() {
        std::vector<int> lo(M, P), hi(M, -1);
        for (int i = 0; i < O; i++) {
            if (theta_const[pose_ids[i] - 1]) continue;
            const int j = (int)point_ids[i] - 1, q = new_of[pose_ids[i] - 1];
            lo[j] = std::min(lo[j], q); hi[j] = std::max(hi[j], q);
        }
        int hb = 0;
        for (int j = 0; j < M; j++) if (hi[j] >= 0) hb = std::max(hb, hi[j] - lo[j]);
        return hb;
    };
    int hb = halfband(), reordered = 0;
    static const bool no_reorder = getenv("SLAMHIP_BA_NO_REORDER") != nullptr;
    std::vector<int> cand;
    if (!no_reorder && M > 0 && O > 0 && (hb > BS_MAXHB || !sg_fold_fits(hb)) && ba_pose_order(P, M, O, theta_const, pose_ids, point_ids, cand)) {
        order = cand;
        for (int k = 0; k < P; k++) new_of[order[k]] = k;
        hb = halfband(); reordered = 1;
    }
    if (order_out) for (int k = 0; k < P; k++) order_out[k] = order[k];
    if (hb_out) *hb_out = hb;
    return reordered;
}

int slam_ba_halfband(const slam_ba *ba) { return ba ? ba->hb : SLAM_ERR_ARG; }
int slam_ba_set_halfband(slam_ba *ba, int hb) { if (!ba || hb < 0) return SLAM_ERR_ARG; ba->hb = hb; return SLAM_OK; }

int slam_ba_flag_outliers(slam_ctx *ctx, slam_ba *ba, double repr_eps, double depth_eps, int *n_out)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_outliers, dim3(ba->nblocks_obs), dim3(256), 0, ctx->stream, ba->d, repr_eps, depth_eps);
    hipLaunchKernelGGL(k_outlier_count, dim3(1), dim3(256), 0, ctx->stream, ba->d, ba->nblocks_obs);
    HIP_TRY(ctx, hipGetLastError());
    LMState h;
    HIP_TRY(ctx, hipMemcpyAsync(&h, ba->d.st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    if (n_out) *n_out = h.n_outliers;
    return SLAM_OK;
}

// cur_known: LMState::cur if the caller has the state on the host already, -1: ask the device (one more synchronisation)
static int ba_download(slam_ctx *ctx, slam_ba *ba, double *theta, uint8_t *outliers, int cur_known)
{
    ARG_TRY(ctx, ctx != nullptr && ba != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const BADev &d = ba->d;
    if (theta) {
        int cur = cur_known;                                 // which buffer pair holds the committed parameters (LMState::cur)
        if (cur < 0) {
            HIP_TRY(ctx, hipMemcpyAsync(&cur, &d.st->cur, sizeof cur, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, slam_stream_wait(ctx->stream));
        }
        if (ba->pose_order.empty()) HIP_TRY(ctx, hipMemcpyAsync(theta, cur ? d.pose_t : d.pose, (size_t)d.n * 8, hipMemcpyDeviceToHost, ctx->stream));
        else {                                               // the solver's pose order -> the caller's
            std::vector<double> tmp(d.n);
            HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), cur ? d.pose_t : d.pose, (size_t)d.n * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, slam_stream_wait(ctx->stream));
            for (int k = 0; k < d.P; k++) memcpy(theta + 6 * ba->pose_order[k], &tmp[6 * k], 48);
        }
        if (d.M > 0) HIP_TRY(ctx, hipMemcpyAsync(theta + d.n, cur ? d.pts_t : d.pts, (size_t)3 * d.M * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    std::vector<uint8_t> tmp;
    if (outliers && d.O > 0) { tmp.resize(d.O); HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), d.outl, (size_t)d.O, hipMemcpyDeviceToHost, ctx->stream)); }
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    if (outliers) for (int s = 0; s < d.O; s++) outliers[ba->perm[s]] = tmp[s];
    return SLAM_OK;
}
int slam_ba_download(slam_ctx *ctx, slam_ba *ba, double *theta, uint8_t *outliers) { return ba_download(ctx, ba, theta, outliers, -1); }

int slam_local_ba(slam_ctx *ctx, double fx, double fy, double cx, double cy, int P, int M, int O,
                  double *theta, const uint8_t *theta_const, const double *pixels_yx,
                  const int64_t *pose_ids, const int64_t *point_ids, uint8_t *outliers,
                  int iters_fast, int iterations, double repr_eps, double *stats)
{
    ARG_TRY(ctx, ctx != nullptr && outliers != nullptr && iters_fast >= 0 && iterations >= 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    slam_ba *ba = nullptr;
    static const bool host_times = getenv("SLAMHIP_BA_HOSTTIME") != nullptr;
    const auto tw0 = std::chrono::steady_clock::now();
    int rc = ba_setup(ctx, fx, fy, cx, cy, P, M, O, theta, theta_const, pixels_yx, pose_ids, point_ids, &ba, true, true);
    if (rc) return rc;
    const auto tw1 = std::chrono::steady_clock::now();
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    BADev d = ba->d;
    auto run_pass = [&](int ignore, int iters) {
        // f / ssr at the start of the pass (LeastSquaresOptim evaluates f!(fcur, x) first)
        hipLaunchKernelGGL(k_linearize, dim3(ba->nblocks_obs), dim3(256), 0, st, d, ignore, 0);
        hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, st, d, 0, ba->nblocks_obs, ba->nblocks_pts, 0, (double *)nullptr);
        hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(1), 0, st, d, ignore ? 1 : 0);
        for (int it = 1; it <= iters; it++) {
            ba_enqueue_build(ctx, ba, ignore, 0.0, 1, ba->reduce);
            ba_enqueue_solve(ctx, ba, ba->reduce, ignore, 0.0, 1, 1, nullptr);
            ba_enqueue_commit(ctx, ba, 0, 1, it);
        }
    };
    run_pass(0, iters_fast);
    hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(1), 0, st, d, 3);
    // flag outliers at theta_1 (bundle_adjustment.jl:45)
    hipLaunchKernelGGL(k_outliers, dim3(ba->nblocks_obs), dim3(256), 0, st, d, repr_eps, 1e-6);
    hipLaunchKernelGGL(k_outlier_count, dim3(1), dim3(256), 0, st, d, ba->nblocks_obs);
    run_pass(1, iterations);
    hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(1), 0, st, d, 2);
    (void)hipEventRecord(e1, st);
    LMState h;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d.st, sizeof h, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = slam_stream_wait(st);
    float ms = 0;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) { slam_ba_destroy(ba); return slam_fail(ctx, SLAM_ERR_HIP, "slam_local_ba: %s", hipGetErrorString(e)); }
    // A failed factorisation leaves the caller's theta and outliers untouched (the reference's LSMR step cannot fail and
    // never leaves cache.theta half-updated): the state is only copied back from a run that completed.
    const auto tw2 = std::chrono::steady_clock::now();
    rc = h.chol_fail ? SLAM_OK : ba_download(ctx, ba, theta, outliers, h.cur);
    const auto tw3 = std::chrono::steady_clock::now();
    slam_ba_destroy(ba);
    if (host_times) {
        const auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "slam_local_ba host: setup %ld us, enqueue + wait %ld us (device %.0f us), download %ld us, destroy %ld us\n",
                us(tw0, tw1), us(tw1, tw2), ms * 1e3, us(tw2, tw3), us(tw3, std::chrono::steady_clock::now()));
    }
    if (rc) return rc;
    if (stats) {
        stats[0] = h.ssr_init; stats[1] = h.ssr_pass1; stats[2] = h.ssr_final; stats[3] = h.iters_pass1; stats[4] = h.iters_pass2;
        stats[5] = h.n_outliers; stats[6] = ms; stats[7] = h.chol_fail;
    }
    if (h.chol_fail) return slam_fail(ctx, SLAM_ERR_NUMERIC, "slam_local_ba: reduced camera system not positive definite (theta and outliers left unchanged)");
    return SLAM_OK;
}


} // extern "C"

// ---------------------------------------------------------------------------------
// pnp_bundle_adjustment (bundle_adjustment.jl:113-171): one pose, n points.  The
// whole two-pass LM (dense 6x6 normal equations, exact Cholesky step) runs inside
// ONE kernel / one workgroup: per iteration two block reductions and a
// single-thread 6x6 solve; no host round trips.

#define PNP_T 256
// (templates + forced inlining keep the partial sums in registers and the LDS / global pointers in their address spaces: the
//  generic version ran with 188 bytes of scratch per lane and FLAT accesses throughout)
typedef const __attribute__((address_space(1))) double *pnp_gcd;
typedef __attribute__((address_space(1))) uint8_t *pnp_gu8;
template <int cnt>
__device__ __forceinline__ void pnp_reduce(double *v, double *sh /* 4*cnt */, double *outv)
{
#pragma unroll
    for (int k = 0; k < cnt; k++) {
        double t = v[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
        v[k] = t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < cnt; k++) sh[(threadIdx.x >> 6) * cnt + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int k = 0; k < cnt; k++) { double t = 0.0; for (int w = 0; w < PNP_T / 64; w++) t += sh[w * cnt + k]; outv[k] = t; }
    __syncthreads();
}

__device__ __forceinline__ int pnp_lm(const PnPArgs &A, pnp_gcd gpx, pnp_gcd gpts, pnp_gu8 goutl, double *X /*shared 6*/, int ignore, int iterations, double *sh, double *red /*shared 40*/,
                      double *Xt /*shared 6*/, double *dxs /*shared 6*/, int *flags /*shared 4*/, double *ssr_out)
{
    const int tid = threadIdx.x, n = A.n;
    double v[28];
    // ssr at X
    v[0] = 0.0;
    for (int i = tid; i < n; i += PNP_T) {
        if (ignore && goutl[i]) continue;
        double r[2];
        obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, nullptr, nullptr, nullptr);
        v[0] += r[0] * r[0] + r[1] * r[1];
    }
    pnp_reduce<1>(v, sh, red);
    double ssr = red[0], delta = LM_DELTA0, decrease = 2.0;
    __shared__ double Hs[36], gs[6];
    int need_jac = 1, converged = 0, iter = 0;
    while (!converged && iter < iterations) {
        iter++;
        if (need_jac) {
#pragma unroll
            for (int k = 0; k < 27; k++) v[k] = 0.0;
            for (int i = tid; i < n; i += PNP_T) {
                if (ignore && goutl[i]) continue;
                double r[2], Jp[12], Jl[6];
                obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, Jp, Jl, nullptr);
                int c = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++) v[c++] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
#pragma unroll
                for (int a = 0; a < 6; a++) v[21 + a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
            }
            pnp_reduce<27>(v, sh, red);
            if (tid == 0) {
                int c = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++) { Hs[a + 6 * b] = red[c]; Hs[b + 6 * a] = red[c]; c++; }
#pragma unroll
                for (int a = 0; a < 6; a++) gs[a] = red[21 + a];
            }
            need_jac = 0;
            __syncthreads();
        }
        if (tid == 0) {
            // 6 x 6 damped normal equations, Cholesky + two triangular solves on one lane; every loop has constant bounds and is
            // unrolled, so H and x live in registers (with run-time bounds they sat in scratch: ~100 dependent scratch round trips
            // per LM iteration)
            double H[36], x[6];
#pragma unroll
            for (int k = 0; k < 36; k++) H[k] = Hs[k];
#pragma unroll
            for (int a = 0; a < 6; a++) { H[a + 6 * a] += fmin(fmax(Hs[a + 6 * a], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * (1 / delta); x[a] = gs[a]; }
            int fail = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                double dj = H[j + 6 * j];
#pragma unroll
                for (int k = 0; k < j; k++) dj -= H[j + 6 * k] * H[j + 6 * k];
                if (!(dj > 0)) fail = 1;
                dj = fail ? 1.0 : sqrt(dj); H[j + 6 * j] = dj;           // (after a failure the remaining values are not used)
#pragma unroll
                for (int i = j + 1; i < 6; i++) {
                    double sv = H[i + 6 * j];
#pragma unroll
                    for (int k = 0; k < j; k++) sv -= H[i + 6 * k] * H[j + 6 * k];
                    H[i + 6 * j] = sv / dj;
                }
            }
            if (!fail) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    double sv = x[i];
#pragma unroll
                    for (int k = 0; k < i; k++) sv -= H[i + 6 * k] * x[k];
                    x[i] = sv / H[i + 6 * i];
                }
#pragma unroll
                for (int i = 5; i >= 0; i--) {
                    double sv = x[i];
#pragma unroll
                    for (int k = i + 1; k < 6; k++) sv -= H[k + 6 * i] * x[k];
                    x[i] = sv / H[i + 6 * i];
                }
            }
#pragma unroll
            for (int a = 0; a < 6; a++) { dxs[a] = fail ? 0.0 : x[a]; Xt[a] = X[a] - dxs[a]; }
            flags[0] = fail;
        }
        __syncthreads();
        if (flags[0]) break;
        v[0] = 0.0; v[1] = 0.0;
        for (int i = tid; i < n; i += PNP_T) {
            if (ignore && goutl[i]) continue;   // zero residual and zero Jacobian row: contributes 0 to both sums
            double r[2], rt[2], Jp[12], Jl[6];
            obs_eval(Xt, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, rt, nullptr, nullptr, nullptr);
            obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, Jp, Jl, nullptr);
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) { a += Jp[k] * dxs[k]; b += Jp[6 + k] * dxs[k]; }
            a -= r[0]; b -= r[1];
            v[0] += rt[0] * rt[0] + rt[1] * rt[1];
            v[1] += a * a + b * b;
        }
        pnp_reduce<2>(v, sh, red);
        const double trial = red[0], pred = red[1];
        double mx = 0.0;
        for (int a = 0; a < 6; a++) mx = fmax(mx, fabs(dxs[a]));
        const double rho = (trial - ssr) / (pred - ssr);
        if (rho > LM_MIN_STEP_QUALITY) {
            const int x_conv = mx <= LM_XTOL;
            const int f_conv = fabs(ssr - trial) / (fabs(ssr) + LM_FTOL) <= LM_FTOL;
            ssr = trial;
            const double u = 2.0 * rho - 1.0;
            delta = fmin(delta / fmax(1.0 / 3.0, 1.0 - u * u * u), LM_MAX_DELTA);
            decrease = 2.0; need_jac = 1;
            converged = x_conv || f_conv;
            __syncthreads();
            if (tid < 6) X[tid] = Xt[tid];
        } else {
            delta = fmax(delta / decrease, LM_MIN_DELTA);
            decrease *= 2.0;
            converged = mx <= LM_XTOL;
        }
        __syncthreads();
    }
    *ssr_out = ssr;
    return iter;
}

__device__ __forceinline__ void pnp_body(const PnPArgs &A)
{
    const pnp_gcd gpx = (pnp_gcd)A.px, gpts = (pnp_gcd)A.pts; const pnp_gu8 goutl = (pnp_gu8)A.outl;
    __shared__ double X[6], Xt[6], dxs[6], sh[4 * 28], red[40];
    __shared__ int flags[4];
    const int tid = threadIdx.x, n = A.n;
    if (tid < 6) X[tid] = A.X0[tid];
    for (int i = tid; i < n; i += PNP_T) goutl[i] = 0;
    __syncthreads();
    double v[1] = {0.0};
    for (int i = tid; i < n; i += PNP_T) {
        double r[2];
        obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, nullptr, nullptr, nullptr);
        v[0] += r[0] * r[0] + r[1] * r[1];
    }
    pnp_reduce<1>(v, sh, red);
    const double err_init = red[0];
    double ssr1 = 0.0, ssr2 = 0.0;
    const int it1 = pnp_lm(A, gpx, gpts, goutl, X, 0, A.iters_fast, sh, red, Xt, dxs, flags, &ssr1);
    v[0] = 0.0;
    for (int i = tid; i < n; i += PNP_T) {
        double r[2], z;
        obs_eval(X, (const double *)(gpts + 3 * i), gpx[2 * i], gpx[2 * i + 1], A.cam, r, nullptr, nullptr, &z);
        const bool o = z < A.depth_eps || (r[0] * r[0] + r[1] * r[1]) > A.repr_eps;
        goutl[i] = o ? 1 : 0;
        v[0] += o ? 1.0 : 0.0;
    }
    __threadfence_block();
    pnp_reduce<1>(v, sh, red);
    const int no = (int)red[0];
    int identity = 0, it2 = 0;
    if (n - no < 5) { identity = 1; ssr2 = ssr1; }
    else it2 = pnp_lm(A, gpx, gpts, goutl, X, 1, A.iterations, sh, red, Xt, dxs, flags, &ssr2);
    __syncthreads();
    if (tid == 0) {
        for (int a = 0; a < 6; a++) A.result[a] = X[a];
        A.result[6] = err_init; A.result[7] = ssr2; A.result[8] = no; A.result[9] = identity; A.result[10] = it1; A.result[11] = it2;
    }
}

__global__ __launch_bounds__(PNP_T) void k_pnp(PnPArgs A) { pnp_body(A); }
// S independent problems, one workgroup each (slam_pnp_ba_batch); the argument blocks live in mapped host memory
__global__ __launch_bounds__(PNP_T) void k_pnp_batch(const PnPArgs *args)
{
    pnp_body(args[blockIdx.x]);                                  // read in place (wave-uniform scalar loads): no LDS copy behind a generic reference
}

int pnp_launch_device(slam_ctx *ctx, int S, const PnPArgs *args_dev)
{
    ProfScope span(ctx, "pnp_ba");
    hipLaunchKernelGGL(k_pnp_batch, dim3(S), dim3(PNP_T), 0, ctx->stream, args_dev);
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}

// RotZYX(pose[1:3,1:3]) -> angles (Rotations.jl), pose column-major: R[i][j] = pose[i + 4j]
static void pnp_pose_to_x(const double *pose_cw, double *X0)
{
    const double R11 = pose_cw[0], R21 = pose_cw[1], R31 = pose_cw[2], R12 = pose_cw[4], R22 = pose_cw[5], R13 = pose_cw[8], R23 = pose_cw[9];
    const double t1 = std::atan2(R21, R11), s1 = std::sin(t1), c1 = std::cos(t1);
    X0[0] = t1; X0[1] = std::atan2(-R31, std::sqrt(R11 * R11 + R21 * R21)); X0[2] = std::atan2(R13 * s1 - R23 * c1, R22 * c1 - R12 * s1);
    X0[3] = pose_cw[12]; X0[4] = pose_cw[13]; X0[5] = pose_cw[14];
}
static void pnp_x_to_pose(const double *res, double *out_pose)
{
    for (int k = 0; k < 16; k++) out_pose[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (res[9] == 0.0) {
        const double s1 = std::sin(res[0]), c1 = std::cos(res[0]), s2 = std::sin(res[1]), c2 = std::cos(res[1]), s3 = std::sin(res[2]), c3 = std::cos(res[2]);
        const double R[9] = {c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3, s1 * c2, s1 * s2 * s3 + c1 * c3, s1 * s2 * c3 - c1 * s3, -s2, c2 * s3, c2 * c3};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out_pose[i + 4 * j] = R[3 * i + j];
        out_pose[12] = res[3]; out_pose[13] = res[4]; out_pose[14] = res[5];
    }
}

// S single-pose refinements in one launch: problem z owns points [offsets[z], offsets[z+1]); cams S x 4 (fx, fy,
// cx, cy), poses_cw / out_poses S x 16 column-major
extern "C" int slam_pnp_ba_batch(slam_ctx *ctx, int S, const int32_t *offsets, const double *cams, const double *poses_cw,
                                 const double *pixels_yx, const double *points_xyz, int iters_fast, int iterations,
                                 double depth_eps, double repr_eps, double *out_poses, double *err_init, double *err_final,
                                 uint8_t *outliers, int *n_outliers)
{
    ARG_TRY(ctx, ctx != nullptr && S >= 0);
    if (S == 0) return SLAM_OK;
    ARG_TRY(ctx, offsets && cams && poses_cw && out_poses && offsets[0] == 0);
    for (int z = 0; z < S; z++) ARG_TRY(ctx, offsets[z + 1] >= offsets[z]);
    const int ntot = offsets[S];
    ARG_TRY(ctx, ntot == 0 || (pixels_yx && points_xyz && outliers));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pxb = al((size_t)ntot * 16 + 8), ptb = al((size_t)ntot * 24 + 8), ob = al((size_t)ntot + 8), rb = al((size_t)S * 128);
    char *s;
    int rc = slam_scratch(ctx, pxb + ptb + ob + rb, (void **)&s);
    if (rc) return rc;
    double *d_px = (double *)s, *d_pts = (double *)(s + pxb); uint8_t *d_o = (uint8_t *)(s + pxb + ptb); double *d_res = (double *)(s + pxb + ptb + ob);
    char *h, *d;
    rc = slam_pinned(ctx, (size_t)S * sizeof(PnPArgs) + (size_t)S * 128, (void **)&h);
    if (rc) return rc;
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&d, h, 0));
    PnPArgs *args = (PnPArgs *)h;
    for (int z = 0; z < S; z++) {
        PnPArgs &A = args[z];
        A.cam = {cams[4 * z], cams[4 * z + 1], cams[4 * z + 2], cams[4 * z + 3]};
        A.n = offsets[z + 1] - offsets[z]; A.iters_fast = iters_fast; A.iterations = iterations;
        A.depth_eps = depth_eps; A.repr_eps = repr_eps;
        pnp_pose_to_x(poses_cw + 16 * z, A.X0);
        A.px = d_px + 2 * (size_t)offsets[z]; A.pts = d_pts + 3 * (size_t)offsets[z]; A.outl = d_o + offsets[z]; A.result = d_res + 16 * z;
    }
    if (ntot > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_px, pixels_yx, (size_t)ntot * 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_pts, points_xyz, (size_t)ntot * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    { ProfScope span(ctx, "pnp_ba");
      hipLaunchKernelGGL(k_pnp_batch, dim3(S), dim3(PNP_T), 0, ctx->stream, (const PnPArgs *)d); }
    HIP_TRY(ctx, hipGetLastError());
    double *res = (double *)(h + (size_t)S * sizeof(PnPArgs));
    HIP_TRY(ctx, hipMemcpyAsync(res, d_res, (size_t)S * 128, hipMemcpyDeviceToHost, ctx->stream));
    if (ntot > 0) HIP_TRY(ctx, hipMemcpyAsync(outliers, d_o, (size_t)ntot, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    for (int z = 0; z < S; z++) {
        const double *r = res + 16 * z;
        if (err_init) err_init[z] = r[6];
        if (err_final) err_final[z] = r[7];
        if (n_outliers) n_outliers[z] = (int)r[8];
        pnp_x_to_pose(r, out_poses + 16 * z);
    }
    return SLAM_OK;
}

extern "C" int slam_pnp_ba(slam_ctx *ctx, double fx, double fy, double cx, double cy,
                           const double pose_cw[16], const double *pixels_yx, const double *points_xyz, int n,
                           int iters_fast, int iterations, double depth_eps, double repr_eps,
                           double out_pose[16], double *err_init, double *err_final, uint8_t *outliers, int *n_outliers)
{
    ARG_TRY(ctx, ctx != nullptr && pose_cw != nullptr && out_pose != nullptr && n >= 0);
    ARG_TRY(ctx, n == 0 || (pixels_yx != nullptr && points_xyz != nullptr && outliers != nullptr));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PnPArgs A;
    A.cam = {fx, fy, cx, cy}; A.n = n; A.iters_fast = iters_fast; A.iterations = iterations;
    A.depth_eps = depth_eps; A.repr_eps = repr_eps;
    pnp_pose_to_x(pose_cw, A.X0);
    const size_t pxb = al((size_t)n * 16 + 8), ptb = al((size_t)n * 24 + 8), ob = al((size_t)n + 8);
    char *s;
    int rc = slam_scratch(ctx, pxb + ptb + ob + 256, (void **)&s);
    if (rc) return rc;
    double *d_px = (double *)s, *d_pts = (double *)(s + pxb); uint8_t *d_o = (uint8_t *)(s + pxb + ptb); double *d_res = (double *)(s + pxb + ptb + ob);
    if (n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_px, pixels_yx, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_pts, points_xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    A.px = d_px; A.pts = d_pts; A.outl = d_o; A.result = d_res;
    hipLaunchKernelGGL(k_pnp, dim3(1), dim3(PNP_T), 0, ctx->stream, A);
    HIP_TRY(ctx, hipGetLastError());
    double res[12];
    HIP_TRY(ctx, hipMemcpyAsync(res, d_res, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
    if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(outliers, d_o, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    if (err_init) *err_init = res[6];
    if (err_final) *err_final = res[7];
    if (n_outliers) *n_outliers = (int)res[8];
    pnp_x_to_pose(res, out_pose);
    return SLAM_OK;
}
