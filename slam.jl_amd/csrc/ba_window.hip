// ba_window.hip -- k_ba_window: the whole two-pass Levenberg-Marquardt of one small window in one (or two) workgroups (src/bundle_adjustment.jl:1-111,
// windows of <= 5 free key-frames: src/estimator.jl:327-331).  Launched by ba_batch.hip.
#include "ba_device.hpp"

// ---- small windows: the WHOLE two-pass Levenberg-Marquardt of one window in ONE workgroup, one launch for the batch -------------------
// The reference's window is at most 5 free key-frames and their constant observers (estimator.jl:327-331): the reduced camera system is
// 30 x 30, a few hundred map points see a free pose at all, and the rest only move themselves.  Spread over the chip kernel by kernel
// (above) such a window costs 5 launches per iteration whose workgroups are mostly latency; here a 512-thread workgroup (or two: below) keeps the
// window to itself for all 5 + 10 iterations -- no launch boundaries, the LM state never leaves the compute unit:
//   A1 thread = observation (coalesced loads, pose data from LDS): residual + Jacobians, stored for the later phases
//   A2 thread = map point: V = sum Jl'Jl + D, V^-1, bl over its (contiguous) observations -- loads only, no evaluation
//   B  the observations of free poses (a host-built list), in chunks that fit LDS: thread = record -> W = Jp'Jl, gradient term;
//      then lane = block pair (a, b) of the 6 x 6 blocks, 32 subsets of 32 lanes walk the chunk's points; fixed-order fold
//      (deterministic, no atomics): S, g, diag U
//   S  dense damped Cholesky of the <= 30 x 30 system by ONE wave (wave-synchronous LDS, no workgroup barriers), L y = g, L' dp = y
//   C1 thread = map point: dl = V^-1 (bl - W' dp), trial point;  C2 thread = observation: trial and predicted residuals
//   D  LeastSquaresOptim's accept / reject (lm_decide), on the device as everywhere
// then the outlier flags between the passes.  128 such windows occupy 128 compute units at once (256 on two workgroups each).  Windows with more free poses, free
// poses that are not consecutive, > 128 poses or > BW_OMAX observations take the batch kernels above.
// (First version, thread = map point with a serial loop over its observations at 512 threads: 235 us per iteration -- two waves per
//  SIMD cannot hide the dependent loads and the Float64 latency of ten evaluations in a row; slower than the kernels it replaces.)
__device__ __forceinline__ double bw_sum(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < BW_T / 64; w++) t += sh[w];
    return t;
}
__device__ __forceinline__ double bw_max(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < BW_T / 64; w++) t = fmax(t, sh[w]);
    return t;
}
__device__ __forceinline__ void bw_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// TWO WORKGROUPS PER WINDOW (two != 0; 128 windows then use all 256 compute units): workgroups b and b + 8 (same XCD) share window
// (b & 7) + 8 (b >> 4); half h takes the map points [0, ksplit) / [ksplit, M) and their observations through every phase, and the two
// exchange (a) the folded partials of the reduced system once per iteration -- both then assemble and solve it, identically --, (b) the
// three sums behind the step decision, (c) the cost at the start of a pass and the outlier count.  The LM state is a copy in LDS that
// both advance identically (half 0 writes it back at the end).  Exchanges go through a double-buffered area in global memory: values and
// a counting flag as agent-scope atomics (performed at the memory side), ordered by s_waitcnt vmcnt(0) -- no cache write-back or invalidate.
__global__ __launch_bounds__(BW_T) void k_ba_window(const BAWin *tab, const int *list, int ns, int two, int iters_fast, int iterations, double repr_eps, double depth_eps, long long xlimit)
{
    const int half = two ? (int)((blockIdx.x >> 3) & 1) : 0;
    const int widx = two ? (int)((blockIdx.x & 7) + 8 * (blockIdx.x >> 4)) : (int)blockIdx.x;
    if (widx >= ns) return;
    BAWin w;
    {   typedef const __attribute__((address_space(4))) unsigned long long *cq_t;
        cq_t q = (cq_t)(const void *)(tab + list[widx]);
        unsigned long long raw[sizeof(BAWin) / 8];
#pragma unroll
        for (int k = 0; k < (int)(sizeof(BAWin) / 8); k++) raw[k] = q[k];
        __builtin_memcpy(&w, raw, sizeof w); }
    BADevG dg;                                                 // the window's arrays as GLOBAL-memory pointers (global_load / global_store, not flat accesses: ba_device.hpp)
    __builtin_memcpy(&dg, &w.d, sizeof dg);
    const BADevG &d = dg;
    extern __shared__ __attribute__((aligned(16))) double bw_sm[];
    const int tid = threadIdx.x;
    const int P = d.P, M = d.M, O = d.O, p0 = w.B.p0, F = w.B.nb, n = 6 * F, nwin = F * (F + 1) / 2;
    double *s_sc = bw_sm;                              // [P][6] sin / cos of the committed poses' angles
    double *s_sct = s_sc + 6 * P;                      // [P][6] of the trial poses
    double *s_tr = s_sct + 6 * P;                      // [P][3] committed translations
    double *s_A = s_tr + 3 * P;                        // [n + 1][n]: damped S (full, row-major), row n = right-hand side
    double *s_dp = s_A + 31 * 30;                      // [32]
    double *s_ud = s_dp + 32;                          // [32] diag U (damping)
    double *s_red = s_ud + 32;                         // [32]
    int *s_flag = (int *)(s_red + 32);                 // [4]
    double *s_pt = s_red + 34;                         // [BW_PC][10] V^-1 (6), bl (3)
    short *s_slot = (short *)(s_pt + BW_PC * 10);      // [BW_PC][BW_FMAX] record of point x for free pose a, or -1
    unsigned char *s_const = (unsigned char *)(s_slot + BW_PC * BW_FMAX);   // [P]
    double *s_W = bw_sm + (((BW_FIXED_DBL(P) * 8 + (size_t)BW_PC * BW_FMAX * 2 + (size_t)P + 15) & ~(size_t)15) >> 3);   // [BW_HC][18]
    double *s_Jp = s_W + BW_HC * 18;                   // [BW_HC][12]
    double *s_gr = s_Jp + BW_HC * 12;                  // [BW_HC][6]
    double *s_fold = s_W;                              // after the last chunk: [15][32 * 18 + n * 7] partials of the waves 1 .. 15
    // phase B: 32 subsets of 32 lanes; lane = (block pair (a <= b), upper / lower three rows of its 6 x 6 block) and / or (slot a2, row r2):
    // half a block per lane keeps the accumulators + one W record under the 128 registers of a 1024-thread workgroup
    const int sub = tid >> 5, wl = tid & 31, w2 = wl >> 1, rh = 3 * (wl & 1);
    int ba_ = 0, bb_ = 0;
    { int r = w2; while (ba_ < F && r >= F - ba_) { r -= F - ba_; ba_++; } bb_ = ba_ + r; }
    const bool live = w2 < nwin, xl = wl < n;
    const int a2 = xl ? wl / 6 : 0, r2 = wl - 6 * (wl / 6);
    __shared__ LMState s_lm;                           // the LM state lives HERE for the whole solve (both halves of a split window advance their copies identically)
    LMState *s = &s_lm;
    if (tid == 0) s_lm = *(const LMState *)d.st;
    // this workgroup's map points [kLo, kHi) (sorted order) and observations [oLo, oHi)
    const int kLo = two && half ? w.ksplit : 0, kHi = two && !half ? w.ksplit : M;
    const int oLo = d.pt_start[kLo], oHi = d.pt_start[kHi];
    // exchange with the other half: own values -> area [half][e & 1], flag[half] = e; wait for flag[1 - half] >= e; the sums are own + other
    int xe = 0, xdead = 0;             // xdead: this half gave up waiting (lane 0 of wave 0 keeps it)
    int *xflag = (int *)w.bwx;
    auto xarea = [&](int h, int e) { return w.bwx + 8 + (size_t)(2 * h + (e & 1)) * 832; };
    auto xpost = [&](int e) {          // (called by the wave that wrote the values)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((tid & 63) == 0 && !xdead) __hip_atomic_store(xflag + half, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // The wait is BOUNDED (xlimit ticks of the 100 MHz wall clock): the launch is not cooperative, so nothing but the host's count of
    // compute units promises that the partner workgroup is resident.  A half that runs out of patience marks the window (xflag[2]), posts a
    // flag no later wait can miss (the partner never waits for it again) and goes on WITHOUT waiting -- it only ever exchanges doubles, every
    // loop bound is an iteration count, so the garbage it then computes ends by itself -- and half 0 reports chol_fail = 2: the host solves
    // the call again on one workgroup per window.  A missing partner is an error code, never a hung queue.
    auto xwait = [&](int e) {
        if ((tid & 63) == 0 && !xdead) {
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(xflag + (1 - half), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
                if ((long long)wall_clock64() - t0 > xlimit) {
                    xdead = 1;
                    __hip_atomic_store(xflag + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(xflag + half, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
    };
    // three scalars (two sums, one maximum) across the halves; every thread holds the workgroup's values on entry and the window's on return
    auto xscal = [&](double &a, double &b, double &c) {
        if (!two) return;
        const int e = ++xe;
        if (tid == 0) {
            double *o = xarea(half, e) + 800;
            __hip_atomic_store(o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(o + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(o + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 64) {
            xpost(e); xwait(e);
            if (tid == 0) {
                const double *q = xarea(1 - half, e) + 800;
                const double a1 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c1 = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_red[8] = half ? a1 + a : a + a1; s_red[9] = half ? b1 + b : b + b1; s_red[10] = fmax(c, c1);      // (half 0's value first on both sides: the same bits)
            }
        }
        __syncthreads();
        a = s_red[8]; b = s_red[9]; c = s_red[10];
        __syncthreads();
    };
    for (int p = tid; p < P; p += BW_T) s_const[p] = d.pconst[p];
    // phase A's chunks: npc consecutive map points (sorted order) per wave and trip, at most BW_WOB observations (d.sg_ob = most observations of one point)
    const int npc_a = max(1, min(64, BW_WOB / d.sg_ob));
#ifdef BW_TRACE
    long long bw_clk[12];
#endif
    // committed pose data -> LDS (sin / cos of the angles, translation)
    auto stage_poses = [&](const ParamBufsG &pb) {
        for (int p = tid; p < P; p += BW_T) {
            { const double ang[3] = {pb.pose[6 * p], pb.pose[6 * p + 1], pb.pose[6 * p + 2]}; pose_sincos(ang, s_sc + 6 * p); }
            s_tr[3 * p] = pb.pose[6 * p + 3]; s_tr[3 * p + 1] = pb.pose[6 * p + 4]; s_tr[3 * p + 2] = pb.pose[6 * p + 5];
        }
    };

    auto pbufs = [&]() { const bool sw = s->cur != 0; return ParamBufsG{sw ? d.pose_t : d.pose, sw ? d.pts_t : d.pts, sw ? d.pose : d.pose_t, sw ? d.pts : d.pts_t, nullptr, nullptr}; };
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {
        const int ignore = pass, iters = pass ? iterations : iters_fast;
        // ---- cost at the committed parameters (LeastSquaresOptim evaluates f!(fcur, x) first)
        {
            const ParamBufsG pb = pbufs();
            __syncthreads();
            stage_poses(pb);
            __syncthreads();
            double ss = 0.0;
            for (int i = oLo + tid; i < oHi; i += BW_T) {
                if (ignore && d.outl[i]) continue;
                const int p = d.opose[i], j = d.opoint[i];
                const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                double r[2];
                obs_eval_sc(s_sc + 6 * p, s_tr + 3 * p, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
                ss += r[0] * r[0] + r[1] * r[1];
            }
            double t = bw_sum(ss, s_red), tz1 = 0.0, tz2 = 0.0;
            xscal(t, tz1, tz2);
            if (tid == 0) {
                s->ssr = t;
                if (pass == 0) { s->ssr_init = t; s->chol_fail = 0; s->n_outliers = 0; }
                s->delta = LM_DELTA0; s->decrease_factor = 2.0; s->converged = 0; s->accept = 0; s->iters = 0;
            }
            __syncthreads();
        }
        for (int it = 1; it <= iters; it++) {
            if (s->converged) break;                           // (uniform: every thread reads the flag after a barrier)
            const ParamBufsG pb = pbufs();
            const double inv_delta = 1.0 / s->delta;
            BW_CLK(0);
            stage_poses(pb);
            __syncthreads();
            BW_CLK(1);
            // ---- A: every WAVE takes chunks of npc_a consecutive map points (their observations are contiguous: coalesced loads, lane = observation):
            //      residual + Jacobians (stored for the later phases), the nine products Jl'Jl / Jl'f into the wave's own LDS block; then lane = map
            //      point of the chunk: V = sum + D, V^-1, bl in the observations' order.  Wave-synchronous -- no workgroup barrier inside the phase,
            //      the eight waves overlap each other's latencies.  (Versions before: thread = point summing from the stored records -- 64 cache
            //      lines per load instruction, 127 k cycles; workgroup-wide tiles with two barriers each -- 171 k.)  V^-1 / bl / dl: SORTED point order.
            {
                const int wvA = tid >> 6, ln = tid & 63;
                double *s_w9 = s_W + (size_t)wvA * BW_WOB * 9;
                for (int k0 = kLo + wvA * npc_a; k0 < kHi; k0 += (BW_T / 64) * npc_a) {
                    const int k1 = min(kHi, k0 + npc_a), o0 = d.pt_start[k0], nobs = d.pt_start[k1] - o0;
                    for (int t = ln; t < nobs; t += 64) {
                        const int i = o0 + t;
                        const int p = d.opose[i], j = d.opoint[i];
                        const bool active = !(ignore && d.outl[i]);
                        const bool hp = active && !s_const[p];
                        double r[2] = {0.0, 0.0}, Jp[12], Jl[6] = {0, 0, 0, 0, 0, 0};
                        if (active) {
                            const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                            obs_eval_sc(s_sc + 6 * p, s_tr + 3 * p, X, d.pix[i], d.pix[O + i], d.cam, r, hp ? Jp : nullptr, Jl, nullptr);
                        }
                        d.hasp[i] = hp ? 1 : 0;
                        st_rec<2>(d.f + 2 * (size_t)i, r);
                        st_rec<6>(d.Jl + (size_t)i * 6, Jl);
                        if (hp) st_rec<12>(d.Jp + (size_t)i * 12, Jp);
                        double *v = s_w9 + t * 9;
                        v[0] = Jl[0] * Jl[0] + Jl[3] * Jl[3]; v[1] = Jl[0] * Jl[1] + Jl[3] * Jl[4]; v[2] = Jl[0] * Jl[2] + Jl[3] * Jl[5];
                        v[3] = Jl[1] * Jl[1] + Jl[4] * Jl[4]; v[4] = Jl[1] * Jl[2] + Jl[4] * Jl[5]; v[5] = Jl[2] * Jl[2] + Jl[5] * Jl[5];
#pragma unroll
                        for (int c = 0; c < 3; c++) v[6 + c] = Jl[c] * r[0] + Jl[3 + c] * r[1];
                    }
                    bw_wave_sync();
                    if (ln < k1 - k0) {
                        const int k = k0 + ln;
                        double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                        const int t0 = d.pt_start[k] - o0, t1 = d.pt_start[k + 1] - o0;
                        for (int t = t0; t < t1; t++) {
#pragma unroll
                            for (int c = 0; c < 9; c++) V[c] += s_w9[t * 9 + c];
                        }
                        V[0] += fmin(fmax(V[0], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                        V[3] += fmin(fmax(V[3], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                        V[5] += fmin(fmax(V[5], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                        double Vi[6];
                        inv3_sym(V, Vi);
#pragma unroll
                        for (int c = 0; c < 6; c++) d.Vinv[(size_t)c * M + k] = Vi[c];
#pragma unroll
                        for (int c = 0; c < 3; c++) d.bl[(size_t)c * M + k] = V[6 + c];
                    }
                    bw_wave_sync();
                }
            }
            __syncthreads();
            BW_CLK(2);
            // ---- B: the reduced camera system from the observations of free poses (records fobs[0 .. NF)), chunk by chunk
            double acc[18], ex[7];
#pragma unroll
            for (int k = 0; k < 18; k++) acc[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 7; k++) ex[k] = 0.0;
            for (int k0 = kLo; k0 < kHi;) {
                // the chunk [k0, k1): <= BW_PC points, <= BW_HC records (pfs = running count of free-pose observations by sorted point)
                const int base = d.pfs[k0];
                int lo = k0 + 1, hi = min(kHi, k0 + BW_PC);
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (d.pfs[mid] - base <= BW_HC) lo = mid; else hi = mid - 1; }
                const int k1 = lo, npc = k1 - k0, nrec = d.pfs[k1] - base;
                if (nrec == 0) { k0 = k1; continue; }                    // no point of the chunk sees a free pose
                for (int x = tid; x < npc; x += BW_T) {
                    const int kk = k0 + x;
#pragma unroll
                    for (int c = 0; c < 6; c++) s_pt[x * 10 + c] = d.Vinv[(size_t)c * M + kk];
#pragma unroll
                    for (int c = 0; c < 3; c++) s_pt[x * 10 + 6 + c] = d.bl[(size_t)c * M + kk];
#pragma unroll
                    for (int a = 0; a < BW_FMAX; a++) s_slot[x * BW_FMAX + a] = -1;
                }
                __syncthreads();
                for (int rec = tid; rec < nrec; rec += BW_T) {           // thread = record
                    const int i = d.fobs[base + rec];
                    if (!d.hasp[i]) continue;                            // an ignored outlier: no slot
                    const int x = d.opk[i] - k0, p = d.opose[i];
                    double jp[12], jl[6], ff[2], Vi[6], bl[3];
                    ld_rec<12>(d.Jp + (size_t)i * 12, jp); ld_rec<6>(d.Jl + (size_t)i * 6, jl); ld_rec<2>(d.f + 2 * (size_t)i, ff);
#pragma unroll
                    for (int c = 0; c < 6; c++) Vi[c] = s_pt[x * 10 + c];
#pragma unroll
                    for (int c = 0; c < 3; c++) bl[c] = s_pt[x * 10 + 6 + c];
                    const double vb0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
                    const double vb1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
                    const double vb2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        const double w0 = jp[a] * jl[0] + jp[6 + a] * jl[3];
                        const double w1 = jp[a] * jl[1] + jp[6 + a] * jl[4];
                        const double w2 = jp[a] * jl[2] + jp[6 + a] * jl[5];
                        s_W[rec * 18 + 3 * a] = w0; s_W[rec * 18 + 3 * a + 1] = w1; s_W[rec * 18 + 3 * a + 2] = w2;
                        s_gr[rec * 6 + a] = (jp[a] * ff[0] + jp[6 + a] * ff[1]) - (w0 * vb0 + w1 * vb1 + w2 * vb2);
                    }
#pragma unroll
                    for (int c = 0; c < 12; c++) s_Jp[rec * 12 + c] = jp[c];
                    s_slot[x * BW_FMAX + (p - p0)] = (short)rec;
                }
                __syncthreads();
                if (live)
                    for (int x = sub; x < npc; x += BW_T / 32) {
                        const int ta = s_slot[x * BW_FMAX + ba_], tb = s_slot[x * BW_FMAX + bb_];
                        if (ta < 0 || tb < 0) continue;
                        double Vi[6], Wb[18];
                        ld_rec<6>(s_pt + x * 10, Vi); ld_rec<18>(s_W + tb * 18, Wb);
#pragma unroll
                        for (int rr = 0; rr < 3; rr++) {
                            const double *wa = s_W + ta * 18 + 3 * (rh + rr);
                            const double a0 = wa[0], a1 = wa[1], a2v = wa[2];
                            const double T0 = fma(a2v, Vi[2], fma(a1, Vi[1], a0 * Vi[0]));
                            const double T1 = fma(a2v, Vi[4], fma(a1, Vi[3], a0 * Vi[1]));
                            const double T2 = fma(a2v, Vi[5], fma(a1, Vi[4], a0 * Vi[2]));
#pragma unroll
                            for (int c = 0; c < 6; c++)
                                acc[6 * rr + c] = fma(-T2, Wb[3 * c + 2], fma(-T1, Wb[3 * c + 1], fma(-T0, Wb[3 * c], acc[6 * rr + c])));
                        }
                    }
                if (xl)
                    for (int x = sub; x < npc; x += BW_T / 32) {
                        const int ta = s_slot[x * BW_FMAX + a2];
                        if (ta < 0) continue;
                        double J[12];
                        ld_rec<12>(s_Jp + ta * 12, J);
                        const double j0 = s_Jp[ta * 12 + r2], j1 = s_Jp[ta * 12 + 6 + r2];      // (two more LDS reads instead of two five-deep select chains over J)
#pragma unroll
                        for (int c = 0; c < 6; c++) ex[c] = fma(j1, J[6 + c], fma(j0, J[c], ex[c]));
                        ex[6] += s_gr[ta * 6 + r2];
                    }
                __syncthreads();
                k0 = k1;
            }
            BW_CLK(3);
            // fold the 32 subsets in a fixed order: the two subsets of a wave by a lane exchange, the 16 waves through LDS by wave 0
#pragma unroll
            for (int k = 0; k < 18; k++) acc[k] += __shfl_xor(acc[k], 32);
#pragma unroll
            for (int k = 0; k < 7; k++) ex[k] += __shfl_xor(ex[k], 32);
            const int fstride = 32 * 18 + n * 7, wv = tid >> 6;
            if (wv >= 1 && (tid & 63) < 32) {
                if (live) st_rec<18>(s_fold + (size_t)(wv - 1) * fstride + wl * 18, acc);
                if (xl) {
#pragma unroll
                    for (int k = 0; k < 7; k++) s_fold[(size_t)(wv - 1) * fstride + 32 * 18 + wl * 7 + k] = ex[k];
                }
            }
            __syncthreads();
            if (tid < 32) {
                if (live)
                    for (int q = 0; q < BW_T / 64 - 1; q++) {
                        double o[18];
                        ld_rec<18>(s_fold + (size_t)q * fstride + wl * 18, o);
#pragma unroll
                        for (int k = 0; k < 18; k++) acc[k] += o[k];
                    }
                if (xl)
                    for (int q = 0; q < BW_T / 64 - 1; q++) {
#pragma unroll
                        for (int k = 0; k < 7; k++) ex[k] += s_fold[(size_t)q * fstride + 32 * 18 + wl * 7 + k];
                    }
            }
            if (two) {                                         // ... and the other half's: own + other, lane by lane (wave 0; every lane of the 32 writes all its 25 values)
                const int e = ++xe;
                if (tid < 64) {
                    if (tid < 32) {
                        double *o = xarea(half, e) + wl * 25;
#pragma unroll
                        for (int k = 0; k < 18; k++) __hip_atomic_store(o + k, acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int k = 0; k < 7; k++) __hip_atomic_store(o + 18 + k, ex[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    xpost(e); xwait(e);
                    if (tid < 32) {
                        const double *q = xarea(1 - half, e) + wl * 25;
#pragma unroll
                        for (int k = 0; k < 18; k++) { const double v = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc[k] = half ? v + acc[k] : acc[k] + v; }
#pragma unroll
                        for (int k = 0; k < 7; k++) { const double v = __hip_atomic_load(q + 18 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ex[k] = half ? v + ex[k] : ex[k] + v; }
                    }
                }
            }
            __syncthreads();                                   // the fold buffer is read: the system goes where no partial lives (s_A)
            // ---- S: wave 0 alone assembles and solves (wave-synchronous LDS: a wave's DS instructions execute in order)
            if (tid < 64) {
                if (tid < 32 && xl) {
                    const double ud = r2 == 0 ? ex[0] : r2 == 1 ? ex[1] : r2 == 2 ? ex[2] : r2 == 3 ? ex[3] : r2 == 4 ? ex[4] : ex[5];
                    s_A[n * n + wl] = ex[6];                   // right-hand side row
                    s_ud[wl] = ud;
#pragma unroll
                    for (int c = 0; c < 6; c++) s_fold[a2 * 36 + r2 * 6 + c] = ex[c];      // Jp'Jp row r2 of slot a2 -> the diagonal block
                }
                bw_wave_sync();
                if (tid < 32 && live) {
                    if (ba_ == bb_) {
#pragma unroll
                        for (int k = 0; k < 18; k++) acc[k] += s_fold[ba_ * 36 + 6 * rh + k];
                    }
#pragma unroll
                    for (int rr = 0; rr < 3; rr++)
#pragma unroll
                        for (int c = 0; c < 6; c++) {
                            s_A[(6 * ba_ + rh + rr) * n + 6 * bb_ + c] = acc[6 * rr + c];
                            if (ba_ != bb_) s_A[(6 * bb_ + c) * n + 6 * ba_ + rh + rr] = acc[6 * rr + c];
                        }
                }
                bw_wave_sync();
                BW_CLK(4);
                // damped Cholesky A = L L': lane i keeps row i of the lower triangle in REGISTERS (lane n: the right-hand side row -- the forward
                // substitution comes for free); the entries of row jc a step needs are lane broadcasts (v_readlane), not LDS round trips
                // (a first version walked the rows in LDS: 79 k cycles per solve, every multiply-add behind an exposed LDS latency)
                if (tid < n) s_A[tid * n + tid] += fmin(fmax(s_ud[tid], LM_MIN_DIAGONAL), LM_MAX_DIAGONAL) * inv_delta;
                bw_wave_sync();
                double a[6 * BW_FMAX];
#pragma unroll
                for (int k = 0; k < 6 * BW_FMAX; k++) a[k] = (tid <= n && k < n) ? s_A[tid * n + k] : 0.0;
                bool bad = false;
#pragma unroll
                for (int jc = 0; jc < 6 * BW_FMAX; jc++) {
                    if (jc < n) {
                        // (four partial sums: a lone wave pays 36 cycles for a DEPENDENT Float64 operation, 9.5 for an independent one; 1 / sqrt from
                        //  v_rsq_f64 + one Newton step, 4e-15 relative, as in k_band_solve -- the solve: 31 k -> 20 k cycles)
                        double sq[4] = {a[jc], 0.0, 0.0, 0.0};
#pragma unroll
                        for (int k = 0; k < jc; k++)
                            sq[k & 3] -= a[k] * __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[k]), jc), __builtin_amdgcn_readlane(__double2loint(a[k]), jc));
                        const double sum = (sq[0] + sq[1]) + (sq[2] + sq[3]);
                        const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sum), jc), __builtin_amdgcn_readlane(__double2loint(sum), jc));
                        const bool okp = piv > 0.0 && piv < 1e300;
                        if (!okp) bad = true;
                        const double pv = okp ? piv : 1.0;
                        const double y0 = __builtin_amdgcn_rsq(pv);
                        const double r0 = __builtin_fma(-(pv * y0), y0, 1.0), rd = __builtin_fma(y0 * 0.5, r0, y0);
                        a[jc] = tid == jc ? pv * rd : sum * rd;
                    }
                }
                // L (rows 0 .. n - 1) and y' = (L^-1 g)' (row n) back to LDS; then lane i takes COLUMN i of L and the back-substitution
                // L' dp = y runs in registers too
                if (tid <= n) {
#pragma unroll
                    for (int k = 0; k < 6 * BW_FMAX; k++) if (k < n) s_A[tid * n + k] = a[k];
                }
                bw_wave_sync();
                double y = tid < n ? s_A[n * n + tid] : 0.0;
                const double dg = 1.0 / (tid < n ? s_A[tid * n + tid] : 1.0);      // (one division per lane, all at once; the chain multiplies)
#pragma unroll
                for (int k = 0; k < 6 * BW_FMAX; k++) a[k] = (tid < n && k < n && k > tid) ? s_A[k * n + tid] : 0.0;      // a[k] = L[k][tid]
#pragma unroll
                for (int jc = 6 * BW_FMAX - 1; jc >= 0; jc--) {
                    if (jc < n) {
                        const double yj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), jc), __builtin_amdgcn_readlane(__double2loint(y), jc));
                        const double dj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dg), jc), __builtin_amdgcn_readlane(__double2loint(dg), jc));
                        const double xj = yj * dj;
                        if (tid == jc) y = xj;
                        if (tid < jc) y -= a[jc] * xj;
                    }
                }
                if (tid < n) s_dp[tid] = y;
                if (tid == 0) s_flag[0] = bad ? 1 : 0;
            }
            __syncthreads();
            if (s_flag[0]) { if (tid == 0) s->chol_fail = 1; }
            BW_CLK(5);
            // ---- C1: trial poses; thread = map point: dl, trial point
            double mx = 0.0;
            if (tid < n) { const int p = p0 + tid / 6, c = tid - 6 * (tid / 6); const double v = s_dp[tid]; pb.pose_t[6 * p + c] = pb.pose[6 * p + c] - v; mx = fabs(v); }
            for (int p = tid; p < P; p += BW_T) {
                if (s_const[p]) {
#pragma unroll
                    for (int c = 0; c < 6; c++) { pb.pose_t[6 * p + c] = pb.pose[6 * p + c]; s_sct[6 * p + c] = s_sc[6 * p + c]; }
                } else {
                    const int a = p - p0;
                    const double tp[3] = {pb.pose[6 * p] - s_dp[6 * a], pb.pose[6 * p + 1] - s_dp[6 * a + 1], pb.pose[6 * p + 2] - s_dp[6 * a + 2]};
                    pose_sincos(tp, s_sct + 6 * p);
                }
            }
            for (int k = kLo + tid; k < kHi; k += BW_T) {
                const int j = d.pt_id[k];
                double bl[3] = {d.bl[k], d.bl[(size_t)M + k], d.bl[(size_t)2 * M + k]};
                for (int rec = d.pfs[k]; rec < d.pfs[k + 1]; rec++) {      // the point's observations of free poses (host list): bl -= Jl' (Jp dp)
                    const int i = d.fobs[rec];
                    if (!d.hasp[i]) continue;
                    const int a = d.opose[i] - p0;
                    double jp[12], jl[6];
                    ld_rec<12>(d.Jp + (size_t)i * 12, jp); ld_rec<6>(d.Jl + (size_t)i * 6, jl);
                    double ua = 0.0, ub = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; c++) { ua += jp[c] * s_dp[6 * a + c]; ub += jp[6 + c] * s_dp[6 * a + c]; }
#pragma unroll
                    for (int c = 0; c < 3; c++) bl[c] -= jl[c] * ua + jl[3 + c] * ub;
                }
                double Vi[6];
#pragma unroll
                for (int c = 0; c < 6; c++) Vi[c] = d.Vinv[(size_t)c * M + k];
                const double l0 = Vi[0] * bl[0] + Vi[1] * bl[1] + Vi[2] * bl[2];
                const double l1 = Vi[1] * bl[0] + Vi[3] * bl[1] + Vi[4] * bl[2];
                const double l2 = Vi[2] * bl[0] + Vi[4] * bl[1] + Vi[5] * bl[2];
                d.dl[3 * k] = l0; d.dl[3 * k + 1] = l1; d.dl[3 * k + 2] = l2;
                pb.pts_t[3 * j] = pb.pts[3 * j] - l0; pb.pts_t[3 * j + 1] = pb.pts[3 * j + 1] - l1; pb.pts_t[3 * j + 2] = pb.pts[3 * j + 2] - l2;
                mx = fmax(mx, fmax(fabs(l0), fmax(fabs(l1), fabs(l2))));
            }
            __syncthreads();
            // ---- C2: thread = observation: trial and predicted residuals
            double st = 0.0, sp = 0.0;
            for (int i = oLo + tid; i < oHi; i += BW_T) {
                const int p = d.opose[i], j = d.opoint[i];
                const bool active = !(ignore && d.outl[i]);
                double jl[6], ff[2], r[2] = {0.0, 0.0}, pa = 0.0, pbv = 0.0;
                ld_rec<6>(d.Jl + (size_t)i * 6, jl); ld_rec<2>(d.f + 2 * (size_t)i, ff);
                const int ks = d.opk[i];
                const double l0 = d.dl[3 * ks], l1 = d.dl[3 * ks + 1], l2 = d.dl[3 * ks + 2];
                const bool fr = !s_const[p];
                const int a = p - p0;
                if (d.hasp[i]) {
                    double jp[12];
                    ld_rec<12>(d.Jp + (size_t)i * 12, jp);
#pragma unroll
                    for (int c = 0; c < 6; c++) { pa += jp[c] * s_dp[6 * a + c]; pbv += jp[6 + c] * s_dp[6 * a + c]; }
                }
                if (active) {
                    const double Xt[3] = {pb.pts_t[3 * j], pb.pts_t[3 * j + 1], pb.pts_t[3 * j + 2]};
                    const double tr[3] = {s_tr[3 * p] - (fr ? s_dp[6 * a + 3] : 0.0), s_tr[3 * p + 1] - (fr ? s_dp[6 * a + 4] : 0.0), s_tr[3 * p + 2] - (fr ? s_dp[6 * a + 5] : 0.0)};
                    obs_eval_sc(s_sct + 6 * p, tr, Xt, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, nullptr);
                }
                pa += jl[0] * l0 + jl[1] * l1 + jl[2] * l2; pbv += jl[3] * l0 + jl[4] * l1 + jl[5] * l2;
                pa -= ff[0]; pbv -= ff[1];
                st += r[0] * r[0] + r[1] * r[1];
                sp += pa * pa + pbv * pbv;
            }
            BW_CLK(6);
            double tt = bw_sum(st, s_red), tp = bw_sum(sp, s_red), tm = bw_max(mx, s_red);
            xscal(tt, tp, tm);
            // ---- D
            if (tid == 0) { s->trial_ssr = tt; s->pred_ssr = tp; s->maxdx = tm; lm_decide(s, tt, tp, tm); }
            __syncthreads();
#ifdef BW_TRACE
            if (tid == 0 && (blockIdx.x == 5 || blockIdx.x == 13) && pass == 0 && it == 3) { bw_clk[7] = clock64();
                printf("k_ba_window (M %d, O %d, F %d, %d free-pose observations): sincos %lld | A %lld | B chunks %lld | fold %lld | solve %lld | C %lld | reduce + decide %lld cycles\n", M, O, F, d.pfs[M],
                       bw_clk[1] - bw_clk[0], bw_clk[2] - bw_clk[1], bw_clk[3] - bw_clk[2], bw_clk[4] - bw_clk[3], bw_clk[5] - bw_clk[4], bw_clk[6] - bw_clk[5], bw_clk[7] - bw_clk[6]); }
#endif
        }
        if (tid == 0) { if (pass == 0) { s->ssr_pass1 = s->ssr; s->iters_pass1 = s->iters; } else { s->ssr_final = s->ssr; s->iters_pass2 = s->iters; } }
        if (pass == 0) {
            // ---- _ba_detect_outliers! at theta_1 (bundle_adjustment.jl:90-111)
            __syncthreads();
            const ParamBufsG pb = pbufs();
            stage_poses(pb);
            __syncthreads();
            double cnt = 0.0;
            for (int i = oLo + tid; i < oHi; i += BW_T) {
                const int p = d.opose[i], j = d.opoint[i];
                const double X[3] = {pb.pts[3 * j], pb.pts[3 * j + 1], pb.pts[3 * j + 2]};
                double r[2], z;
                obs_eval_sc(s_sc + 6 * p, s_tr + 3 * p, X, d.pix[i], d.pix[O + i], d.cam, r, nullptr, nullptr, &z);
                const bool out = z < depth_eps || (r[0] * r[0] + r[1] * r[1]) > repr_eps;
                d.outl[i] = out ? 1 : 0;
                cnt += out ? 1.0 : 0.0;
            }
            double tc = bw_sum(cnt, s_red), tz1 = 0.0, tz2 = 0.0;
            xscal(tc, tz1, tz2);
            if (tid == 0) s->n_outliers = (int)tc;
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0 && half == 0) {
        if (two && (xdead || __hip_atomic_load(xflag + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) s_lm.chol_fail = 2;     // a half gave up waiting: nothing of this window is valid
        *(LMState *)d.st = s_lm;
    }
}

