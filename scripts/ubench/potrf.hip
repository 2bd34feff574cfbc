// Cycle count of the BA's 32x32 tile factorisation + inverse (tile_potrf_inv, csrc/ba_device.hpp) on one workgroup.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I slam.jl_amd/csrc scripts/ubench/potrf.hip -o scripts/ubench/potrf && scripts/ubench/potrf
#include "../../slam.jl_amd/csrc/ctx.hip"
#include "../../slam.jl_amd/csrc/ba_device.hpp"
#include <vector>
#include <cstdio>

__global__ __launch_bounds__(256) void k_bench(const double *A, double *out, long long *cycles, int reps, int *fail)
{
    __shared__ double t[CT][CT + 1], inv[CT][CT + 1];
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; t[i][j] = i >= j ? A[i + CT * j] : 0.0; }
    __syncthreads();
    const long long c0 = clock64();
    tile_potrf_inv(t, inv, CT, CT, fail);
    __syncthreads();
    const long long c1 = clock64();
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; out[e] = t[i][j]; out[CT * CT + e] = inv[i][j]; }
    if (threadIdx.x == 0) *cycles = c1 - c0;
}


__device__ __forceinline__ double bcast_lane(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane), hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
// ---- experimental variants -------------------------------------------------------------------------------------
#ifndef RSQ
#define RSQ(x) rsqrt(x)
#endif
template <int MODE> __device__ __forceinline__ void potrf_var(double (*t)[CT + 1], double (*inv)[CT + 1], int h, int w, int *fail)
{
    __shared__ double s_rdiag[CT];
    __shared__ volatile int s_prog;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane < CT ? lane : CT - 1;
    if (threadIdx.x == 0) s_prog = 0;
    __syncthreads();
    if (wv == 0) {
        double lrow[CT];
        double dsq = 0.0;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < CT; j++) {
            const bool active = j < w;
            double acc = t[li][j];
            double dj = t[j][j];
            if (MODE & 2) {
                double pa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < j; m++) pa[m & 3] += lrow[m] * t[j][m];
                acc -= (pa[0] + pa[1]) + (pa[2] + pa[3]);
                dj -= bcast_lane(dsq, j);                            // sum of squares of row j so far, kept by lane j
            } else {
                double pa[4] = {0.0, 0.0, 0.0, 0.0}, pd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < j; m++) {
                    const double ljm = t[j][m];
                    pa[m & 3] += lrow[m] * ljm;
                    pd[m & 3] += ljm * ljm;
                }
                acc -= (pa[0] + pa[1]) + (pa[2] + pa[3]);
                dj -= (pd[0] + pd[1]) + (pd[2] + pd[3]);
            }
            bad = bad || (active && !(dj > 0));
            dj = (active && dj > 0) ? dj : 1.0;
            const double rd = RSQ(dj);
            const double l = (lane == j) ? dj * rd : acc * rd;
            lrow[j] = l;
            dsq += l * l;
            if (active && lane >= j && lane < CT) t[lane][j] = l;
            if (MODE & 1) {
                if (lane == 0) s_rdiag[j] = rd;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) s_prog = j + 1;
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (bad && lane == 0) *fail = 1;
    } else if (wv == 1 && (MODE & 1)) {
        double x[CT];
#pragma unroll
        for (int i = 0; i < CT; i++) {
            while (s_prog <= i) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            double ps[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < i; m++) ps[m & 3] += t[i][m] * x[m];
            const double sacc = ((i == lane) ? 1.0 : 0.0) - ((ps[0] + ps[1]) + (ps[2] + ps[3]));
            x[i] = (i < w && lane <= i) ? sacc * s_rdiag[i] : 0.0;
        }
        if (lane < CT) {
#pragma unroll
            for (int i = 0; i < CT; i++) inv[i][lane] = (lane < w) ? x[i] : 0.0;
        }
    }
}
template <int MODE> __global__ __launch_bounds__(256) void k_var(const double *A, double *out, long long *cycles, int *fail)
{
    __shared__ double t[CT][CT + 1], inv[CT][CT + 1];
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; t[i][j] = i >= j ? A[i + CT * j] : 0.0; }
    __syncthreads();
    const long long c0 = clock64();
    potrf_var<MODE>(t, inv, CT, CT, fail);
    __syncthreads();
    const long long c1 = clock64();
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; out[e] = t[i][j]; out[CT * CT + e] = inv[i][j]; }
    if (threadIdx.x == 0) *cycles = c1 - c0;
}


// ---- pipelined variant: wave 0 factors with the broadcast row prefetched one step ahead (its newest element through
// a constant-lane readlane), then inverts from the finished rows
__device__ __forceinline__ void potrf_pipe(double (*t)[CT + 1], double (*inv)[CT + 1], int h, int w, int *fail)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane < CT ? lane : CT - 1;
    __syncthreads();
    if (wv == 0) {
        double lrow[CT], rdv[CT], rj[CT], rn[CT];
        bool bad = false;
        double aij = t[li][0], ajj = t[0][0];
#pragma unroll
        for (int j = 0; j < CT; j++) {
            const bool active = j < w;
            // (1) prefetch for step j+1: row j+1 of L up to column j-1 (final since earlier steps), a_{i,j+1}, a_{j+1,j+1}
            double aij_n = 0.0, ajj_n = 1.0;
            if (j + 1 < CT) {
#pragma unroll
                for (int m = 0; m < j; m++) rn[m] = t[j + 1][m];
                aij_n = t[li][j + 1]; ajj_n = t[j + 1][j + 1];
            }
            // (2) this step
            double acc = aij, dj = ajj;
            {
                double pa[4] = {0.0, 0.0, 0.0, 0.0}, pd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < j; m++) { pa[m & 3] += lrow[m] * rj[m]; pd[m & 3] += rj[m] * rj[m]; }
                acc -= (pa[0] + pa[1]) + (pa[2] + pa[3]);
                dj -= (pd[0] + pd[1]) + (pd[2] + pd[3]);
            }
            bad = bad || (active && !(dj > 0));
            dj = (active && dj > 0) ? dj : 1.0;
            const double rd = rsqrt(dj);
            const double l = (lane == j) ? dj * rd : acc * rd;
            lrow[j] = l; rdv[j] = rd;
            if (active && lane >= j && lane < CT) t[lane][j] = l;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            // (3) the newest element of row j+1 comes straight from lane j+1
            if (j + 1 < CT) rn[j] = bcast_lane(l, j + 1);
#pragma unroll
            for (int m = 0; m <= j; m++) rj[m] = rn[m];
            aij = aij_n; ajj = ajj_n;
        }
        if (bad && lane == 0) *fail = 1;
        // inverse: lane c solves L x = e_c, rows of L as broadcast reads prefetched one step ahead
        double x[CT];
#pragma unroll
        for (int m = 0; m < CT; m++) rj[m] = 0.0;
#pragma unroll
        for (int i = 0; i < CT; i++) {
            if (i + 1 < CT) {
#pragma unroll
                for (int m = 0; m <= i; m++) rn[m] = t[i + 1][m];
            }
            double ps[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < i; m++) ps[m & 3] += rj[m] * x[m];
            const double sacc = ((i == lane) ? 1.0 : 0.0) - ((ps[0] + ps[1]) + (ps[2] + ps[3]));
            x[i] = (i < w && lane <= i) ? sacc * rdv[i] : 0.0;
#pragma unroll
            for (int m = 0; m <= i; m++) rj[m] = rn[m];
        }
        if (lane < CT) {
#pragma unroll
            for (int i = 0; i < CT; i++) inv[i][lane] = (lane < w) ? x[i] : 0.0;
        }
    }
}
__global__ __launch_bounds__(256) void k_pipe(const double *A, double *out, long long *cycles, int *fail)
{
    __shared__ double t[CT][CT + 1], inv[CT][CT + 1];
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; t[i][j] = i >= j ? A[i + CT * j] : 0.0; }
    __syncthreads();
    const long long c0 = clock64();
    potrf_pipe(t, inv, CT, CT, fail);
    __syncthreads();
    const long long c1 = clock64();
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; out[e] = t[i][j]; out[CT * CT + e] = inv[i][j]; }
    if (threadIdx.x == 0) *cycles = c1 - c0;
}


// ---- one-wave variant: the inverse advances inside the factor loop (row j of L is already in registers for the dot
// products), pivots from running sums of squares
__device__ __forceinline__ void potrf_fused(double (*t)[CT + 1], double (*inv)[CT + 1], int h, int w, int *fail)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane < CT ? lane : CT - 1;
    __syncthreads();
    if (wv == 0) {
        double lrow[CT], x[CT];
        double dsq = 0.0;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < CT; j++) {
            const bool active = j < w;
            double acc = t[li][j];
            double dj = t[j][j] - bcast_lane(dsq, j);
            double pa[4] = {0.0, 0.0, 0.0, 0.0}, ps[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < j; m++) {
                const double ljm = t[j][m];                          // broadcast, final since step m
                pa[m & 3] += lrow[m] * ljm;
                ps[m & 3] += ljm * x[m];
            }
            acc -= (pa[0] + pa[1]) + (pa[2] + pa[3]);
            bad = bad || (active && !(dj > 0));
            dj = (active && dj > 0) ? dj : 1.0;
            const double rd = rsqrt(dj);
            const double l = (lane == j) ? dj * rd : acc * rd;
            lrow[j] = l;
            dsq += l * l;
            const double sacc = ((j == lane) ? 1.0 : 0.0) - ((ps[0] + ps[1]) + (ps[2] + ps[3]));
            x[j] = (active && lane <= j) ? sacc * rd : 0.0;
            if (active && lane >= j && lane < CT) t[lane][j] = l;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
        }
        if (bad && lane == 0) *fail = 1;
        if (lane < CT) {
#pragma unroll
            for (int i = 0; i < CT; i++) inv[i][lane] = (lane < w) ? x[i] : 0.0;
        }
    }
}
__global__ __launch_bounds__(256) void k_fused(const double *A, double *out, long long *cycles, int *fail)
{
    __shared__ double t[CT][CT + 1], inv[CT][CT + 1];
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; t[i][j] = i >= j ? A[i + CT * j] : 0.0; }
    __syncthreads();
    const long long c0 = clock64();
    potrf_fused(t, inv, CT, CT, fail);
    __syncthreads();
    const long long c1 = clock64();
    for (int e = threadIdx.x; e < CT * CT; e += 256) { const int i = e % CT, j = e / CT; out[e] = t[i][j]; out[CT * CT + e] = inv[i][j]; }
    if (threadIdx.x == 0) *cycles = c1 - c0;
}

int main()
{
    std::vector<double> B(CT * CT), A(CT * CT, 0.0);
    srand(1);
    for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < CT; i++) for (int j = 0; j < CT; j++) { double s = i == j ? CT : 0.0; for (int k = 0; k < CT; k++) s += B[i + CT * k] * B[j + CT * k]; A[i + CT * j] = s; }
    double *dA, *dO; long long *dC; int *dF;
    hipMalloc(&dA, sizeof(double) * CT * CT); hipMalloc(&dO, sizeof(double) * 2 * CT * CT); hipMalloc(&dC, 8); hipMalloc(&dF, 4); hipMemset(dF, 0, 4);
    hipMemcpy(dA, A.data(), sizeof(double) * CT * CT, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_bench, dim3(1), dim3(256), 0, 0, dA, dO, dC, 20, dF);
    hipDeviceSynchronize();
    long long c; std::vector<double> O(2 * CT * CT);
    hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(O.data(), dO, sizeof(double) * 2 * CT * CT, hipMemcpyDeviceToHost);
    double err = 0.0;                                   // || L L' - A ||_max
    for (int i = 0; i < CT; i++) for (int j = 0; j <= i; j++) { double s = 0.0; for (int k = 0; k <= j; k++) s += O[i + CT * k] * O[j + CT * k]; err = fmax(err, fabs(s - A[i + CT * j])); }
    {
        std::vector<double> R(2 * CT * CT);
        hipMemcpy(R.data(), dO, sizeof(double) * 2 * CT * CT, hipMemcpyDeviceToHost);       // reference: tile_potrf_inv
        for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_pipe, dim3(1), dim3(256), 0, 0, dA, dO, dC, dF);
        hipDeviceSynchronize();
        long long cv; hipMemcpy(&cv, dC, 8, hipMemcpyDeviceToHost);
        std::vector<double> Q(2 * CT * CT);
        hipMemcpy(Q.data(), dO, sizeof(double) * 2 * CT * CT, hipMemcpyDeviceToHost);
        double dl = 0.0, di = 0.0;
        for (int i = 0; i < CT; i++) for (int j = 0; j <= i; j++) { dl = fmax(dl, fabs(Q[i + CT * j] - R[i + CT * j])); di = fmax(di, fabs(Q[CT * CT + i + CT * j] - R[CT * CT + i + CT * j])); }
        printf("pipelined: %lld cycles; max |L - L_ref| = %.3e, max |inv - inv_ref| = %.3e\n", cv, dl, di);
        for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_fused, dim3(1), dim3(256), 0, 0, dA, dO, dC, dF);
        hipDeviceSynchronize();
        hipMemcpy(&cv, dC, 8, hipMemcpyDeviceToHost);
        hipMemcpy(Q.data(), dO, sizeof(double) * 2 * CT * CT, hipMemcpyDeviceToHost);
        dl = 0.0; di = 0.0;
        double li_err = 0.0;                              // || L inv - I ||_max
        for (int i = 0; i < CT; i++) for (int j = 0; j <= i; j++) {
            dl = fmax(dl, fabs(Q[i + CT * j] - R[i + CT * j])); di = fmax(di, fabs(Q[CT * CT + i + CT * j] - R[CT * CT + i + CT * j]));
            double s2 = 0.0; for (int k = j; k <= i; k++) s2 += Q[i + CT * k] * Q[CT * CT + k + CT * j];
            li_err = fmax(li_err, fabs(s2 - (i == j ? 1.0 : 0.0)));
        }
        printf("fused one-wave: %lld cycles; max |L - L_ref| = %.3e, max |inv - inv_ref| = %.3e, |L inv - I|max = %.3e\n", cv, dl, di, li_err);
    }
    for (int mode = 0; mode < 4; mode++) {
        for (int it = 0; it < 3; it++) {
            if (mode == 0) hipLaunchKernelGGL(k_var<0>, dim3(1), dim3(256), 0, 0, dA, dO, dC, dF);
            else if (mode == 1) hipLaunchKernelGGL(k_var<1>, dim3(1), dim3(256), 0, 0, dA, dO, dC, dF);
            else if (mode == 2) hipLaunchKernelGGL(k_var<2>, dim3(1), dim3(256), 0, 0, dA, dO, dC, dF);
            else hipLaunchKernelGGL(k_var<3>, dim3(1), dim3(256), 0, 0, dA, dO, dC, dF);
        }
        hipDeviceSynchronize();
        long long cv; hipMemcpy(&cv, dC, 8, hipMemcpyDeviceToHost);
        {
            std::vector<double> Q(2 * CT * CT);
            hipMemcpy(Q.data(), dO, sizeof(double) * 2 * CT * CT, hipMemcpyDeviceToHost);
            double e2 = 0.0;
            for (int i = 0; i < CT; i++) for (int j = 0; j <= i; j++) { double s2 = 0.0; for (int k = 0; k <= j; k++) s2 += Q[i + CT * k] * Q[j + CT * k]; e2 = fmax(e2, fabs(s2 - A[i + CT * j])); }
            printf("variant mode %d (bit0: inverse wave, bit1: incremental pivots): %lld cycles, |LL'-A|max = %.3e\n", mode, cv, e2);
        }
    }
    printf("tile_potrf_inv: %lld cycles (s_memtime ticks) per call, |LL'-A|max = %.3e\n", c, err);
    return 0;
}
