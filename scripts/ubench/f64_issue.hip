// micro-benchmark (scratch tool): issue rate of independent f64 operations on ONE wave (cycles per instruction by s_memtime),
// and of v_readlane_b32 / v_ldexp_f64 / v_rsq_f64 / rsqrt().   hipcc -O3 --offload-arch=gfx950 scripts/ubench/f64_issue.hip -o scripts/ubench/f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH> __global__ void k_fma(double *o, double a, double b, int n, long long *cyc)
{
    double x[NCH];
    for (int c = 0; c < NCH; c++) x[c] = o[threadIdx.x] + c;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < NCH; c++) x[c] = fma(x[c], a, b);
    }
    const long long t1 = clock64();
    double s = 0; for (int c = 0; c < NCH; c++) s += x[c];
    o[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH> __global__ void k_mul(double *o, double a, int n, long long *cyc)
{
    double x[NCH];
    for (int c = 0; c < NCH; c++) x[c] = o[threadIdx.x] + c;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < NCH; c++) x[c] = x[c] * a;
    }
    const long long t1 = clock64();
    double s = 0; for (int c = 0; c < NCH; c++) s += x[c];
    o[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH> __global__ void k_rsq(double *o, int n, long long *cyc, int lib)
{
    double x[NCH];
    for (int c = 0; c < NCH; c++) x[c] = o[threadIdx.x] + c + 2.0;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < NCH; c++) x[c] = lib ? rsqrt(x[c]) + 1.5 : __builtin_amdgcn_rsq(x[c]) + 1.5;
    }
    const long long t1 = clock64();
    double s = 0; for (int c = 0; c < NCH; c++) s += x[c];
    o[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_readlane(double *o, int n, long long *cyc)
{
    double x = o[threadIdx.x], acc = 0;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int l = 0; l < 16; l++) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
            acc = fma(__hiloint2double(hi, lo), 1.0000001, acc);
        }
        x = acc;
    }
    const long long t1 = clock64();
    o[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// ---- accuracy of v_rsq_f64 and of one / two Newton steps on it (max relative error over n inputs spread over [2^-40, 2^40]) ----
__global__ void k_rsq_acc(double *out, int n)
{
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const double x = ldexp(1.0 + (double)((i * 2654435761u) & 0xfffff) / 1048576.0, (i % 81) - 40);
        const double ref = 1.0 / sqrt(x);
        const double y0 = __builtin_amdgcn_rsq(x);
        const double r0 = fma(-(x * y0), y0, 1.0), y1 = fma(y0 * 0.5, r0, y0);
        const double r1 = fma(-(x * y1), y1, 1.0), y2 = fma(y1 * 0.5, r1, y1);
        e0 = fmax(e0, fabs(y0 - ref) / ref); e1 = fmax(e1, fabs(y1 - ref) / ref); e2 = fmax(e2, fabs(y2 - ref) / ref);
    }
    for (int o = 32; o; o >>= 1) { e0 = fmax(e0, __shfl_xor(e0, o)); e1 = fmax(e1, __shfl_xor(e1, o)); e2 = fmax(e2, __shfl_xor(e2, o)); }
    if (threadIdx.x == 0) { out[0] = e0; out[1] = e1; out[2] = e2; }
}
int main()
{
    { double *q; hipMalloc(&q, 64); hipLaunchKernelGGL(k_rsq_acc, dim3(1), dim3(64), 0, 0, q, 1 << 20); double h[3]; hipMemcpy(h, q, 24, hipMemcpyDeviceToHost);
      printf("v_rsq_f64 max relative error %.3g; after one Newton step %.3g; after two %.3g\n", h[0], h[1], h[2]); }
    double *d; long long *c; hipMalloc(&d, 8 * 256); hipMemset(d, 0, 8 * 256); hipHostMalloc(&c, 64);
    const int n = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
#define RUN(name, per, ...) for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); hipLaunchKernelGGL(__VA_ARGS__); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); \
    if (rep) printf("%-44s %.2f s_memtime ticks = %.2f ns per instruction (%.2f ticks per ns)\n", name, (double)c[0] / ((double)n * (per)), ms * 1e6 / ((double)n * (per)), c[0] / (ms * 1e6)); }
    RUN("v_fma_f64, 1 dependent chain", 1, k_fma<1>, dim3(1), dim3(64), 0, 0, d, 0.999, 1e-3, n, c)
    RUN("v_fma_f64, 2 chains", 2, k_fma<2>, dim3(1), dim3(64), 0, 0, d, 0.999, 1e-3, n, c)
    RUN("v_fma_f64, 4 chains", 4, k_fma<4>, dim3(1), dim3(64), 0, 0, d, 0.999, 1e-3, n, c)
    RUN("v_fma_f64, 8 chains", 8, k_fma<8>, dim3(1), dim3(64), 0, 0, d, 0.999, 1e-3, n, c)
    RUN("v_fma_f64, 8 chains, 2 waves on the CU", 8, k_fma<8>, dim3(1), dim3(128), 0, 0, d, 0.999, 1e-3, n, c)
    RUN("v_fma_f64, 8 chains, 8 waves (2 per SIMD)", 8, k_fma<8>, dim3(1), dim3(512), 0, 0, d, 0.999, 1e-3, n, c)
    RUN("v_mul_f64, 1 chain", 1, k_mul<1>, dim3(1), dim3(64), 0, 0, d, 0.999, n, c)
    RUN("v_mul_f64, 8 chains", 8, k_mul<8>, dim3(1), dim3(64), 0, 0, d, 0.999, n, c)
    RUN("v_rsq_f64 + add, 1 chain (2 instr)", 1, k_rsq<1>, dim3(1), dim3(64), 0, 0, d, n, c, 0)
    RUN("v_rsq_f64 + add, 6 chains (per pair)", 6, k_rsq<6>, dim3(1), dim3(64), 0, 0, d, n, c, 0)
    RUN("rsqrt() + add, 1 chain (per call)", 1, k_rsq<1>, dim3(1), dim3(64), 0, 0, d, n, c, 1)
    RUN("rsqrt() + add, 6 chains (per call)", 6, k_rsq<6>, dim3(1), dim3(64), 0, 0, d, n, c, 1)
    RUN("2 v_readlane_b32 + fma (per triple)", 16, k_readlane, dim3(1), dim3(64), 0, 0, d, n, c)
    return 0;
}
