// micro-benchmark: dependent f64 chain latency on one wave (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_add(double *o, double a, int n) { double x = o[threadIdx.x]; for (int i = 0; i < n; i++) { x = x + a; x = x + a; x = x + a; x = x + a; } o[threadIdx.x] = x; }
__global__ void k_iir(double *o, double a1, double a2, double a3, int n) {
    double w1 = o[threadIdx.x], w2 = w1 * 0.5, w3 = w1 * 0.25, x = 0.125;
    for (int i = 0; i < n; i++) { double t = ((x + a1 * w1) + a2 * w2) + a3 * w3; w3 = w2; w2 = w1; w1 = t; }
    o[threadIdx.x] = w1;
}
__global__ void k_empty(double *o) { if (threadIdx.x == 1000) o[0] = 1; }
int main() {
    double *d; hipMalloc(&d, 8 * 256); hipMemset(d, 0, 8 * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        const int n = 100000;
        hipEventRecord(a); hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, 0, d, 1e-9, n); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); printf("dep add: %.2f ns/op (%.1f cycles @2.38GHz)\n", ms * 1e6 / (4.0 * n), ms * 1e6 / (4.0 * n) * 2.38);
        hipEventRecord(a); hipLaunchKernelGGL(k_iir, dim3(1), dim3(64), 0, 0, d, 0.5, -0.2, 0.05, n); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); printf("iir step: %.2f ns/step (%.1f cycles)\n", ms * 1e6 / n, ms * 1e6 / n * 2.38);
        hipEventRecord(a); for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, d); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); printf("empty kernel chain: %.2f us/launch\n", ms * 1e3 / 100);
    }
    return 0;
}
