// PMC calibration: coalesced 8-byte-per-lane read + write of a buffer far larger than the Infinity Cache
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_copy8(double *dst, const double *src, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) dst[i] = src[i] + 1.0; }
int main() {
    const size_t n = (size_t)1 << 27;   // 1 GiB per buffer
    double *a, *b;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
    (void)hipMemset(a, 0, n * 8);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_copy8, dim3((n + 255) / 256), dim3(256), 0, 0, b, a, n);
    (void)hipDeviceSynchronize();
    printf("copied %zu bytes read + %zu bytes written per launch\n", n * 8, n * 8);
    return 0;
}
