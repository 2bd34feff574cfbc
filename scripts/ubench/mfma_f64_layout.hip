// v_mfma_f64_16x16x4_f64 on gfx950: which lane supplies / receives which matrix element?  (scripts/ubench, not part of the library)
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout mfma_f64_layout.hip && ./mfma_f64_layout
// Result (MI355X): A[i][k] from lane i + 16 k, B[k][j] from lane j + 16 k, D[4 r + lane / 16][lane % 16] in accumulator register r
// (NOT 4 (lane / 16) + r as for the f32 16x16x4 instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D, long long *cyc)
{
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];      // A[i][k], row-major 16 x 4
    const double b = B[(l / 16) * 16 + l % 16];     // B[k][j], row-major 4 x 16
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * r + l / 16) * 16 + l % 16] = acc[r];
    // issue rate: 64 dependent + 64 x 4 independent MFMAs
    d4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
    long long t0 = wall_clock64();
    for (int i = 0; i < 4096; i++) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); asm volatile("" : "+v"(c0)); }
    long long t1 = wall_clock64();
    for (int i = 0; i < 4096; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    }
    long long t2 = wall_clock64();
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) D[0] = 0;
}
int main()
{
    double hA[64], hB[64], hD[256], *A, *B, *D; long long *cy, hc[2];
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) hA[i * 4 + k] = 1 + i + 0.01 * k;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) hB[k * 16 + j] = (k == 0 ? 1.0 : k == 1 ? 100.0 : k == 2 ? 1e4 : 1e6) * (j + 1);
    hipMalloc(&A, 512); hipMalloc(&B, 512); hipMalloc(&D, 2048); hipMalloc(&cy, 16);
    hipMemcpy(A, hA, 512, hipMemcpyHostToDevice); hipMemcpy(B, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D, cy);
    hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost); hipMemcpy(hc, cy, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        double ref = 0; for (int k = 0; k < 4; k++) ref += hA[i * 4 + k] * hB[k * 16 + j];
        if (fabs(ref - hD[i * 16 + j]) > 1e-9 * fabs(ref)) bad++;
    }
    printf("layout hypothesis: %s (%d mismatches); 4096 dependent MFMAs %.1f ns each, 16384 in four chains %.1f ns each (100 MHz wall clock)\n", bad ? "WRONG" : "confirmed", bad, hc[0] * 10.0 / 4096, hc[1] * 10.0 / 16384);
    return bad != 0;
}
