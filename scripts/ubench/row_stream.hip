// micro-benchmark (scratch tool): the memory pattern of the dim-2 recurrence passes -- lanes = consecutive rows, a wave walks the
// line sample by sample (stride = pitch), forward read-only sweep then backward read + write sweep -- without / with the
// dependent IIR chain, 8 or 16 bytes per lane, block size B and D blocks of prefetch.  Reports moved bytes / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int LB> struct V;
template <> struct V<1> { typedef double T; static __device__ double sum(double v) { return v; } static __device__ double mk(double a, double) { return a; } };
template <> struct V<2> { typedef double2 T; };
template <int LB, int B, int D, bool CHAIN, bool PASSA>
__global__ __launch_bounds__(64) void k(double *base, int H, int W, int P, size_t zs, double a1, double a2, double a3)
{
    const int y = (blockIdx.x * 64 + threadIdx.x) * LB;
    if (y >= H) return;
    double *p = base + (size_t)blockIdx.y * zs + y;
    const long s = P;
    double buf[D + 1][B][LB];
    double w1[LB], w2[LB], w3[LB];
    for (int r = 0; r < LB; r++) { w1[r] = 0.1 * r; w2[r] = 0.2; w3[r] = 0.3; }
    const int nb = W / B;
    auto load = [&](int j, int slot) {
        const double *q = p + (long)j * B * s;
#pragma unroll
        for (int e = 0; e < B; e++) {
            if (LB == 1) buf[slot][e][0] = q[(long)e * s];
            else { const double2 v = *(const double2 *)(q + (long)e * s); buf[slot][e][0] = v.x; buf[slot][e][LB - 1] = v.y; }
        }
    };
    if (PASSA) {
#pragma unroll
        for (int d = 0; d < D; d++) load(d, d);
        for (int j0 = 0; j0 < nb; j0 += D + 1) {
#pragma unroll
            for (int u = 0; u < D + 1; u++) {
                const int j = j0 + u;
                if (j < nb) {
                    if (j + D < nb) load(j + D, (u + D) % (D + 1));
#pragma unroll
                    for (int e = 0; e < B; e++)
#pragma unroll
                        for (int r = 0; r < LB; r++) {
                            if (CHAIN) { const double t = ((buf[u][e][r] + a1 * w1[r]) + a2 * w2[r]) + a3 * w3[r]; w3[r] = w2[r]; w2[r] = w1[r]; w1[r] = t; }
                            else w1[r] += buf[u][e][r];
                        }
                }
            }
        }
    }
    // backward: read + write
#pragma unroll
    for (int d = 0; d < D; d++) load(nb - 1 - d, d);
    for (int j0 = 0; j0 < nb; j0 += D + 1) {
#pragma unroll
        for (int u = 0; u < D + 1; u++) {
            const int j = j0 + u;                    // j-th block from the right
            if (j < nb) {
                if (j + D < nb) load(nb - 1 - (j + D), (u + D) % (D + 1));
                double *q = p + (long)(nb - 1 - j) * B * s;
                if (CHAIN) {                          // forward recompute, then backward
#pragma unroll
                    for (int e = 0; e < B; e++)
#pragma unroll
                        for (int r = 0; r < LB; r++) { const double t = ((buf[u][e][r] + a1 * w1[r]) + a2 * w2[r]) + a3 * w3[r]; w3[r] = w2[r]; w2[r] = w1[r]; w1[r] = t; buf[u][e][r] = t; }
                }
#pragma unroll
                for (int e = B - 1; e >= 0; e--) {
#pragma unroll
                    for (int r = 0; r < LB; r++) {
                        if (CHAIN) { const double t = ((buf[u][e][r] + a1 * w1[r]) + a2 * w2[r]) + a3 * w3[r]; w3[r] = w2[r]; w2[r] = w1[r]; w1[r] = t; buf[u][e][r] = t; }
                        else buf[u][e][r] += w1[r];
                    }
                }
#pragma unroll
                for (int e = 0; e < B; e++) {
                    if (LB == 1) q[(long)e * s] = buf[u][e][0];
                    else *(double2 *)(q + (long)e * s) = make_double2(buf[u][e][0], buf[u][e][LB - 1]);
                }
            }
        }
    }
}
template <int LB, int B, int D, bool CHAIN, bool PASSA>
static void run(const char *name, double *d, int H, int W, int P, int NP)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t zs = (size_t)P * W;
    dim3 grid((H + 64 * LB - 1) / (64 * LB), NP);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<LB, B, D, CHAIN, PASSA>), grid, dim3(64), 0, 0, d, H, W, P, zs, 1e-3, -2e-3, 1e-3);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double bytes = (double)NP * H * (W / B * B) * 8 * (PASSA ? 3 : 2);
    printf("%-34s waves %5d  %8.1f us  %6.2f TB/s moved\n", name, grid.x * grid.y, best * 1e3, bytes / best / 1e9);
}
int main(int argc, char **argv)
{
    const int H = 370, W = 1226, P = 384, NP = argc > 1 ? atoi(argv[1]) : 128;
    double *d; hipMalloc(&d, (size_t)P * W * NP * 8); hipMemset(d, 0, (size_t)P * W * NP * 8);
    run<1, 32, 1, false, true>("8B/lane B32 D1 copy A+B", d, H, W, P, NP);
    run<1, 32, 1, false, false>("8B/lane B32 D1 copy B only", d, H, W, P, NP);
    run<1, 16, 3, false, true>("8B/lane B16 D3 copy A+B", d, H, W, P, NP);
    run<1, 32, 2, false, true>("8B/lane B32 D2 copy A+B", d, H, W, P, NP);
    run<2, 16, 1, false, true>("16B/lane B16 D1 copy A+B", d, H, W, P, NP);
    run<2, 16, 2, false, true>("16B/lane B16 D2 copy A+B", d, H, W, P, NP);
    run<2, 16, 3, false, true>("16B/lane B16 D3 copy A+B", d, H, W, P, NP);
    run<2, 16, 2, false, false>("16B/lane B16 D2 copy B only", d, H, W, P, NP);
    run<1, 32, 1, true, true>("8B/lane B32 D1 chain A+B", d, H, W, P, NP);
    run<1, 32, 2, true, true>("8B/lane B32 D2 chain A+B", d, H, W, P, NP);
    run<1, 16, 3, true, true>("8B/lane B16 D3 chain A+B", d, H, W, P, NP);
    run<2, 16, 1, true, true>("16B/lane B16 D1 chain A+B", d, H, W, P, NP);
    run<2, 16, 2, true, true>("16B/lane B16 D2 chain A+B", d, H, W, P, NP);
    run<2, 16, 3, true, true>("16B/lane B16 D3 chain A+B", d, H, W, P, NP);
    run<2, 16, 2, true, false>("16B/lane B16 D2 chain B only", d, H, W, P, NP);
    run<1, 32, 1, true, false>("8B/lane B32 D1 chain B only", d, H, W, P, NP);
    return 0;
}
