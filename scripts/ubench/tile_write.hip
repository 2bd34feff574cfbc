// micro-benchmark (scratch tool): write bandwidth of the column-pass store pattern.  A wave owns 64 columns of a column-major
// plane (pitch P doubles) and sweeps the rows bottom to top in blocks of RB rows; per block it stores, for every column, RB*8
// contiguous bytes (128-byte lines: 8 lanes x 16 B per column, 8 columns per instruction).  Variants: rows per block, planes per
// wave, nontemporal stores.  Reports bytes / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int RB, int NT>   // RB = rows per block (16, 32, 64); NT: 0 plain, 1 nontemporal
__global__ __launch_bounds__(256, 2) void k_tiles(double *base, int H, int W, int P, size_t zs, size_t ps, int nplanes_per_wave)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, rp = lane & 7, cg = lane >> 3;
    const int x0 = 62 * blockIdx.x;
    const int nb = (H + RB - 1) / RB;
    for (int b = nb - 1; b >= 0; b--) {
        const int rb = b * RB;
        for (int q = 0; q < nplanes_per_wave; q++) {
            double *dst = base + (size_t)blockIdx.z * zs + (size_t)(w * nplanes_per_wave + q) * ps;
#pragma unroll
            for (int t = 0; t < RB / 16; t++) {
                const unsigned voff = (unsigned)(cg * P + 2 * rp);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int col = x0 + 8 * r + cg;
                    if (col < W && col < x0 + 62) {
                        double *p = dst + ((size_t)(x0 + 8 * r) * P + rb + 16 * t) + voff;
                        const double2 v = make_double2((double)b, (double)r);
                        typedef double v2d __attribute__((ext_vector_type(2)));
                        if (NT) { v2d vv = {v.x, v.y}; __builtin_nontemporal_store(vv, (v2d *)p); } else *(double2 *)p = v;
                    }
                }
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
// the row-pass pattern for comparison: lanes = rows, a wave walks the columns right to left, 512 B per store
__global__ __launch_bounds__(64) void k_rows(double *base, int H, int W, int P, size_t ps)
{
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= H) return;
    double *p = base + (size_t)blockIdx.y * ps + y;
    for (int x = W - 1; x >= 0; x--) p[(size_t)x * P] = (double)x;
}
int main(int argc, char **argv)
{
    const int H = 370, W = 1226, P = 384, S = argc > 1 ? atoi(argv[1]) : 32, NPL = 6;
    const size_t ps = (size_t)P * W + 2048, zs = ps * NPL;
    double *d; hipMalloc(&d, zs * S * 8 + (1 << 20)); hipMemset(d, 0, zs * S * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int ntx = (W + 61) / 62;
    const double bytes = (double)S * NPL * H * W * 8;
    auto timeit = [&](const char *name, auto launch) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-46s %8.1f us  %5.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
    };
    timeit("tiles RB=32, 4 waves x 1.5 planes (6 planes)", [&] { hipLaunchKernelGGL((k_tiles<32, 0>), dim3(ntx, 1, S), dim3(256), 0, 0, d, H, W, P, zs, ps * 3 / 2, 1); hipLaunchKernelGGL((k_tiles<32, 0>), dim3(ntx, 1, S), dim3(128), 0, 0, d + 4 * ps, H, W, P, zs, ps, 1); });
    timeit("tiles RB=32, 3 waves x 2 planes", [&] { hipLaunchKernelGGL((k_tiles<32, 0>), dim3(ntx, 1, S), dim3(192), 0, 0, d, H, W, P, zs, ps, 2); });
    timeit("tiles RB=16, 3 waves x 2 planes", [&] { hipLaunchKernelGGL((k_tiles<16, 0>), dim3(ntx, 1, S), dim3(192), 0, 0, d, H, W, P, zs, ps, 2); });
    timeit("tiles RB=64, 3 waves x 2 planes", [&] { hipLaunchKernelGGL((k_tiles<64, 0>), dim3(ntx, 1, S), dim3(192), 0, 0, d, H, W, P, zs, ps, 2); });
    timeit("tiles RB=32 nontemporal, 3 waves x 2 planes", [&] { hipLaunchKernelGGL((k_tiles<32, 1>), dim3(ntx, 1, S), dim3(192), 0, 0, d, H, W, P, zs, ps, 2); });
    timeit("tiles RB=64 nontemporal, 3 waves x 2 planes", [&] { hipLaunchKernelGGL((k_tiles<64, 1>), dim3(ntx, 1, S), dim3(192), 0, 0, d, H, W, P, zs, ps, 2); });
    timeit("rows pattern (512 B per store), 6 planes", [&] { for (int z = 0; z < S; z++) hipLaunchKernelGGL(k_rows, dim3((H + 63) / 64, NPL), dim3(64), 0, 0, d + z * zs, H, W, P, ps); });
    return 0;
}
