// micro-benchmark (scratch tool): the memory pattern of k_rows_tol -- a workgroup owns R rows of a column-major plane, thread = (row(s), column
// segment of SL samples): every sample is loaded once and stored once.  What rate does the pattern itself reach, for R = 8 / 16 / 32 rows,
// 8 or 16 bytes per lane, and for which column pitch?   hipcc -O3 --offload-arch=gfx950 seg_rows.hip -o seg_rows && ./seg_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int R, int NS, int SL, int LB, bool NT>
__global__ __launch_bounds__(R / LB * NS) void k(const double *src, double *dst, int H, int W, int P, size_t zs)
{
    constexpr int LPW = R / LB;
    const int t = threadIdx.x, l = t % LPW, g = t / LPW;
    const int y = (blockIdx.x * LPW + l) * LB;
    const size_t z = (size_t)blockIdx.y * zs;
    const int nseg = (W + SL - 1) / SL;
    if (g >= nseg) return;
    const int yc = y < H ? y : (H - LB);
    int b = g * SL; if (b > W - SL) b = W - SL;
    const char *pb = (const char *)(src + z); char *wb = (char *)(dst + z);
    double x[SL][LB];
#pragma unroll
    for (int j = 0; j < SL; j++) {
        const unsigned off = ((unsigned)(b + j) * (unsigned)P + (unsigned)yc) * 8u;
        if (LB == 1) x[j][0] = NT ? __builtin_nontemporal_load((const double *)(pb + off)) : *(const double *)(pb + off);
        else { const double2 v = *(const double2 *)(pb + off); x[j][0] = v.x; x[j][LB - 1] = v.y; }
    }
    if (y >= H) return;
#pragma unroll
    for (int j = 0; j < SL; j++) {
        const unsigned off = ((unsigned)(b + j) * (unsigned)P + (unsigned)yc) * 8u;
        if (LB == 1) { if (NT) __builtin_nontemporal_store(x[j][0] + 1.0, (double *)(wb + off)); else *(double *)(wb + off) = x[j][0] + 1.0; }
        else *(double2 *)(wb + off) = make_double2(x[j][0] + 1.0, x[j][LB - 1] + 1.0);
    }
}
template <int R, int NS, int SL, int LB, bool NT>
static void run(const char *name, double *a, double *c, int H, int W, int P, size_t zs, int S)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((H + R - 1) / R, S), block(R / LB * NS);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<R, NS, SL, LB, NT>), grid, block, 0, 0, a, c, H, W, P, zs);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<R, NS, SL, LB, NT>), grid, block, 0, 0, a, c, H, W, P, zs);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = 2.0 * S * (double)H * W * 8;
    printf("%-34s P=%4d  %8.1f us  %7.1f GB/s (R+W of H*W)\n", name, P, best * 1e3, bytes / (best * 1e-3) / 1e9);
}
int main(int argc, char **argv)
{
    const int H = 370, W = 1226, S = 512;                       // 4 planes x 128 images
    for (int P : {384, 400, 392, 448, 370 + 6}) {
        const int Pp = (P + 1) & ~1;
        const size_t zs = (size_t)Pp * W + 4096;
        double *a, *c;
        hipMalloc(&a, zs * S * 8); hipMalloc(&c, zs * S * 8);
        hipMemset(a, 0, zs * S * 8); hipMemset(c, 0, zs * S * 8);
        run<16, 64, 20, 1, true>("R16 NS64 SL20 8B nt", a, c, H, W, Pp, zs, S);
        run<16, 64, 20, 1, false>("R16 NS64 SL20 8B", a, c, H, W, Pp, zs, S);
        run<8, 64, 20, 1, true>("R8 NS64 SL20 8B nt", a, c, H, W, Pp, zs, S);
        run<32, 32, 40, 1, true>("R32 NS32 SL40 8B nt", a, c, H, W, Pp, zs, S);
        run<32, 64, 20, 2, false>("R32 NS64 SL20 16B (2 rows/lane)", a, c, H, W, Pp, zs, S);
        run<16, 64, 20, 2, false>("R16 NS64 SL20 16B (2 rows/lane)", a, c, H, W, Pp, zs, S);
        run<64, 32, 40, 2, false>("R64 NS32 SL40 16B (2 rows/lane)", a, c, H, W, Pp, zs, S);
        run<64, 16, 80, 1, true>("R64 NS16 SL80 8B nt", a, c, H, W, Pp, zs, S);
        hipFree(a); hipFree(c);
    }
    return 0;
}
