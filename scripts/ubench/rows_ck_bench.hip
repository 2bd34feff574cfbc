// scratch harness: the library's dim-2 kernels alone on level-0 shaped planes (S images x 4 planes), so that variants of
// k_iir_rows_ck can be timed without the rest of the build.  Build: see scripts/ubench/Makefile-less one-liner in the file header:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../slam.jl_amd/csrc -o rows_ck_bench rows_ck_bench.hip ../../slam.jl_amd/csrc/{ctx,detect,lk,ba,brief,triangulate,pose,fivepoint,comm,kpset}.o -ldl
#define CF4_EXP 1
#include "../../slam.jl_amd/csrc/pyramid.hip"
#include <cstdio>
__global__ void k_rand(double *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = (h & 0xFFFFF) * (1.0 / 1048576.0); } }
int main(int argc, char **argv)
{
    const int H = 370, W = 1226, P = 384, S = argc > 1 ? atoi(argv[1]) : 32;
    const size_t zs = (size_t)P * W;
    double *pl[4];
    for (int k = 0; k < 4; k++) { hipMalloc(&pl[k], zs * S * 8); hipLaunchKernelGGL(k_rand, dim3((zs * S + 255) / 256), dim3(256), 0, 0, pl[k], zs * S); }
    double *ck; const size_t nlines = (size_t)S * 4 * ((H + 63) / 64) * 64;
    hipMalloc(&ck, (size_t)((W + CK_B - 1) / CK_B + 1) * 3 * nlines * 8);
    PlaneSet ps{}; ps.n = 4; ps.zs = zs;
    for (int k = 0; k < 4; k++) { ps.p[k] = pl[k]; ps.coef[k] = k ? 1 : 0; ps.fill0[k] = k ? 1 : 0; }
    IIRPair cf; cf.c[0] = slam_iir_coef(1.0); cf.c[1] = slam_iir_coef(4.0);
    RowResize rz{}; 
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int var = 0; var < 2; var++) {
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_rand, dim3((zs * S + 255) / 256), dim3(256), 0, 0, pl[k], zs * S);
            hipEventRecord(a);
            if (var == 0) hipLaunchKernelGGL(k_iir_rows_ck, lines_grid(H, 4, S), dim3(LINE_THREADS), 0, 0, ps, H, W, P, cf, ck, rz);
            else hipLaunchKernelGGL(k_iir_rows, lines_grid(H, 4, S), dim3(LINE_THREADS), 0, 0, ps, H, W, P, cf);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%s S=%d: %.1f us\n", var == 0 ? "k_iir_rows_ck" : "k_iir_rows", S, best * 1e3);
    }
    // ---- k_cols_fused, level-0 shape, layer from a pitched plane; parts switched off through cf4_exp ----
    {
        double *L, *T, *Iy, *Ix, *Q[3];
        hipMalloc(&L, zs * S * 8 + 65536); hipLaunchKernelGGL(k_rand, dim3((zs * S + 255) / 256), dim3(256), 0, 0, L, zs * S);
        T = pl[0]; Q[0] = pl[1]; Q[1] = pl[2]; Q[2] = pl[3];
        hipMalloc(&Iy, zs * S * 8 + 65536); hipMalloc(&Ix, zs * S * 8 + 65536);
        ColsFusedArgs A{}; A.L = L; A.T = T; A.Iy = Iy; A.Ix = Ix; A.Qyy = Q[0]; A.Qxx = Q[1]; A.Qyx = Q[2]; A.H = H; A.W = W; A.P = P; A.zs = zs; A.src_kind = 0; A.srctab = nullptr;
        const int ntx = (W + CF4_COLS - 1) / CF4_COLS;
        double *ck2; hipMalloc(&ck2, (size_t)((H >> 5) + 2) * 3 * (size_t)S * 4 * ntx * 64 * 8);
        const int masks[] = {0, 32, 1, 1 | 32, 2 | 4, 2 | 4 | 32, 31, 63};
        for (int Sv : {S, 25}) for (int m : masks) {
            hipMemcpyToSymbol(HIP_SYMBOL(cf4_exp), &m, sizeof(int));
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL((k_cols_fused<false, false>), dim3(ntx, 1, Sv), dim3(256), 0, 0, A, cf, ck2);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("k_cols_fused S=%d off[%s%s%s%s%s]: %.1f us\n", Sv, m & 1 ? " stores" : "", m & 2 ? " chains" : "", m & 4 ? " scharr" : "", m & 8 ? " ckloads" : "", m & 16 ? " layerloads" : "", best * 1e3); if (m & 32) printf("   (^ without the workgroup barriers: results invalid, timing only)\n");
        }
    }
    return 0;
}
