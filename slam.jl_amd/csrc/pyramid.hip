// pyramid.hip -- device-resident LKPyramid: Gaussian pyramid, Scharr gradients and
// the sigma=4 smoothed gradient products stored as integral images.
//
// Replaces LKPyramid(...) / update! / copy! / deepcopy of the reference
// (src/optical_flow/pyramid.jl:16-137, src/optical_flow/lucas_kanade.jl:102-138).
//
// Parity design: every recurrence (Young-van Vliet IIR Gaussian with
// Triggs-Sdika boundaries, cumulative sums) runs sequentially along its line in
// the reference's operation order, one lane per line, so all six planes are
// bit-identical to the CPU oracle (-ffp-contract=off).  Lines are independent:
// lanes map to lines, and a launch carries all planes that are ready.
//
// Layout in HBM: one allocation per pyramid (or per batch of S pyramids), 6 planes x
// sum_l(P_l*W_l) doubles, each plane column-major H_l x W_l (y fastest) like the Julia
// arrays but with a column pitch P_l = H_l rounded up to 16 doubles (every column starts
// on a 128-byte line), plus one blur scratch plane.  Row passes (recurrence along x) are
// naturally coalesced (lanes = consecutive y); column passes walk y inside a lane and move
// whole aligned lines through an LDS transpose.  Batched launches take the image index
// from blockIdx.z; bandwidth-bound ones switch to the checkpointed IIR kernels
// (k_iir_*_ck) and the one-pass integral image (k_cum_fused), all bit-identical.
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <tuple>
#include <utility>

#define LINE_THREADS 64

struct PlaneSet {
    double *p[4];          // planes processed by one launch
    const double *nrm[4];  // optional NA() normaliser (divide at the end), or nullptr
    int coef[4];           // which IIRCoef (0 = pyramid sigma, 1 = sigma 4)
    int fill0[4];          // border: 0 replicate, 1 Fill(0)
    const double *sq[4];   // k_iir_cols_ck only: if set, the plane's input is sq[k] squared (Iyy = Iy * Iy, Ixx = Ix * Ix formed on the fly)
    int n;
    size_t zs;             // batch: image blockIdx.z lives zs doubles after image 0 (0 for a single pyramid)
};
// batched launches: grid.z = image index; every plane pointer of image z is image 0's + z * zs
// Plane attributes are picked with wave-uniform selects (never by indexing the kernel argument with
// blockIdx.y, which would move the struct to scratch and the plane pointer into vector registers).
#define PS_PICK(ps, f, pl) ((pl) == 0 ? (ps).f[0] : (pl) == 1 ? (ps).f[1] : (pl) == 2 ? (ps).f[2] : (ps).f[3])
__device__ __forceinline__ double *ps_plane(const PlaneSet &ps, int pl) { return PS_PICK(ps, p, pl) + (size_t)blockIdx.z * ps.zs; }
__device__ __forceinline__ const double *ps_nrm(const PlaneSet &ps, int pl) { return PS_PICK(ps, nrm, pl); }
__device__ __forceinline__ int ps_coef(const PlaneSet &ps, int pl) { return PS_PICK(ps, coef, pl); }
__device__ __forceinline__ bool ps_fill0(const PlaneSet &ps, int pl) { return PS_PICK(ps, fill0, pl) != 0; }
__device__ __forceinline__ const double *ps_sq(const PlaneSet &ps, int pl) { const double *q = PS_PICK(ps, sq, pl); return q ? q + (size_t)blockIdx.z * ps.zs : nullptr; }

struct IIRPair { IIRCoef c[2]; };

// Software-pipelined sequential sweep over `count` elements of a strided line:
// element j lives at base[j*step] (step may be negative), out[j] = f(in[j]) with
// f carrying the recurrence state.  The recurrence itself is inherently
// sequential (its exact operation order is what makes the planes bit-identical
// to the oracle); the memory traffic is not: inputs are fetched NB*C elements
// ahead into a register ring so the dependent chain never waits on HBM/L2
// latency, and results are stored behind it.
template <int C, int NB, class F>
__device__ __forceinline__ void stream_line(const double *src, double *dst, long step, int count, F f)
{
    double buf[NB][C];
    const int nfull = count / C;
    const long cstep = (long)C * step;
    const double *lp = src;            // next chunk to load
    double *sp = dst;                  // next chunk to store
    int loaded = 0;
#pragma unroll
    for (int b = 0; b < NB; b++)
        if (b < nfull) {
#pragma unroll
            for (int c = 0; c < C; c++) buf[b][c] = lp[c * step];
            lp += cstep; loaded++;
        }
    int done = 0;
    while (done + NB <= nfull) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int c = 0; c < C; c++) buf[b][c] = f(buf[b][c]);
#pragma unroll
            for (int c = 0; c < C; c++) sp[c * step] = buf[b][c];
            sp += cstep;
            if (loaded < nfull) {
#pragma unroll
                for (int c = 0; c < C; c++) buf[b][c] = lp[c * step];
                lp += cstep; loaded++;
            }
        }
        done += NB;
    }
#pragma unroll
    for (int b = 0; b < NB; b++)
        if (done + b < nfull) {
#pragma unroll
            for (int c = 0; c < C; c++) buf[b][c] = f(buf[b][c]);
#pragma unroll
            for (int c = 0; c < C; c++) sp[c * step] = buf[b][c];
            sp += cstep;
        }
    for (int j = nfull * C; j < count; j++) dst[(long)j * step] = f(src[(long)j * step]);
}

// read-only variant (segment pre-pass of the tolerance-mode kernels): f consumes, nothing is stored
template <int C, int NB, class F>
__device__ __forceinline__ void stream_read(const double *src, long step, int count, F f)
{
    double buf[NB][C];
    const int nfull = count / C;
    const long cstep = (long)C * step;
    const double *lp = src;
    int loaded = 0;
#pragma unroll
    for (int b = 0; b < NB; b++)
        if (b < nfull) {
#pragma unroll
            for (int c = 0; c < C; c++) buf[b][c] = lp[c * step];
            lp += cstep; loaded++;
        }
    int done = 0;
    while (done + NB <= nfull) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int c = 0; c < C; c++) f(buf[b][c]);
            if (loaded < nfull) {
#pragma unroll
                for (int c = 0; c < C; c++) buf[b][c] = lp[c * step];
                lp += cstep; loaded++;
            }
        }
        done += NB;
    }
#pragma unroll
    for (int b = 0; b < NB; b++)
        if (done + b < nfull) {
#pragma unroll
            for (int c = 0; c < C; c++) f(buf[b][c]);
        }
    for (int j = nfull * C; j < count; j++) f(src[(long)j * step]);
}

// ---- line I/O policies ------------------------------------------------------
// RowIO: the lane's line is strided (recurrence along x, lanes = consecutive y):
// every access of the wave is one coalesced 512-byte segment.
struct RowIO {
    const double *src; double *dst; long s;
    __device__ __forceinline__ double ld_src(int i) const { return src[(long)i * s]; }
    __device__ __forceinline__ double ld_dst(int i) const { return dst[(long)i * s]; }
    __device__ __forceinline__ void st(int i, double v) const { dst[(long)i * s] = v; }
    // sweep `count` elements starting at i0 in direction DIR (+1/-1); from_dst: read dst instead of src
    template <int DIR, class F> __device__ __forceinline__ void sweep(int i0, int count, bool from_dst, F f) const
    {
        stream_line<8, 4>((from_dst ? dst : src) + (long)i0 * s, dst + (long)i0 * s, DIR * s, count, f);
    }
    __device__ __forceinline__ void fence() const {}
};

// ColIO: the lane's line is contiguous (recurrence along y, lanes = 64
// consecutive columns).  Direct access would touch 64 cache lines per
// wave-instruction; instead 16-row x 64-column tiles move between HBM and
// registers as whole 128-byte lines (8 lanes x 16 bytes per column; the column
// pitch is a multiple of 16 doubles and tiles start on multiples of 16 rows, so
// every global access is a full, aligned line) and are transposed through one
// 9 KB LDS tile, so each lane ends up with the 16 consecutive samples of its own
// column.  A ring of NB tiles is prefetched (NB = 3 for a single image, where the wave is alone on
// its SIMD and the dependent chain must never wait for HBM; NB = 2 for batched launches, which keeps
// the kernel under 256 VGPRs so that two waves share a SIMD).  Rows of the first / last tile
// that lie outside the swept range are carried through untouched and not stored.
#define COL_LS 18                       // LDS column stride in doubles: 16 + 2 keeps ds_*_b128 aligned and conflict-free
typedef double v2d __attribute__((ext_vector_type(2)));
template <int COL_NB> struct ColIO {
    const double *src; double *dst;     // plane bases
    int H, W, P, x0;                    // rows, columns, column pitch, first column of this wave
    int slo, shi;                       // columns this wave may store: [slo, shi] (whole image: 0, W - 1; the fused column kernel: its 62 own columns)
    double *lds;                        // 64 x COL_LS doubles
    bool nt = false;                    // full tiles leave as nontemporal 16-byte stores: the bandwidth-bound batch kernels only (streams far larger than the
                                        // caches; for a single image the next kernel finds the plane in L2 / MALL and nontemporal stores cost 39 % of the frame rate)
    __device__ __forceinline__ int xown() const { int x = x0 + (int)(threadIdx.x & 63); return x < W ? x : W - 1; }
    __device__ __forceinline__ bool valid() const { const int c = x0 + (int)(threadIdx.x & 63); return c >= slo && c <= shi; }
    __device__ __forceinline__ double ld_src(int i) const { return src[(size_t)i + (size_t)xown() * P]; }
    __device__ __forceinline__ double ld_dst(int i) const { return dst[(size_t)i + (size_t)xown() * P]; }
    __device__ __forceinline__ void st(int i, double v) const { if (valid()) dst[(size_t)i + (size_t)xown() * P] = v; }
    __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

    // tile registers: t[2r], t[2r+1] = rows rb + 2*rp, +1 of column x0 + 8r + cg   (rp = lane & 7, cg = lane >> 3)
    // Addresses are (wave-uniform base of column group r) + (32-bit lane offset shared by all r): the
    // loads/stores take the SGPR-base form and no per-column address registers.  Loads of the last
    // workgroup may run past column W-1 into the next plane / the allocation's tail padding (values of
    // lanes without a column are never stored).
    __device__ __forceinline__ void tile_load(const double *base, int rb, double t[16]) const
    {
        const int lane = threadIdx.x & 63, rp = lane & 7, cg = lane >> 3;
        const unsigned voff = (unsigned)(cg * P + 2 * rp);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double *cb = base + ((size_t)(x0 + 8 * r) * P + rb);
            const double2 q = *(const double2 *)(cb + voff);
            t[2 * r] = q.x; t[2 * r + 1] = q.y;
        }
    }
    __device__ __forceinline__ void tile_store(int rb, const double t[16], int lo, int hi, bool partial) const
    {
        const int lane = threadIdx.x & 63, rp = lane & 7, cg = lane >> 3;
        const int row = rb + 2 * rp;
        const unsigned voff = (unsigned)(cg * P + 2 * rp);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int col = x0 + 8 * r + cg;
            if (col >= slo && col <= shi) {
                double *q = dst + ((size_t)(x0 + 8 * r) * P + rb) + voff;
                if (!partial && nt) { const v2d v = {t[2 * r], t[2 * r + 1]}; __builtin_nontemporal_store(v, (v2d *)q); }   // one 16-byte store (a plain one is split and merged with the partial path below: 8-byte stores)
                else if (!partial) *(double2 *)q = make_double2(t[2 * r], t[2 * r + 1]);
                else {
                    if (row >= lo && row <= hi) q[0] = t[2 * r];
                    if (row + 1 >= lo && row + 1 <= hi) q[1] = t[2 * r + 1];
                }
            }
        }
    }
    // The transposes go through LDS: tile registers --ds_write_b128--> [col][row] --ds_read_b128--> the lane's own
    // 16 samples, and back for the results.  Both round trips are software-pipelined against the recurrence:
    // while the dependent f64 chain of chunk c runs, the input transpose of chunk c+1 and the output transpose
    // of chunk c are in flight, so the LDS latency never sits on the chain.  One tile serves both directions:
    // the workgroup is a single wave and the LDS executes a wave's DS instructions in program order.
    __device__ __forceinline__ void lds_put_tile(const double t[16]) const
    {
        const int lane = threadIdx.x & 63, rp = lane & 7, cg = lane >> 3;
#pragma unroll
        for (int r = 0; r < 8; r++) *(double2 *)(lds + (8 * r + cg) * COL_LS + 2 * rp) = make_double2(t[2 * r], t[2 * r + 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void lds_get_tile(double t[16]) const
    {
        const int lane = threadIdx.x & 63, rp = lane & 7, cg = lane >> 3;
#pragma unroll
        for (int r = 0; r < 8; r++) { const double2 q = *(const double2 *)(lds + (8 * r + cg) * COL_LS + 2 * rp); t[2 * r] = q.x; t[2 * r + 1] = q.y; }
        __builtin_amdgcn_wave_barrier();
    }
    // the lane's own column, in sweep order (DIR < 0: v[e] = row rb + 15 - e)
    template <int DIR> __device__ __forceinline__ void lds_get_col(double v[16]) const
    {
        const double *L = lds + (threadIdx.x & 63) * COL_LS;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double2 q = *(const double2 *)(L + 2 * j);
            if (DIR > 0) { v[2 * j] = q.x; v[2 * j + 1] = q.y; } else { v[15 - 2 * j] = q.x; v[14 - 2 * j] = q.y; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    template <int DIR> __device__ __forceinline__ void lds_put_col(const double v[16]) const
    {
        double *L = lds + (threadIdx.x & 63) * COL_LS;
#pragma unroll
        for (int j = 0; j < 8; j++)
            *(double2 *)(L + 2 * j) = DIR > 0 ? make_double2(v[2 * j], v[2 * j + 1]) : make_double2(v[15 - 2 * j], v[14 - 2 * j]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // sweep `count` elements starting at row i0 in direction DIR (+1/-1); from_dst: read dst instead of src
    template <int DIR, class F> __device__ __forceinline__ void sweep(int i0, int count, bool from_dst, F f) const
    {
        if (count <= 0) return;
        const double *base = from_dst ? dst : src;
        const int lo = DIR > 0 ? i0 : i0 - count + 1, hi = DIR > 0 ? i0 + count - 1 : i0;    // rows [lo, hi]
        const int ta = lo >> 4, tb = hi >> 4, NT = tb - ta + 1;
        auto rbase = [&](int k) { return (DIR > 0 ? ta + k : tb - k) << 4; };
        double ring[COL_NB][16];
        int loaded = 0;
#pragma unroll
        for (int b = 0; b < COL_NB; b++)
            if (b < NT) { tile_load(base, rbase(loaded), ring[b]); loaded++; }
        double vcur[16], vnext[16], tprev[16];
        lds_put_tile(ring[0]); lds_get_col<DIR>(vcur);
        int prev_rb = 0; bool prev_partial = false;
        for (int k0 = 0; k0 < NT; k0 += COL_NB) {
#pragma unroll
            for (int s = 0; s < COL_NB; s++) {
                const int k = k0 + s;
                if (k < NT) {
                    const int rb = rbase(k);
                    // (1) input transpose of chunk k+1 (ring slot s+1), consumed next iteration
                    if (k + 1 < NT) { lds_put_tile(ring[(s + 1) % COL_NB]); lds_get_col<DIR>(vnext); }
                    // (2) refill ring slot s (its tile went through the LDS one iteration ago)
                    if (loaded < NT) { tile_load(base, rbase(loaded), ring[s]); loaded++; }
                    // (3) the recurrence on chunk k
                    const bool partial = rb < lo || rb + 15 > hi;
                    if (!partial) {
#pragma unroll
                        for (int e = 0; e < 16; e++) vcur[e] = f(vcur[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; e++) { const int row = DIR > 0 ? rb + e : rb + 15 - e; if (row >= lo && row <= hi) vcur[e] = f(vcur[e]); }
                    }
                    // (4) store chunk k-1 (its read-back was issued last iteration); output transpose of chunk k
                    if (k > 0) tile_store(prev_rb, tprev, lo, hi, prev_partial);
                    lds_put_col<DIR>(vcur); lds_get_tile(tprev);
                    prev_rb = rb; prev_partial = partial;
#pragma unroll
                    for (int e = 0; e < 16; e++) vcur[e] = vnext[e];
                }
            }
        }
        tile_store(prev_rb, tprev, lo, hi, prev_partial);
    }
};

// One line of ImageFiltering._imfilter_dim!(::TriggsSdika): left border, forward
// recursion, Triggs-Sdika right border, backward recursion, final scaling.
template <class IO>
__device__ __forceinline__ void iir_line(const IO &io, int n, const IIRCoef &k, bool fill0, const double *nrm, long nrm_s)
{
    const double a1 = k.a1, a2 = k.a2, a3 = k.a3;
    const double x0 = io.ld_src(0);
    const double iminus = fill0 ? 0.0 : x0;
    const double iplus = fill0 ? 0.0 : io.ld_src(n - 1);
    const double uminus = iminus / k.inv1masum;
    double o0 = ((x0 + a1 * uminus) + a2 * uminus) + a3 * uminus;
    double o1 = ((io.ld_src(1) + a1 * o0) + a2 * uminus) + a3 * uminus;
    double o2 = ((io.ld_src(2) + a1 * o1) + a2 * o0) + a3 * uminus;
    io.st(0, o0); io.st(1, o1); io.st(2, o2);
    double w3 = o0, w2 = o1, w1 = o2;
    io.template sweep<+1>(3, n - 3, false, [&](double x) {
        const double t = ((x + a1 * w1) + a2 * w2) + a3 * w3;
        w3 = w2; w2 = w1; w1 = t;
        return t;
    });
    const double uplus = iplus / k.inv1masum;
    const double vplus = uplus / k.inv1mbsum;
    const double d0 = w1 - uplus, d1 = w2 - uplus, d2 = w3 - uplus;
    const double vr0 = ((k.M[0] * d0 + k.M[1] * d1) + k.M[2] * d2) + vplus;
    const double vr1 = ((k.M[3] * d0 + k.M[4] * d1) + k.M[5] * d2) + vplus;
    const double vr2 = ((k.M[6] * d0 + k.M[7] * d1) + k.M[8] * d2) + vplus;
    double vA = vr0;                                              // v[n-1]
    double vB = ((w2 + a1 * vA) + a2 * vr1) + a3 * vr2;           // v[n-2]
    double vC = ((w3 + a1 * vB) + a2 * vA) + a3 * vr1;            // v[n-3]
    double v1 = vC, v2 = vB, v3 = vA;
    const double scale = k.scale;
    io.fence();                                                   // forward results visible to every lane of the wave
    if (nrm) {   // NA() border (constructor semantics, frames 1-2 only): plain loop
        io.st(n - 1, (vA * scale) / nrm[(long)(n - 1) * nrm_s]);
        io.st(n - 2, (vB * scale) / nrm[(long)(n - 2) * nrm_s]);
        io.st(n - 3, (vC * scale) / nrm[(long)(n - 3) * nrm_s]);
        for (int i = n - 4; i >= 0; i--) {
            const double t = ((io.ld_dst(i) + a1 * v1) + a2 * v2) + a3 * v3;
            io.st(i, (t * scale) / nrm[(long)i * nrm_s]);
            v3 = v2; v2 = v1; v1 = t;
        }
        return;
    }
    io.st(n - 1, vA * scale); io.st(n - 2, vB * scale); io.st(n - 3, vC * scale);
    io.template sweep<-1>(n - 4, n - 3, true, [&](double x) {
        const double t = ((x + a1 * v1) + a2 * v2) + a3 * v3;
        v3 = v2; v2 = v1; v1 = t;
        return t * scale;
    });
}

// dim-1 pass: one lane per column, LDS-transposed tile I/O.  src may differ from
// dst for plane 0 (the blur reads the layer and writes the scratch plane).
template <int NB>
__global__ __launch_bounds__(LINE_THREADS) void k_iir_cols(PlaneSet ps, const double *src0, int H, int W, int P, IIRPair cf)
{
    if (src0) src0 += (size_t)blockIdx.z * ps.zs;
    __shared__ __attribute__((aligned(16))) double tile[64 * COL_LS];
    const int pl = blockIdx.y;
    ColIO<NB> io;
    io.dst = ps_plane(ps, pl); io.src = (pl == 0 && src0) ? src0 : io.dst;
    io.H = H; io.W = W; io.P = P; io.x0 = blockIdx.x * LINE_THREADS; io.lds = tile; io.slo = 0; io.shi = W - 1;
    iir_line(io, H, ps_coef(ps, pl) == 0 ? cf.c[0] : cf.c[1], ps_fill0(ps, pl), nullptr, 0);
}

// dim-2 pass: one lane per row, in place; consecutive lanes = consecutive y.
__global__ __launch_bounds__(LINE_THREADS) void k_iir_rows(PlaneSet ps, int H, int W, int P, IIRPair cf)
{
    const int y = blockIdx.x * LINE_THREADS + threadIdx.x, pl = blockIdx.y;
    if (y >= H) return;
    RowIO io; io.src = ps_plane(ps, pl) + y; io.dst = ps_plane(ps, pl) + y; io.s = P;
    const double *nrm = ps_nrm(ps, pl);
    iir_line(io, W, ps_coef(ps, pl) == 0 ? cf.c[0] : cf.c[1], ps_fill0(ps, pl), nrm ? nrm + y : nullptr, P);
}

// dim-2 pass for batched, bandwidth-bound launches.  The backward sweep needs the forward results of the whole
// line, which do not fit on chip (64 lines x 1226 samples = 628 KB per wave); k_iir_rows writes them to HBM and
// reads them back (2 reads + 2 writes per sample).  Here the forward sweep only reads: it keeps its state every
// CK_B samples (3 doubles per block and line in a scratch buffer), and the backward sweep walks the blocks right to
// left, re-reads a block's input, recomputes its forward values in registers from the checkpoint (the same
// operations from the same state: bit-identical), runs the backward recurrence on them and stores once:
// 2 reads + 1 write per sample for 1.5x the arithmetic, which a bandwidth-bound launch has to spare.
#define CK_B 32
// Fused imresize! (even H only): plane 0 of a level with a successor is the blurred layer, whose only reader is k_resize.  With
// `rz.dst` set, the backward sweep of plane 0 does not store its results: every lane interpolates its row horizontally as the
// samples appear (right to left; the source column and weight of the current output column are wave-uniform), row pairs
// (2k, 2k+1) -- lanes (2k, 2k+1) of one wave -- are averaged through a shuffle and the even lanes store the next level's
// layer: the arithmetic of k_resize for an exact 2:1 row ratio, term by term.
struct RowResize { double *dst; int Hd, Wd, Pd; };
__device__ __forceinline__ double dpp_pair_next(double v)        // lanes 2k and 2k+1 both receive lane 2k+1's value (quad_perm [1,1,3,3])
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xF5, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xF5, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// every sample of this kernel is touched once per sweep by one wave, and the two sweeps of a line are ~2 GB of other lines' traffic
// apart: nontemporal loads and stores (no line kept for a reuse that cannot happen) -- k_iir_rows_ck -10 % (same-box A/B, S = 64)
#define RCK_LD(ptr) __builtin_nontemporal_load(ptr)
#define RCK_ST(ptr, v) __builtin_nontemporal_store(v, ptr)
// (the checkpoints stay plain accesses: nontemporal ones measured 1-2 % slower per build)
#define CK_LD(ptr) (*(ptr))
#define CK_ST(ptr, v) (*(ptr) = (v))
__global__ __launch_bounds__(LINE_THREADS) void k_iir_rows_ck(PlaneSet ps, int H, int W, int P, IIRPair cf, double *ck, RowResize rz)
{
    const int y = blockIdx.x * LINE_THREADS + threadIdx.x, pl = blockIdx.y;
    if (y >= H) return;
    const bool resize = pl == 0 && rz.dst != nullptr;
    // resize state: xo = current output column (1-based, descending), c0 = 0-based source column of its left sample, fx = weight
    const double rsx = (double)W / (double)(rz.Wd > 0 ? rz.Wd : 1), rox = 1 - 0.5 - rsx * (1 - 0.5);
    int xo = rz.Wd, c0 = -1; double fx = 0.0, tprev = 0.0;
    double *rzp = resize ? rz.dst + (size_t)blockIdx.z * ps.zs + (y >> 1) : nullptr;
    const bool exact2 = 2 * rz.Wd == W;                           // exact 2:1 columns: c = 2 xo - 0.5 -> ixx = 2 xo - 1, fx = 0.5 (what the general path computes, without the floor)
    auto rz_target = [&]() {                                     // k_resize: c = sx * x + ox; ixx = floor(c) clamped to [1, Ws - 1]; fx = c - ixx
        if (xo < 1) { c0 = -1; return; }
        if (exact2) { c0 = 2 * xo - 2; fx = 0.5; return; }
        const double c = rsx * xo + rox;
        int ixx = (int)floor(c);
        if (ixx > W - 1) ixx = W - 1;
        if (ixx < 1) ixx = 1;
        fx = c - ixx;
        c0 = __builtin_amdgcn_readfirstlane(ixx - 1);
    };
    if (resize) rz_target();
    auto emit = [&](int x, double val) {                         // called for x = W-1 .. 0 in descending order with the finished sample T[y, x]
        if (x == c0) {
            const double h = (1 - fx) * val + fx * tprev;        // r0 / r1 of k_resize for this lane's row
            const double hn = dpp_pair_next(h);                   // h of row y + 1 (the odd lane of the pair), no LDS round trip
            const double fy = 0.5;
            const double o = (1 - fy) * h + fy * hn;
            if ((y & 1) == 0) rzp[(size_t)(xo - 1) * rz.Pd] = o;
            xo--; rz_target();
        }
        tprev = val;
    };
    const size_t nlines = (size_t)gridDim.z * gridDim.y * gridDim.x * LINE_THREADS;
    const size_t lineid = ((size_t)blockIdx.z * gridDim.y + pl) * gridDim.x * LINE_THREADS + y;
    double *p = ps_plane(ps, pl) + y;
    const long s = P;
    const int n = W;
    const IIRCoef &k = ps_coef(ps, pl) == 0 ? cf.c[0] : cf.c[1];
    const bool fill0 = ps_fill0(ps, pl);
    const double a1 = k.a1, a2 = k.a2, a3 = k.a3, scale = k.scale;
    const double x0 = p[0];
    const double iminus = fill0 ? 0.0 : x0, iplus = fill0 ? 0.0 : p[(long)(n - 1) * s];
    const double uminus = iminus / k.inv1masum;
    const double o0 = ((x0 + a1 * uminus) + a2 * uminus) + a3 * uminus;
    const double o1 = ((p[s] + a1 * o0) + a2 * uminus) + a3 * uminus;
    const double o2 = ((p[2 * s] + a1 * o1) + a2 * o0) + a3 * uminus;
    // ---- pass A: forward over i = 3 .. n-1, read only, block by block (next block prefetched while the current
    //      one runs); the state before block j >= 1 is its checkpoint.  The last block absorbs the 3 trailing
    //      samples (n-3 .. n-1), which only the boundary computation needs. ----
    const int m = n - 6, nb = (m + CK_B - 1) / CK_B;            // the backward sweep needs forward values on [3, n-4]
    double w3 = o0, w2 = o1, w1 = o2;
    double cur[CK_B], nxt[CK_B];
    auto load_x = [&](int j, double *buf) {                     // block j = samples [3 + j CK_B, ...), CK_B of them (clamped reads)
        const double *q = p + (long)(3 + j * CK_B) * s;
        const int len = n - (3 + j * CK_B);                     // samples left in the line
        if (len >= CK_B) {
#pragma unroll
            for (int e = 0; e < CK_B; e++) buf[e] = RCK_LD(q + (long)e * s);
        } else {
#pragma unroll
            for (int e = 0; e < CK_B; e++) buf[e] = e < len ? q[(long)e * s] : 0.0;
        }
    };
    const int nfull = (n - 3) / CK_B;                           // blocks of pass A that are complete
    const int rem = (n - 3) - nfull * CK_B;                     // samples of the trailing partial block (block nfull), prefetched like the others
    load_x(0, cur);
    for (int j = 0; j < nfull; j++) {
        load_x(j + 1, nxt);                                     // (block nfull: clamped reads, the missing samples are zeros)
        if (j > 0 && j < nb) { double *c = ck + ((size_t)j * 3) * nlines + lineid; CK_ST(c, w1); CK_ST(c + nlines, w2); CK_ST(c + 2 * nlines, w3); }
#pragma unroll
        for (int e = 0; e < CK_B; e++) { const double t = ((cur[e] + a1 * w1) + a2 * w2) + a3 * w3; w3 = w2; w2 = w1; w1 = t; }
#pragma unroll
        for (int e = 0; e < CK_B; e++) cur[e] = nxt[e];
    }
    {   // remainder (< CK_B samples, in `cur`): its start may still be a checkpoint
        if (nfull > 0 && nfull < nb) { double *c = ck + ((size_t)nfull * 3) * nlines + lineid; CK_ST(c, w1); CK_ST(c + nlines, w2); CK_ST(c + 2 * nlines, w3); }
#pragma unroll
        for (int e = 0; e < CK_B; e++)
            if (e < rem) { const double t = ((cur[e] + a1 * w1) + a2 * w2) + a3 * w3; w3 = w2; w2 = w1; w1 = t; }
    }
    const bool have_last = nfull == nb - 1;                     // `cur` already holds the inputs of pass B's first block
    // ---- Triggs-Sdika right boundary (as iir_line) ----
    const double uplus = iplus / k.inv1masum, vplus = uplus / k.inv1mbsum;
    const double d0 = w1 - uplus, d1 = w2 - uplus, d2 = w3 - uplus;
    const double vr0 = ((k.M[0] * d0 + k.M[1] * d1) + k.M[2] * d2) + vplus;
    const double vr1 = ((k.M[3] * d0 + k.M[4] * d1) + k.M[5] * d2) + vplus;
    const double vr2 = ((k.M[6] * d0 + k.M[7] * d1) + k.M[8] * d2) + vplus;
    const double vA = vr0;
    const double vB = ((w2 + a1 * vA) + a2 * vr1) + a3 * vr2;
    const double vC = ((w3 + a1 * vB) + a2 * vA) + a3 * vr1;
    double v1 = vC, v2 = vB, v3 = vA;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // own checkpoints visible
    if (resize) { emit(n - 1, vA * scale); emit(n - 2, vB * scale); emit(n - 3, vC * scale); }
    else { p[(long)(n - 1) * s] = vA * scale; p[(long)(n - 2) * s] = vB * scale; p[(long)(n - 3) * s] = vC * scale; }
    // ---- pass B: blocks right to left; block j covers i in [3 + j CK_B, 3 + min((j+1) CK_B, m)) ----
    double f1n = 0, f2n = 0, f3n = 0, f1 = 0, f2 = 0, f3 = 0;
    auto load_ck = [&](int j, double &g1, double &g2, double &g3) {
        if (j > 0) { const double *c = ck + ((size_t)j * 3) * nlines + lineid; g1 = CK_LD(c); g2 = CK_LD(c + nlines); g3 = CK_LD(c + 2 * nlines); }
        else { g1 = o2; g2 = o1; g3 = o0; }
    };
    if (nb > 0) {
        // rightmost block: possibly partial (1 .. CK_B samples), predicated
        const int j = nb - 1, a = 3 + j * CK_B, len = m - j * CK_B;
        if (!have_last) load_x(j, cur);
        load_ck(j, f1, f2, f3);
        if (j > 0) { load_x(j - 1, nxt); load_ck(j - 1, f1n, f2n, f3n); }
#pragma unroll
        for (int e = 0; e < CK_B; e++)
            if (e < len) { const double t = ((cur[e] + a1 * f1) + a2 * f2) + a3 * f3; f3 = f2; f2 = f1; f1 = t; cur[e] = t; }
#pragma unroll
        for (int e = CK_B - 1; e >= 0; e--)
            if (e < len) { const double t = ((cur[e] + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = t; cur[e] = t * scale; if (resize) emit(a + e, cur[e]); }
        double *q = p + (long)a * s;
        if (!resize) {
#pragma unroll
            for (int e = 0; e < CK_B; e++) if (e < len) RCK_ST(q + (long)e * s, cur[e]);
        }
#pragma unroll
        for (int e = 0; e < CK_B; e++) cur[e] = nxt[e];
        f1 = f1n; f2 = f2n; f3 = f3n;
    }
    for (int j = nb - 2; j >= 0; j--) {                          // full blocks: straight-line code
        if (j > 0) { load_x(j - 1, nxt); load_ck(j - 1, f1n, f2n, f3n); }
#pragma unroll
        for (int e = 0; e < CK_B; e++) { const double t = ((cur[e] + a1 * f1) + a2 * f2) + a3 * f3; f3 = f2; f2 = f1; f1 = t; cur[e] = t; }
#pragma unroll
        for (int e = CK_B - 1; e >= 0; e--) { const double t = ((cur[e] + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = t; cur[e] = t * scale; if (resize) emit(3 + j * CK_B + e, cur[e]); }
        double *q = p + (long)(3 + j * CK_B) * s;
        if (!resize) {
#pragma unroll
            for (int e = 0; e < CK_B; e++) RCK_ST(q + (long)e * s, cur[e]);
        }
#pragma unroll
        for (int e = 0; e < CK_B; e++) cur[e] = nxt[e];
        f1 = f1n; f2 = f2n; f3 = f3n;
    }
    {   // i = 2, 1, 0: forward values o2, o1, o0
        double t = ((o2 + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = t; if (resize) emit(2, t * scale); else p[2 * s] = t * scale;
        t = ((o1 + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = t; if (resize) emit(1, t * scale); else p[s] = t * scale;
        t = ((o0 + a1 * v1) + a2 * v2) + a3 * v3; if (resize) emit(0, t * scale); else p[0] = t * scale;
    }
}

// dim-1 pass for bandwidth-bound batched launches: the checkpoint scheme of k_iir_rows_ck on column tiles.  Blocks
// are 32 rows = two aligned 16-row tiles; the forward sweep saves its state before every block, the backward sweep
// takes a block's two input tiles through the LDS transpose, recomputes the forward values of the lane's 32 samples
// in registers, runs the backward recurrence on them and stores the two tiles: 2R+1W instead of 2R+2W.  The next
// block's tiles are requested as soon as the current ones have gone through the transpose (no extra registers).
__global__ __launch_bounds__(LINE_THREADS) void k_iir_cols_ck(PlaneSet ps, const double *src0, int H, int W, int P, IIRPair cf, double *ck)
{
    if (src0) src0 += (size_t)blockIdx.z * ps.zs;
    __shared__ __attribute__((aligned(16))) double tile[64 * COL_LS];
    const int pl = blockIdx.y, lane = threadIdx.x & 63;
    ColIO<2> io;
    const double *sqsrc = ps_sq(ps, pl);                       // squared input (Iy -> Iyy, Ix -> Ixx), or nullptr
    const bool sq = sqsrc != nullptr;
    io.dst = ps_plane(ps, pl); io.src = sq ? sqsrc : ((pl == 0 && src0) ? src0 : io.dst);
    io.H = H; io.W = W; io.P = P; io.x0 = blockIdx.x * LINE_THREADS; io.lds = tile; io.slo = 0; io.shi = W - 1; io.nt = true;
    const size_t nlines = (size_t)gridDim.z * gridDim.y * gridDim.x * LINE_THREADS;
    const size_t lineid = (((size_t)blockIdx.z * gridDim.y + pl) * gridDim.x + blockIdx.x) * LINE_THREADS + lane;
    const int n = H;
    const IIRCoef &k = ps_coef(ps, pl) == 0 ? cf.c[0] : cf.c[1];
    const bool fill0 = ps_fill0(ps, pl);
    const double a1 = k.a1, a2 = k.a2, a3 = k.a3, scale = k.scale;
    auto in = [&](double v) { return sq ? v * v : v; };
    const double x0 = in(io.ld_src(0));
    const double iminus = fill0 ? 0.0 : x0, iplus = fill0 ? 0.0 : in(io.ld_src(n - 1));
    const double uminus = iminus / k.inv1masum;
    const double o0 = ((x0 + a1 * uminus) + a2 * uminus) + a3 * uminus;
    const double o1 = ((in(io.ld_src(1)) + a1 * o0) + a2 * uminus) + a3 * uminus;
    const double o2 = ((in(io.ld_src(2)) + a1 * o1) + a2 * o0) + a3 * uminus;
    double w3 = o0, w2 = o1, w1 = o2;
    double raw[32], x[32];
    // ---- pass A: forward over rows 3 .. n-1, read only; checkpoint before rows 32 b (b >= 1) ----
    {
        const int NT = ((n - 1) >> 4) + 1;
        io.tile_load(io.src, 0, raw);
        for (int t = 0; t < NT; t++) {
            const int rb = t << 4;
            io.lds_put_tile(raw); io.template lds_get_col<+1>(x);
            if (sq) {
#pragma unroll
                for (int e = 0; e < 16; e++) x[e] = x[e] * x[e];
            }
            if (t + 1 < NT) io.tile_load(io.src, rb + 16, raw);
            if ((t & 1) == 0 && t >= 2) { double *c = ck + ((size_t)(t >> 1) * 3) * nlines + lineid; c[0] = w1; c[nlines] = w2; c[2 * nlines] = w3; }
            if (rb >= 3 && rb + 15 <= n - 1) {
#pragma unroll
                for (int e = 0; e < 16; e++) { const double tt = ((x[e] + a1 * w1) + a2 * w2) + a3 * w3; w3 = w2; w2 = w1; w1 = tt; }
            } else {
#pragma unroll
                for (int e = 0; e < 16; e++) { const int row = rb + e; if (row >= 3 && row <= n - 1) { const double tt = ((x[e] + a1 * w1) + a2 * w2) + a3 * w3; w3 = w2; w2 = w1; w1 = tt; } }
            }
        }
    }
    // ---- Triggs-Sdika right boundary (as iir_line) ----
    const double uplus = iplus / k.inv1masum, vplus = uplus / k.inv1mbsum;
    const double d0 = w1 - uplus, d1 = w2 - uplus, d2 = w3 - uplus;
    const double vr0 = ((k.M[0] * d0 + k.M[1] * d1) + k.M[2] * d2) + vplus;
    const double vr1 = ((k.M[3] * d0 + k.M[4] * d1) + k.M[5] * d2) + vplus;
    const double vr2 = ((k.M[6] * d0 + k.M[7] * d1) + k.M[8] * d2) + vplus;
    const double vA = vr0;
    const double vB = ((w2 + a1 * vA) + a2 * vr1) + a3 * vr2;
    const double vC = ((w3 + a1 * vB) + a2 * vA) + a3 * vr1;
    double v1 = vC, v2 = vB, v3 = vA;
    io.fence();
    // ---- pass B: blocks of 32 rows, bottom to top; rows [3, n-4] carry the recurrence, n-3 .. n-1 and 2 .. 0 are direct ----
    const int NBk = ((n - 4) >> 5) + 1;
    const int ntile = P >> 4;                                     // tiles that exist in the pitched plane
    auto load_block = [&](int b) {
        io.tile_load(io.src, b << 5, raw);
        if (2 * b + 1 < ntile) io.tile_load(io.src, (b << 5) + 16, raw + 16);
    };
    if (n - 4 >= 3) load_block(NBk - 1);
    io.st(n - 1, vA * scale); io.st(n - 2, vB * scale); io.st(n - 3, vC * scale);
    for (int b = NBk - 1; b >= 0 && n - 4 >= 3; b--) {
        const int rb = b << 5, lo = rb > 3 ? rb : 3, hi = rb + 31 < n - 4 ? rb + 31 : n - 4;
        const bool two = 2 * b + 1 < ntile;
        io.lds_put_tile(raw); io.template lds_get_col<+1>(x);
        if (two) { io.lds_put_tile(raw + 16); io.template lds_get_col<+1>(x + 16); }
        if (sq) {
#pragma unroll
            for (int e = 0; e < 32; e++) x[e] = x[e] * x[e];
        }
        if (b > 0) load_block(b - 1);
        double f1, f2, f3;
        if (b > 0) { const double *c = ck + ((size_t)b * 3) * nlines + lineid; f1 = c[0]; f2 = c[nlines]; f3 = c[2 * nlines]; }
        else { f1 = o2; f2 = o1; f3 = o0; }
        if (lo == rb && hi == rb + 31) {
#pragma unroll
            for (int e = 0; e < 32; e++) { const double tt = ((x[e] + a1 * f1) + a2 * f2) + a3 * f3; f3 = f2; f2 = f1; f1 = tt; x[e] = tt; }
#pragma unroll
            for (int e = 31; e >= 0; e--) { const double tt = ((x[e] + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = tt; x[e] = tt * scale; }
        } else {
#pragma unroll
            for (int e = 0; e < 32; e++) { const int row = rb + e; if (row >= lo && row <= hi) { const double tt = ((x[e] + a1 * f1) + a2 * f2) + a3 * f3; f3 = f2; f2 = f1; f1 = tt; x[e] = tt; } }
#pragma unroll
            for (int e = 31; e >= 0; e--) { const int row = rb + e; if (row >= lo && row <= hi) { const double tt = ((x[e] + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = tt; x[e] = tt * scale; } }
        }
        double u[16];
        io.template lds_put_col<+1>(x); io.lds_get_tile(u);
        io.tile_store(rb, u, lo, hi, !(lo <= rb && hi >= rb + 15));
        if (two && hi >= rb + 16) {
            io.template lds_put_col<+1>(x + 16); io.lds_get_tile(u);
            io.tile_store(rb + 16, u, lo, hi, !(lo <= rb + 16 && hi >= rb + 31));
        }
    }
    {   // rows 2, 1, 0: forward values o2, o1, o0
        double tt = ((o2 + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = tt; io.st(2, tt * scale);
        tt = ((o1 + a1 * v1) + a2 * v2) + a3 * v3; v3 = v2; v2 = v1; v1 = tt; io.st(1, tt * scale);
        tt = ((o0 + a1 * v1) + a2 * v2) + a3 * v3; io.st(0, tt * scale);
    }
}

// ---- the whole dim-1 stage of a level in ONE kernel (bandwidth-bound batches) -----------------------------------
// k_scharr_products + k_iir_cols_ck read the layer 1.1 + 2 times, write Iy / Ix / Iy Ix, and read Iy, Ix, Iy Ix twice
// each (16 plane passes per level).  Here a 256-thread workgroup owns a strip of 62 columns and its four waves run the
// four dim-1 recurrences of the level -- wave 0: the sigma-1 blur of the layer, waves 1-3: the sigma-4 filter of Iy^2,
// Ix^2, Iy Ix -- straight from the LAYER: every wave takes the layer tiles through its own LDS transpose (lanes =
// columns x_l .. x_l + 63, x_l = first own column - 1: one halo column on each side), forms the Scharr terms of its
// 16 / 32 rows in registers (vertical taps from the lane's own samples + one halo row above and below, horizontal taps
// from the neighbour lanes: the arithmetic of k_scharr_products term by term), and feeds the products to the
// checkpointed recurrence of k_iir_cols_ck.  Waves 1 and 2 also store Iy / Ix on the way.  The four waves sweep the
// strip together (one barrier per block keeps them within a block of each other), so the layer is fetched from HBM once
// per pass and served to the other three waves by the L2: 2 reads + 6 writes of a plane per level instead of 9 + 7.
#define CF4_COLS 62
struct ColsFusedArgs {
    const double *L;                    // the level's layer (image 0 of the batch)
    double *T;                          // dim-1 blurred layer (scratch plane), nullptr at the last level
    double *Iy, *Ix, *Qyy, *Qxx, *Qyx;  // gradient planes; product planes (dim-1 filtered here, finished by the row pass)
    int H, W, P; size_t zs;
    // level 0 with the ingest fused: the layer is read from the caller's dense column-major image of each stream (src_kind 1:
    // Float64, 2: 8-bit, converted raw / 255 like k_gather_images_u8) through a device-side pointer table, and wave 0 writes the
    // pitched layer plane on the way (src_kind 0: the layer plane itself, already ingested)
    int src_kind; const void *const *srctab;
    // tolerance build (TOL): the product planes leave this kernel as EXCLUSIVE SUFFIX SUMS along y of their dim-1-filtered values,
    // E[y] = sum_{i > y} v[i] (formed bottom-up, the order the backward sweep emits the rows), and the column totals go to
    // tot[(z * 3 + role - 1) * tot_stride + x]; the row kernel k_rows_tol takes C1[y] = tot - E[y] = cumsum along dim 1
    double *tot; int tot_stride;
    // tolerance build, even H with P % 32 == 0: the dim-1-blurred layer leaves HALVED along y -- Th[k] = (v[2k] + v[2k+1]) / 2, column pitch
    // P / 2 -- the dim-1 half of imresize! (a blur along x and an average along y commute); the row kernel finishes it
    int dec;
};

// neighbour lanes of the whole wave through DPP (wave_shr:1 / wave_shl:1 of the GFX9 family): lane i receives lane i-1 / i+1,
// lane 0 / 63 keeps its own value -- __shfl_up / __shfl_down by one lane without the LDS crossbar (ds_bpermute_b32: two per
// double, ~64 per 8-row Scharr step, queued behind the tile traffic of all eight waves of the CU)
__device__ __forceinline__ double wave_prev(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_next(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// x[0..N) holds the lane's layer samples of rows rb .. rb+N-1 on entry, the recurrence inputs of its role on exit
// (ROLE 1: Iy^2, 2: Ix^2, 3: Iy Ix); g receives Iy (ROLE 1) / Ix (ROLE 2).  top / bot: the layer at rows rb-1 / rb+N.
// one step of the third-order recurrence: the reference's operation order, or (tolerance build, mode 3) three fused multiply-adds whose
// dependent chain is the single operation that takes the newest state
template <bool TOL>
__device__ __forceinline__ double iir3(double x, double a1, double w1, double a2, double w2, double a3, double w3)
{
    if (TOL) return __builtin_fma(a1, w1, __builtin_fma(a2, w2, __builtin_fma(a3, w3, x)));
    return ((x + a1 * w1) + a2 * w2) + a3 * w3;
}
// the two factors of the separable Scharr pair along one line: derivative (-1, 0, 1) / 2 and smoothing (3, 10, 3) / 16 of (a, b, c).
// Bit-exact build: imfilter's accumulation from 0.0, term by term; tolerance build: the same sums in 2 + 3 operations
template <bool TOL>
__device__ __forceinline__ void scharr_pair(double a, double b, double c, double &d, double &s)
{
    const double dk[3] = {-1.0 / 2, 0.0 / 2, 1.0 / 2}, sk[3] = {3.0 / 16, 10.0 / 16, 3.0 / 16};
    if (TOL) { d = 0.5 * (c - a); s = __builtin_fma(sk[0], a + c, sk[1] * b); return; }
    double dd = 0.0; dd += a * dk[0]; dd += b * dk[1]; dd += c * dk[2];
    double ss = 0.0; ss += a * sk[0]; ss += b * sk[1]; ss += c * sk[2];
    d = dd; s = ss;
}
template <int ROLE, int N, bool TOL = false>
__device__ __forceinline__ void cf4_inputs(double *x, double top, double bot, int rb, int H, bool edgeL, bool edgeR, double *g)
{
    const double dk[3] = {-1.0 / 2, 0.0 / 2, 1.0 / 2}, sk[3] = {3.0 / 16, 10.0 / 16, 3.0 / 16};
    double a = top;
#pragma unroll
    for (int e = 0; e < N; e++) {
        const int row = rb + e;
        const double b = x[e];
        double c = e + 1 < N ? x[e + 1] : bot;
        const double aa = row == 0 ? b : a;                       // replicate border (scharr_col, border 0)
        c = row == H - 1 ? b : c;
        double iy = 0.0, ix = 0.0;
        double d, s;
        scharr_pair<TOL>(aa, b, c, d, s);
        if (ROLE == 1 || ROLE == 3) {
            double dl = wave_prev(d), dr = wave_next(d);
            dl = edgeL ? d : dl; dr = edgeR ? d : dr;
            if (TOL) iy = __builtin_fma(sk[0], dl + dr, sk[1] * d);
            else { iy += dl * sk[0]; iy += d * sk[1]; iy += dr * sk[2]; }
        }
        if (ROLE == 2 || ROLE == 3) {
            double sl = wave_prev(s), sr = wave_next(s);
            sl = edgeL ? s : sl; sr = edgeR ? s : sr;
            if (TOL) ix = 0.5 * (sr - sl);
            else { ix += sl * dk[0]; ix += s * dk[1]; ix += sr * dk[2]; }
        }
        double prod;
        if (ROLE == 1) { prod = iy * iy; g[e] = iy; }
        else if (ROLE == 2) { prod = ix * ix; g[e] = ix; }
        else prod = iy * ix;
        asm volatile("" : "+v"(prod));          // materialise the row's input HERE: otherwise the arithmetic behind the shuffles sinks to
        x[e] = prod;                            // its use in the recurrence and all 32 rows' shuffle results stay live (500 registers)
        a = b;
    }
}

// Shared LDS blocks of a workgroup, [column][row] with strides chosen so that a lane = column access of consecutive rows
// (ds_*_b128) is conflict-free: LB = layer rows rb-2 .. rb+35 of the 64 columns (row rb at index 2), IYB / IXB = Iy / Ix
// of rows rb .. rb+31, QB = output staging of wave 3.  After the inputs have been consumed LB / IYB / IXB are the output
// staging of waves 0 / 1 / 2.
// workgroup barrier that orders the LDS traffic only.  __syncthreads() also waits for every outstanding global load and store
// of the wave (s_waitcnt vmcnt(0)): the next block's prefetch and the previous block's stores would be drained at each of the
// three or four barriers of a block, a full memory round trip each time.  Global memory is never shared between the waves here.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#define CF4_DEFER_U8 1
#define CF4_LS 38
#define CF4_GS 34
#define CF4_LDS_DOUBLES (64 * CF4_LS + 3 * 64 * CF4_GS + 3 * 64)      // + the blur wave's three trailing rows (tolerance build, halved layer)

// phase 2: wave w forms Iy, Ix of rows rb + 8w .. rb + 8w + 7 for the 64 columns (lane = column) from the shared layer block
template <bool TOL>
__device__ __forceinline__ void cf4_scharr8(const double *LB, double *IYB, double *IXB, int w, int rb, int H, bool edgeL, bool edgeR)
{
    const double dk[3] = {-1.0 / 2, 0.0 / 2, 1.0 / 2}, sk[3] = {3.0 / 16, 10.0 / 16, 3.0 / 16};
    const int lane = threadIdx.x & 63;
    double v[12];                                                 // layer rows rb + 8w - 2 .. rb + 8w + 9
    const double *q = LB + lane * CF4_LS + 8 * w;
#pragma unroll
    for (int j = 0; j < 6; j++) { const double2 t = *(const double2 *)(q + 2 * j); v[2 * j] = t.x; v[2 * j + 1] = t.y; }
    double iy8[8], ix8[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int row = rb + 8 * w + e;
        const double b = v[2 + e];
        const double a = row == 0 ? b : v[1 + e];                 // replicate border (scharr_col, border 0)
        const double c = row == H - 1 ? b : v[3 + e];
        double d, s;
        scharr_pair<TOL>(a, b, c, d, s);
        double dl = wave_prev(d), dr = wave_next(d), sl = wave_prev(s), sr = wave_next(s);
        dl = edgeL ? d : dl; dr = edgeR ? d : dr; sl = edgeL ? s : sl; sr = edgeR ? s : sr;
        double iy = 0.0, ix = 0.0;
        if (TOL) { iy = __builtin_fma(sk[0], dl + dr, sk[1] * d); ix = 0.5 * (sr - sl); }
        else {
            iy += dl * sk[0]; iy += d * sk[1]; iy += dr * sk[2];
            ix += sl * dk[0]; ix += s * dk[1]; ix += sr * dk[2];
        }
        asm volatile("" : "+v"(iy), "+v"(ix));                    // finish the row here (keeps the shuffle results from piling up in registers)
        iy8[e] = iy; ix8[e] = ix;
    }
    double *py = IYB + lane * CF4_GS + 8 * w, *px = IXB + lane * CF4_GS + 8 * w;
#pragma unroll
    for (int j = 0; j < 4; j++) { *(double2 *)(py + 2 * j) = make_double2(iy8[2 * j], iy8[2 * j + 1]); *(double2 *)(px + 2 * j) = make_double2(ix8[2 * j], ix8[2 * j + 1]); }
}

template <int ROLE, bool TOL, bool DEC>
__device__ __forceinline__ void cols_fused_wave(const ColsFusedArgs &A, const IIRCoef &k, double *ck, double *LB, double *IYB, double *IXB, double *QB, const double *lut)
{
    const int lane = threadIdx.x & 63, rp = lane & 7, cg = lane >> 3;
    const int H = A.H, W = A.W, P = A.P, n = H;
    const int own0 = CF4_COLS * (int)blockIdx.x;
    const int xl = own0 > 0 ? own0 - 1 : 0;                      // first column of the workgroup (lane 0)
    const size_t z = (size_t)blockIdx.z * A.zs;
    const bool active = ROLE != 0 || A.T != nullptr;              // last level: no blur, wave 0 only helps with the shared phases
    const int nroles = A.T != nullptr ? 4 : 3;
    double *stage = ROLE == 0 ? LB : ROLE == 1 ? IYB : ROLE == 2 ? IXB : QB;
    ColIO<2> io;                                                  // global tile addressing (lanes <-> aligned 128-byte lines)
    io.src = A.L + z;
    io.dst = (ROLE == 0 ? A.T : ROLE == 1 ? A.Qyy : ROLE == 2 ? A.Qxx : A.Qyx) + z;
    io.H = H; io.W = W; io.P = P; io.x0 = xl; io.lds = stage;
    io.slo = own0; io.shi = own0 + CF4_COLS - 1 < W - 1 ? own0 + CF4_COLS - 1 : W - 1; io.nt = true;
    ColIO<2> iog = io;                                            // gradient plane written on the way (ROLE 1: Iy, ROLE 2: Ix)
    iog.dst = (ROLE == 1 ? A.Iy : A.Ix) + z;
    const int col = xl + lane;
    const bool edgeL = col == 0, edgeR = col == W - 1;
    const size_t nlines = (size_t)gridDim.z * nroles * gridDim.x * LINE_THREADS;
    const size_t lineid = (((size_t)blockIdx.z * nroles + (ROLE - (4 - nroles))) * gridDim.x + blockIdx.x) * LINE_THREADS + lane;
    const double a1 = k.a1, a2 = k.a2, a3 = k.a3, scale = k.scale;
    const int w = ROLE;                                           // wave index inside the workgroup
    // the layer: the pitched plane, or (level 0, fused ingest) the stream's dense source image
    const int skind = A.src_kind;
    // (the pointers are given their address spaces explicitly: a pointer loaded from the table, or the LDS table behind a
    //  parameter, is generic to the compiler, and FLAT loads make every s_waitcnt of the kernel a full vmcnt(0) drain)
    typedef const __attribute__((address_space(1))) double *gsrc_f64;
    typedef const __attribute__((address_space(1))) unsigned char *gsrc_u8;
    typedef const __attribute__((address_space(3))) double *lds_f64;
    const void *simg = skind ? A.srctab[blockIdx.z] : nullptr;
    const gsrc_f64 simg_d = (gsrc_f64)simg; const gsrc_u8 simg_b = (gsrc_u8)simg; const lds_f64 lutl = (lds_f64)lut;
    auto src_at = [&](int row, int colx) -> double {              // dense source sample, indices in range; 8-bit: lut[v] = (double)v / 255.0 (k_gather_images_u8's conversion, tabulated)
        return skind == 1 ? simg_d[(size_t)row + (size_t)colx * H] : lutl[simg_b[(size_t)row + (size_t)colx * H]];
    };
    auto ld_layer = [&](int row) -> double { return skind ? src_at(row, io.xown()) : io.ld_src(row); };
    auto tile_layer = [&](int rb, double *t) {                    // rows rb + 2 rp, + 1 of columns x0 + 8 r + cg (the global tile layout)
        if (!skind) { io.tile_load(io.src, rb, t); return; }
        const int r0 = rb + 2 * rp < H ? rb + 2 * rp : H - 1;
        const bool pair = rb + 2 * rp + 1 < H;                    // both rows exist: one load of two vertically adjacent samples
        if (skind == 2) {                                        // 8-bit: the raw bytes travel in t[r] (two per register pair, nothing waits for them here);
#pragma unroll                                                    // publish() turns them into samples where the tile is consumed
            for (int r = 0; r < 8; r++) {
                int cx = xl + 8 * r + cg; cx = cx < W ? cx : W - 1;
                const gsrc_u8 q = simg_b + (size_t)r0 + (size_t)cx * H;
                const int b0 = q[0], b1 = q[pair ? 1 : 0];
                t[r] = __hiloint2double(b1, b0);
            }
            if (!CF4_DEFER_U8) {
#pragma unroll
                for (int r = 7; r >= 0; r--) { const int lo = __double2loint(t[r]), hi = __double2hiint(t[r]); t[2 * r] = lutl[lo]; t[2 * r + 1] = lutl[hi]; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                int cx = xl + 8 * r + cg; cx = cx < W ? cx : W - 1;
                const gsrc_f64 q = simg_d + (size_t)r0 + (size_t)cx * H;
                t[2 * r] = q[0]; t[2 * r + 1] = pair ? q[1] : q[0];
            }
        }
    };
    // recurrence input of a single row (boundary rows of the filter): the same arithmetic on scalar loads
    // the six layer rows the boundary inputs need (0 .. 3, n-2, n-1), requested together: one memory round trip instead of one per use
    double brow[6];
    {
        const int rr[6] = {0, 1, 2, n > 3 ? 3 : n - 1, n - 2, n - 1};
        if (!skind) {
#pragma unroll
            for (int q = 0; q < 6; q++) brow[q] = io.ld_src(rr[q]);
        } else if (skind == 1) {
#pragma unroll
            for (int q = 0; q < 6; q++) brow[q] = simg_d[(size_t)rr[q] + (size_t)io.xown() * H];
        } else {
            unsigned char bb[6];
#pragma unroll
            for (int q = 0; q < 6; q++) bb[q] = simg_b[(size_t)rr[q] + (size_t)io.xown() * H];
#pragma unroll
            for (int q = 0; q < 6; q++) brow[q] = lutl[bb[q]];
        }
    }
    auto brow_at = [&](int y) { return y <= 3 ? (y == 0 ? brow[0] : y == 1 ? brow[1] : y == 2 ? brow[2] : brow[3]) : (y == n - 2 ? brow[4] : brow[5]); };
    auto in_row = [&](int y) {
        double v[1] = {brow_at(y)};
        if (ROLE != 0) { double gdummy[1]; cf4_inputs<ROLE, 1, TOL>(v, brow_at(y > 0 ? y - 1 : 0), brow_at(y + 1 < H ? y + 1 : H - 1), y, H, edgeL, edgeR, gdummy); }
        return v[0];
    };
    const double x0 = in_row(0);
    const double iminus = x0, iplus = in_row(n - 1);
    const double uminus = iminus / k.inv1masum;
    const double o0 = iir3<TOL>(x0, a1, uminus, a2, uminus, a3, uminus);
    const double o1 = iir3<TOL>(in_row(1), a1, o0, a2, uminus, a3, uminus);
    const double o2 = iir3<TOL>(in_row(2), a1, o1, a2, o0, a3, uminus);
    double w3 = o0, w2 = o1, w1 = o2;
    const int NBk = ((n - 1) >> 5) + 1;                           // 32-row blocks of the plane
    const int ntile = P >> 4;
    double pre[16], x[32], u[16];
    // phase 1 operands of block b, requested a block ahead: wave 0 the block's first tile, wave 1 its second tile, wave 2 the
    // layer row above the block, wave 3 the row below it
    auto prefetch = [&](int b) {
        const int rb = b << 5;
        if (ROLE == 0) tile_layer(rb, pre);
        else if (ROLE == 1) {
            if (2 * b + 1 < ntile) tile_layer(rb + 16, pre);
            else {
#pragma unroll
                for (int e = 0; e < 16; e++) pre[e] = 0.0;
            }
        }
        else {                                                    // halo row above (wave 2) / below (wave 3) the block
            const int hrow = ROLE == 2 ? (rb > 0 ? rb - 1 : 0) : (rb + 32 < H ? rb + 32 : H - 1);
            if (skind == 2 && CF4_DEFER_U8) pre[0] = __hiloint2double(0, (int)simg_b[(size_t)hrow + (size_t)io.xown() * H]);      // raw byte, converted in publish()
            else pre[0] = ld_layer(hrow);
        }
    };
    auto publish = [&]() {                                        // phase 1: operands -> shared layer block
        if (ROLE <= 1 && skind == 2 && CF4_DEFER_U8) {            // 8-bit tile: bytes -> samples (table of v / 255.0)
#pragma unroll
            for (int r = 7; r >= 0; r--) {                        // in place, top down: pre[2r], pre[2r+1] <- the two bytes in pre[r]
                const int lo = __double2loint(pre[r]), hi = __double2hiint(pre[r]);
                pre[2 * r] = lutl[lo]; pre[2 * r + 1] = lutl[hi];
            }
        }
        if (ROLE <= 1) {
#pragma unroll
            for (int r = 0; r < 8; r++) *(double2 *)(LB + (8 * r + cg) * CF4_LS + 2 + 16 * ROLE + 2 * rp) = make_double2(pre[2 * r], pre[2 * r + 1]);
        } else LB[lane * CF4_LS + (ROLE == 2 ? 1 : 34)] = skind == 2 && CF4_DEFER_U8 ? (double)lutl[__double2loint(pre[0]) & 255] : pre[0];
    };
    auto read_inputs = [&]() {                                    // phase 3: the lane's 32 recurrence inputs
        if (ROLE == 0) {
            const double *q = LB + lane * CF4_LS + 2;
#pragma unroll
            for (int j = 0; j < 16; j++) { const double2 t = *(const double2 *)(q + 2 * j); x[2 * j] = t.x; x[2 * j + 1] = t.y; }
        } else if (ROLE == 1 || ROLE == 2) {
            const double *q = (ROLE == 1 ? IYB : IXB) + lane * CF4_GS;
#pragma unroll
            for (int j = 0; j < 16; j++) { const double2 t = *(const double2 *)(q + 2 * j); x[2 * j] = t.x * t.x; x[2 * j + 1] = t.y * t.y; }
        } else {
            const double *q = IYB + lane * CF4_GS, *r = IXB + lane * CF4_GS;
#pragma unroll
            for (int j = 0; j < 16; j++) { const double2 t = *(const double2 *)(q + 2 * j), v = *(const double2 *)(r + 2 * j); x[2 * j] = t.x * v.x; x[2 * j + 1] = t.y * v.y; }
        }
    };
    auto get_tile = [&](const double *blk, int t, double *uu) {   // rows 16 t .. 16 t + 15 of a [column][CF4_GS] block in global tile layout
#pragma unroll
        for (int r = 0; r < 8; r++) { const double2 q = *(const double2 *)(blk + (8 * r + cg) * CF4_GS + 16 * t + 2 * rp); uu[2 * r] = q.x; uu[2 * r + 1] = q.y; }
    };
    // ---- pass A: forward over rows 3 .. n-1, read only; checkpoint before rows 32 b (b >= 1) ----
    prefetch(0);
    for (int b = 0; b < NBk; b++) {
        const int rb = b << 5;
        lds_barrier();                                            // the previous block's shared data has been consumed
        publish();
        lds_barrier();  
        if (b + 1 < NBk) prefetch(b + 1);
        cf4_scharr8<TOL>(LB, IYB, IXB, w, rb, H, edgeL, edgeR);
        lds_barrier();  
        if (!active) continue;
        read_inputs();
        if (b >= 1) { double *c = ck + ((size_t)b * 3) * nlines + lineid; c[0] = w1; c[nlines] = w2; c[2 * nlines] = w3; }
        if (rb >= 3 && rb + 31 <= n - 1) {
#pragma unroll
            for (int e = 0; e < 32; e++) { const double tt = iir3<TOL>(x[e], a1, w1, a2, w2, a3, w3); w3 = w2; w2 = w1; w1 = tt; }
        } else {
#pragma unroll
            for (int e = 0; e < 32; e++) { const int row = rb + e; if (row >= 3 && row <= n - 1) { const double tt = iir3<TOL>(x[e], a1, w1, a2, w2, a3, w3); w3 = w2; w2 = w1; w1 = tt; } }
        }
    }
    // ---- Triggs-Sdika right boundary (as iir_line) ----
    const double uplus = iplus / k.inv1masum, vplus = uplus / k.inv1mbsum;
    const double d0 = w1 - uplus, d1 = w2 - uplus, d2 = w3 - uplus;
    const double vr0 = ((k.M[0] * d0 + k.M[1] * d1) + k.M[2] * d2) + vplus;
    const double vr1 = ((k.M[3] * d0 + k.M[4] * d1) + k.M[5] * d2) + vplus;
    const double vr2 = ((k.M[6] * d0 + k.M[7] * d1) + k.M[8] * d2) + vplus;
    const double vA = vr0;
    const double vB = iir3<TOL>(w2, a1, vA, a2, vr1, a3, vr2);
    const double vC = iir3<TOL>(w3, a1, vB, a2, vA, a3, vr1);
    double v1 = vC, v2 = vB, v3 = vA;
    io.fence();
    // ---- pass B: blocks bottom to top.  Every block is visited (the gradient planes need all rows); rows [3, n-4] carry the
    //      recurrence, n-3 .. n-1 and 2 .. 0 are direct ----
    // the forward state at the top of a block (its checkpoint) is requested a block ahead, like the layer operands: loaded where
    // it is used, its s_waitcnt would also wait for every store issued before it (vmcnt counts in order) -- the block's gradient
    // lines and the previous block's results -- and the stores would never overlap the recurrences
    double f1n = o2, f2n = o1, f3n = o0;
    auto load_ck = [&](int b) {
        if (b > 0) { const double *c = ck + ((size_t)b * 3) * nlines + lineid; f1n = c[0]; f2n = c[nlines]; f3n = c[2 * nlines]; }
        else { f1n = o2; f2n = o1; f3n = o0; }
    };
    prefetch(NBk - 1);
    if (active) load_ck(NBk - 1);
    constexpr bool SFX = TOL && ROLE != 0;                        // this wave's outputs leave as exclusive suffix sums (see ColsFusedArgs::tot)
    constexpr bool dec = TOL && DEC && ROLE == 0;                 // the blurred layer leaves halved along y (ColsFusedArgs::dec; a compile-time variant: the plain store path is not in this instantiation)
    double sfx = 0.0;
    if (active && !SFX && !dec) { io.st(n - 1, vA * scale); io.st(n - 2, vB * scale); io.st(n - 3, vC * scale); }
    double *tailv = QB + 64 * CF4_GS + 3 * lane;                  // (dec) rows n-1, n-2, n-3 wait here for their block: not in registers across the loop
    if (active && dec) { tailv[0] = vA * scale; tailv[1] = vB * scale; tailv[2] = vC * scale; }
    if (active && SFX) { io.st(n - 1, sfx); sfx = __builtin_fma(vA, scale, sfx); io.st(n - 2, sfx); sfx = __builtin_fma(vB, scale, sfx); io.st(n - 3, sfx); sfx = __builtin_fma(vC, scale, sfx); }
    for (int b = NBk - 1; b >= 0; b--) {
        const int rb = b << 5, lo = rb > 3 ? rb : 3, hi = rb + 31 < n - 4 ? rb + 31 : n - 4;     // recurrence rows of the block (may be empty: lo > hi)
        const bool two = 2 * b + 1 < ntile;
        lds_barrier();  
        publish();
        lds_barrier();  
        cf4_scharr8<TOL>(LB, IYB, IXB, w, rb, H, edgeL, edgeR);
        lds_barrier();  
        if (active) read_inputs();
        if (ROLE == 0 && skind) {                                 // fused ingest: the block's layer rows -> the pitched layer plane
            ColIO<2> iol = io; iol.dst = const_cast<double *>(A.L) + z;
            const int ghi = H - 1;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (t == 0 || (two && rb + 16 <= ghi)) {
#pragma unroll
                    for (int r = 0; r < 8; r++) { const double2 q = *(const double2 *)(LB + (8 * r + cg) * CF4_LS + 2 + 16 * t + 2 * rp); u[2 * r] = q.x; u[2 * r + 1] = q.y; }
                    iol.tile_store(rb + 16 * t, u, rb + 16 * t, ghi, rb + 16 * t + 15 > ghi);
                }
            }
        }
        if (ROLE == 1 || ROLE == 2) {                             // Iy / Ix of the block -> their planes
            const double *blk = ROLE == 1 ? IYB : IXB;
            const int ghi = H - 1;
            get_tile(blk, 0, u); iog.tile_store(rb, u, rb, ghi, rb + 15 > ghi);
            if (two && rb + 16 <= ghi) { get_tile(blk, 1, u); iog.tile_store(rb + 16, u, rb + 16, ghi, rb + 31 > ghi); }
        }
        lds_barrier();                                            // inputs consumed: the shared blocks become output staging
        double f1 = f1n, f2 = f2n, f3 = f3n;
        if (b > 0) { prefetch(b - 1); if (active) load_ck(b - 1); }
        if (!active || (lo > hi && !dec)) continue;               // (a trailing block may hold rows n-3 .. n-1 only)
        if (lo > hi) {}
        else if (lo == rb && hi == rb + 31) {
#pragma unroll
            for (int e = 0; e < 32; e++) { const double tt = iir3<TOL>(x[e], a1, f1, a2, f2, a3, f3); f3 = f2; f2 = f1; f1 = tt; x[e] = tt; }
#pragma unroll
            for (int e = 31; e >= 0; e--) { const double tt = iir3<TOL>(x[e], a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt;
                                            if (SFX) { x[e] = sfx; sfx = __builtin_fma(tt, scale, sfx); } else x[e] = tt * scale; }
        } else {
#pragma unroll
            for (int e = 0; e < 32; e++) { const int row = rb + e; if (row >= lo && row <= hi) { const double tt = iir3<TOL>(x[e], a1, f1, a2, f2, a3, f3); f3 = f2; f2 = f1; f1 = tt; x[e] = tt; } }
#pragma unroll
            for (int e = 31; e >= 0; e--) { const int row = rb + e; if (row >= lo && row <= hi) { const double tt = iir3<TOL>(x[e], a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt;
                                                                                             if (SFX) { x[e] = sfx; sfx = __builtin_fma(tt, scale, sfx); } else x[e] = tt * scale; } }
        }
        if (dec) {
            // the block's 32 rows go to the staging block as in the plain path; the three rows below the recurrence and (block 0) the three
            // above it -- which the plain path stores one by one -- are patched in there (LDS takes a run-time index, registers do not);
            // then row pairs are averaged in place and the 16 half-height rows leave as one tile
            double *q = stage + lane * CF4_GS;
#pragma unroll
            for (int j = 0; j < 16; j++) *(double2 *)(q + 2 * j) = make_double2(x[2 * j], x[2 * j + 1]);
            if (rb + 31 >= n - 3) {
                if (n - 1 >= rb) q[n - 1 - rb] = tailv[0];
                if (n - 2 >= rb) q[n - 2 - rb] = tailv[1];
                if (n - 3 >= rb) q[n - 3 - rb] = tailv[2];
            }
            if (b == 0) {
                double tt = iir3<TOL>(o2, a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; q[2] = tt * scale;
                tt = iir3<TOL>(o1, a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; q[1] = tt * scale;
                tt = iir3<TOL>(o0, a1, v1, a2, v2, a3, v3); q[0] = tt * scale;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {                         // in place: pair j writes q[2j], q[2j+1] < the next pair's reads q[4j+4 ..]
                const double2 a = *(const double2 *)(q + 4 * j), c = *(const double2 *)(q + 4 * j + 2);
                *(double2 *)(q + 2 * j) = make_double2(0.5 * a.x + 0.5 * a.y, 0.5 * c.x + 0.5 * c.y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            get_tile(stage, 0, u);
            const int rbh = rb >> 1, hih = (H >> 1) - 1;
            ColIO<2> iod = io; iod.P = P >> 1; iod.H = H >> 1;    // the half-height plane Th (same memory as T)
            iod.tile_store(rbh, u, rbh, hih, rbh + 15 > hih);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        {   // outputs -> own staging block (lane = column), back in tile layout, stored as aligned lines
            double *q = stage + lane * CF4_GS;
#pragma unroll
            for (int j = 0; j < 16; j++) *(double2 *)(q + 2 * j) = make_double2(x[2 * j], x[2 * j + 1]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            get_tile(stage, 0, u);
            io.tile_store(rb, u, lo, hi, !(lo <= rb && hi >= rb + 15));
            if (two && hi >= rb + 16) { get_tile(stage, 1, u); io.tile_store(rb + 16, u, lo, hi, !(lo <= rb + 16 && hi >= rb + 31)); }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (active && !SFX && !dec) {   // rows 2, 1, 0: forward values o2, o1, o0
        double tt = iir3<TOL>(o2, a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; io.st(2, tt * scale);
        tt = iir3<TOL>(o1, a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; io.st(1, tt * scale);
        tt = iir3<TOL>(o0, a1, v1, a2, v2, a3, v3); io.st(0, tt * scale);
    }
    if (active && SFX) {
        double tt = iir3<TOL>(o2, a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; io.st(2, sfx); sfx = __builtin_fma(tt, scale, sfx);
        tt = iir3<TOL>(o1, a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; io.st(1, sfx); sfx = __builtin_fma(tt, scale, sfx);
        tt = iir3<TOL>(o0, a1, v1, a2, v2, a3, v3); io.st(0, sfx); sfx = __builtin_fma(tt, scale, sfx);
        if (io.valid()) A.tot[((size_t)blockIdx.z * 3 + (ROLE - 1)) * A.tot_stride + io.xown()] = sfx;
    }
}

template <bool TOL, bool DEC = false>
__global__ __launch_bounds__(256, 2) void k_cols_fused(ColsFusedArgs A, IIRPair cf, double *ck)
{
    __shared__ __attribute__((aligned(16))) double sh[CF4_LDS_DOUBLES];
    __shared__ double lut[256];                                  // 8-bit ingest: (double)v / 255.0
    double *LB = sh, *IYB = sh + 64 * CF4_LS, *IXB = IYB + 64 * CF4_GS, *QB = IXB + 64 * CF4_GS;
    if (A.src_kind == 2) { lut[threadIdx.x] = (double)threadIdx.x / 255.0; __syncthreads(); }
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w == 0) cols_fused_wave<0, TOL, DEC>(A, cf.c[0], ck, LB, IYB, IXB, QB, lut);
    else if (w == 1) cols_fused_wave<1, TOL, DEC>(A, cf.c[1], ck, LB, IYB, IXB, QB, lut);
    else if (w == 2) cols_fused_wave<2, TOL, DEC>(A, cf.c[1], ck, LB, IYB, IXB, QB, lut);
    else cols_fused_wave<3, TOL, DEC>(A, cf.c[1], ck, LB, IYB, IXB, QB, lut);
}

// integral_image!, lucas_kanade.jl:131-138: cumsum along dim 1 ...
template <int NB>
__global__ __launch_bounds__(LINE_THREADS) void k_cum_cols(PlaneSet ps, int H, int W, int P)
{
    __shared__ __attribute__((aligned(16))) double tile[64 * COL_LS];
    const int pl = blockIdx.y;
    ColIO<NB> io; io.dst = ps_plane(ps, pl); io.src = io.dst; io.H = H; io.W = W; io.P = P; io.x0 = blockIdx.x * LINE_THREADS; io.lds = tile; io.slo = 0; io.shi = W - 1;
    double acc = io.ld_src(0);
    io.template sweep<+1>(1, H - 1, false, [&](double x) { acc = acc + x; return acc; });
}
// ... then along dim 2.
__global__ __launch_bounds__(LINE_THREADS) void k_cum_rows(PlaneSet ps, int H, int W, int P)
{
    const int y = blockIdx.x * LINE_THREADS + threadIdx.x, pl = blockIdx.y;
    if (y >= H) return;
    double *p = ps_plane(ps, pl) + y;
    double acc = p[0];
    stream_line<8, 8>(p + P, p + P, P, W - 1, [&](double x) { acc = acc + x; return acc; });
}

// integral_image! in ONE pass for batched launches (one read + one write per sample instead of two of each).
// One workgroup owns a plane; wave w owns the 64-row band [64 w, 64 w + 63] and walks it left to right in blocks of
// 32 columns through its own LDS block Cb[column][row]: (a) the block arrives as aligned 128-byte lines (16 bytes per
// lane) and is written to Cb; (b) lanes = columns: each lane adds its column's 64 samples onto the column's running
// sum, which enters from the band above (wave w-1 publishes the sums of its bottom row per block in LDS, with a
// ready counter; double-buffered with a consumed counter as back-pressure) and leaves to the band below; (c) lanes =
// rows: each lane adds its row's 32 column sums left to right onto the row's carry (a register: the wave keeps its
// rows for the whole plane); (d) the block is stored as aligned lines.  The bands form a pipeline (wave w runs one
// block behind wave w-1), so a plane takes (blocks + bands) block-times instead of blocks x bands.  Every sum is
// formed exactly as by k_cum_cols followed by k_cum_rows (dim 1 first, each element added to its predecessor's
// result).  Bound by its dependent adds, not by bandwidth (~100 workgroups at S = 32): it runs on the forked stream
// next to the following level's kernels and leaves the HBM to them.
#define CF_LS 66                         // Cb column stride in doubles (even: ds_*_b128 alignment)
#define CF_W 32                          // columns per block
#define CF_MAXW 8                        // bands (waves) per workgroup: H <= 512 (2 waves per SIMD -> 256 VGPRs each)
#define CF_SEGW 4                        // bands per row segment of taller planes (chained workgroups, see k_cum_fused)
// block of 64 rows x 32 columns as aligned lines: sub-tile s, column group r -> rows 2rp, 2rp+1 of column 8r + cg.  Rows past
// H inside the pitch and columns past W are read as well (in-bounds by the layout, never stored).
__device__ __forceinline__ void cf_load_block(const double *plane, int P, int x0, int r0, unsigned voff, double (&raw)[4][8])
{
#pragma unroll
    for (int sub = 0; sub < 4; sub++) {
        const int rb = min(r0 + sub * 16, P - 16);
#pragma unroll
        for (int r = 0; r < 4; r++) { const v2d q = __builtin_nontemporal_load((const v2d *)(plane + ((size_t)(x0 + 8 * r) * P + rb) + voff)); raw[sub][2 * r] = q.x; raw[sub][2 * r + 1] = q.y; }
    }
}
// Planes taller than CF_MAXW bands (1080-row frames) are cut into row SEGMENTS of CF_SEGW bands (256 rows: 68 KB of LDS, two workgroups per
// compute unit -- the 5 x 3 x 32 workgroups of a 32-image FHD launch are resident at once), one workgroup each (blockIdx.x): the band
// chain continues across the workgroups -- the last wave of segment g publishes its bottom row's column sums, block by block, in global
// memory (xc, agent-scope stores + a counting flag per (image, plane, boundary)), wave 0 of segment g + 1 takes them as its carry.  A
// segment only ever waits for a workgroup with a smaller blockIdx.x (dispatched before it); the consumer zeroes the flag when it is done,
// so a replay of the launch finds it clear.  Same sums in the same order as one workgroup would form them.
// (SEG = false: the single-workgroup kernel, compiled without the hand-over code -- with it the 6-band launches of the 370-row frames ran 1-3 % slower)
template <bool SEG>
__global__ __launch_bounds__(CF_MAXW * 64) void k_cum_fused(PlaneSet ps, int H, int W, int P, double *xc, int *xf)
{
    extern __shared__ __attribute__((aligned(16))) double cf_lds[];          // [nw][CF_W * CF_LS] blocks, then [nw][2][CF_W] column carries
    __shared__ int s_ready[CF_MAXW], s_done[CF_MAXW];            // read and written with relaxed workgroup-scope atomics: plain ds_read / ds_write (as `volatile` they were
                                                                  // FLAT accesses with a vmcnt(0) in front of every poll: the block prefetch was drained once per block)
    const int pl = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rp = lane & 7, cg = lane >> 3;
    double *plane = ps_plane(ps, pl);
    // LDS pointers carry their address space: as generic pointers they became FLAT accesses, which turn every s_waitcnt of the
    // kernel into a vmcnt(0) drain of the block prefetch and the stores; the hand-off fences order LDS only, for the same reason
    typedef __attribute__((address_space(3))) double ldsd;
    typedef __attribute__((address_space(3))) v2d ldsv2;
    ldsd *Cb = (ldsd *)cf_lds + (size_t)w * CF_W * CF_LS;
    ldsd *carry_out = (ldsd *)cf_lds + (size_t)nw * CF_W * CF_LS + (size_t)w * 2 * CF_W;  // published by this wave
    const ldsd *carry_in = carry_out - 2 * CF_W;                                           // published by wave w - 1
    auto flag_ld = [](int *f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto flag_st = [](int *f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (lane == 0) { flag_st(&s_ready[w], 0); flag_st(&s_done[w], 0); }
    __syncthreads();
    const int seg = SEG ? (int)blockIdx.x : 0, nseg = SEG ? (int)gridDim.x : 1;
    const int nwh = SEG ? min(nw, (H - seg * nw * 64 + 63) / 64) : nw;           // bands of this segment: nw = blockDim.x / 64 per segment, the last one may be short
    if (SEG && w >= nwh) return;                                  // (after the one workgroup barrier of the kernel)
    const int r0 = (seg * nw + w) * 64, hr = H - r0 < 64 ? H - r0 : 64;       // rows of this band that exist (>= 1 by the launch)
    // hand-over between row segments: carry row of W doubles and a flag per (image, plane, boundary)
    const size_t xslot = ((size_t)blockIdx.z * 3 + pl) * (size_t)(nseg > 1 ? nseg - 1 : 1);
    double *xc_out = xc + (xslot + seg) * (size_t)W; int *xf_out = xf + xslot + seg;                  // published by this segment's last wave (seg + 1 < nseg)
    const double *xc_in = xc + (xslot + seg - 1) * (size_t)W; int *xf_in = xf + xslot + seg - 1;       // taken by this segment's wave 0 (seg > 0)
    const int ncb = (W + CF_W - 1) / CF_W;
    const unsigned voff = (unsigned)(cg * P + 2 * rp);
    double raw[4][8];
    double rcarry = 0.0;                                          // this lane's row: running sum along x
    cf_load_block(plane, P, 0, r0, voff, raw);
    for (int cb = 0; cb < ncb; cb++) {
        const int x0 = cb * CF_W, wc = W - x0 < CF_W ? W - x0 : CF_W;
        // (a) block -> Cb[column][row]
#pragma unroll
        for (int sub = 0; sub < 4; sub++)
#pragma unroll
            for (int r = 0; r < 4; r++) { const v2d t2 = {raw[sub][2 * r], raw[sub][2 * r + 1]}; *(ldsv2 *)(Cb + (8 * r + cg) * CF_LS + sub * 16 + 2 * rp) = t2; }
        cf_load_block(plane, P, cb + 1 < ncb ? x0 + CF_W : x0, r0, voff, raw);       // next block (the last iteration re-reads, unused)
        // column sums of the band above
        double acc = 0.0;
        if (SEG && w == 0 && seg > 0) {
            // (no agent-scope fences: an acquire would invalidate the XCD's L2 under every workgroup on it, a release write it back, once per
            //  block -- measured 6.5 vs 5.0 ms per 32-image build.  The carries and the flag are agent-scope atomics, which are performed at the
            //  memory side; the producer's s_waitcnt vmcnt(0) orders them)
            if (lane == 0) while (__hip_atomic_load(xf_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= cb) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (lane < CF_W && x0 + lane < W) acc = __hip_atomic_load(xc_in + x0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (w > 0) {
            while (flag_ld(&s_ready[w - 1]) <= cb) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if (lane < CF_W) acc = carry_in[(cb & 1) * CF_W + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) flag_st(&s_done[w], cb + 1);                   // slot cb & 1 may be overwritten with block cb + 2
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        // (b) column sums, lane = column
        if (lane < CF_W) {
            ldsd *c = Cb + lane * CF_LS;
            if (hr == 64 && (SEG ? r0 > 0 : w > 0)) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    double v[32];
#pragma unroll
                    for (int j = 0; j < 16; j++) { const v2d q = *(const ldsv2 *)(c + 32 * half + 2 * j); v[2 * j] = q.x; v[2 * j + 1] = q.y; }
#pragma unroll
                    for (int e = 0; e < 32; e++) { acc = acc + v[e]; v[e] = acc; }
#pragma unroll
                    for (int j = 0; j < 16; j++) { const v2d t2 = {v[2 * j], v[2 * j + 1]}; *(ldsv2 *)(c + 32 * half + 2 * j) = t2; }
                }
            } else {
                for (int e = 0; e < hr; e++) { acc = (r0 + e == 0) ? c[e] : acc + c[e]; c[e] = acc; }
            }
        }
        if (w + 1 < nwh) {                                        // publish the bottom row's sums to the band below
            while (flag_ld(&s_done[w + 1]) + 2 <= cb) __builtin_amdgcn_s_sleep(1);
            if (lane < CF_W) carry_out[(cb & 1) * CF_W + lane] = acc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) flag_st(&s_ready[w], cb + 1);
        }
        else if (SEG && seg + 1 < nseg) {                         // ... or to the segment below, through global memory
            if (lane < CF_W && x0 + lane < W) __hip_atomic_store(xc_out + x0 + lane, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the carries have arrived before the flag leaves
            if (lane == 0) __hip_atomic_store(xf_out, cb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        // (c) row sums across the block's columns, lane = row
        if (lane < hr) {
            ldsd *c = Cb + lane;
            if (wc == CF_W && cb > 0) {
                double v[CF_W];
#pragma unroll
                for (int j = 0; j < CF_W; j++) v[j] = c[j * CF_LS];
#pragma unroll
                for (int j = 0; j < CF_W; j++) { rcarry = rcarry + v[j]; v[j] = rcarry; }
#pragma unroll
                for (int j = 0; j < CF_W; j++) c[j * CF_LS] = v[j];
            } else {
                for (int j = 0; j < wc; j++) { rcarry = (cb == 0 && j == 0) ? c[j * CF_LS] : rcarry + c[j * CF_LS]; c[j * CF_LS] = rcarry; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        // (d) store as aligned lines (rows >= H and columns >= W masked)
#pragma unroll
        for (int sub = 0; sub < 4; sub++) {
            const int rb = r0 + sub * 16;
            if (rb < H) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int col = x0 + 8 * r + cg, row = rb + 2 * rp;
                    const v2d q = *(const ldsv2 *)(Cb + (8 * r + cg) * CF_LS + sub * 16 + 2 * rp);
                    double *g = plane + ((size_t)(x0 + 8 * r) * P + rb) + voff;
                    if (col < W) {
                        if (row + 1 < H) __builtin_nontemporal_store(q, (v2d *)g);
                        else if (row < H) g[0] = q.x;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    if (SEG && w == 0 && seg > 0 && lane == 0) __hip_atomic_store(xf_in, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every block consumed: clear for the next launch
}

// ---- tolerance mode ("fast", mode 3): parallel recurrences -------------------------
// The exact kernels above are bound by the dependent f64 chain: 68 cycles per IIR
// step x line length (measured, scripts/ubench/dep_chain.hip), on the few CUs that a
// few hundred lines can occupy.  Here a 1024-thread workgroup owns LPW lines and
// cuts each into up to 1024/LPW segments of SL <= 16 samples, one per thread:
//   (load) the thread reads its SL samples ONCE into registers;
//   (A) runs the recurrence from a zero state, keeps the end state;
//   (B) entry states by a two-level fold of the affine maps s -> M^SL s + z through
//       LDS (powers of the 3x3 companion matrix M come from the host);
//   (C) re-runs the SAME recurrence from the true entry state, in registers;
// the backward pass repeats A-C right-to-left on the register-resident forward
// values, then stores.  HBM traffic is exactly one read and one write per sample
// (the algorithmic minimum) and lines spread over ~190 workgroups instead of 24.
// Inside a segment the arithmetic is the sequential one; only the entry states are
// rounded differently (tested: planes <= 1e-11 relative to the exact mode).
#define PAR_T 1024
#define PAR_SLMAX 16
#define PAR_G 8
struct SegPow { double P[2][PAR_G][9]; double Plast[2][9]; };   // M^(SL*q), q = 1..8; M^(len_last - 1)

__device__ __forceinline__ void mv3(const double *P, double &a, double &b, double &c, double za, double zb, double zc)
{
    const double n1 = __builtin_fma(P[0], a, __builtin_fma(P[1], b, __builtin_fma(P[2], c, za)));      // (mode-3 kernels only: contracted)
    const double n2 = __builtin_fma(P[3], a, __builtin_fma(P[4], b, __builtin_fma(P[5], c, zb)));
    const double n3 = __builtin_fma(P[6], a, __builtin_fma(P[7], b, __builtin_fma(P[8], c, zc)));
    a = n1; b = n2; c = n3;
}

// entry state of segment index `k` (0-based in fold order) of line l: two-level fold.
// Z: zero-state end states [3][nslots][LPW] indexed by fold order through `slot(k)`.
template <int ZG, int ZL, int GG, int GL, int ZN, int GN, class Slot>
__device__ __forceinline__ void fold_entry(const SegPow &sp, int cs, double (*Z)[ZN], double (*GT)[GN],
                                           int l, int k, bool has, Slot slot, double s0a, double s0b, double s0c,
                                           double &ea, double &eb, double &ec)
{
    const int q = k % PAR_G, grp = k / PAR_G;
    const double *P1 = sp.P[cs][0], *P8 = sp.P[cs][PAR_G - 1];
    double a = 0.0, b = 0.0, c = 0.0;
    if (has) for (int i = 0; i < q; i++) { const int s = slot(grp * PAR_G + i); mv3(P1, a, b, c, Z[0][s * ZG + l * ZL], Z[1][s * ZG + l * ZL], Z[2][s * ZG + l * ZL]); }
    if (has && q == PAR_G - 1) {
        double ta = a, tb = b, tc = c; const int s = slot(k);
        mv3(P1, ta, tb, tc, Z[0][s * ZG + l * ZL], Z[1][s * ZG + l * ZL], Z[2][s * ZG + l * ZL]);
        GT[0][grp * GG + l * GL] = ta; GT[1][grp * GG + l * GL] = tb; GT[2][grp * GG + l * GL] = tc;
    }
    __syncthreads();
    double Sa = s0a, Sb = s0b, Sc = s0c;
    if (has) {
        for (int h = 0; h < grp; h++) mv3(P8, Sa, Sb, Sc, GT[0][h * GG + l * GL], GT[1][h * GG + l * GL], GT[2][h * GG + l * GL]);
        if (q > 0) mv3(sp.P[cs][q - 1], Sa, Sb, Sc, 0.0, 0.0, 0.0);
    }
    ea = Sa + a; eb = Sb + b; ec = Sc + c;
}

// T threads own LPW lines of T / LPW segment slots each (1024 / 8: up to 128 segments per line; 256 / 8: 32 segments of <= 16 samples, the
// variant of a single image's columns -- a segment's share of the two folds costs more instructions than its samples, and with four
// waves per SIMD the kernel is bound by instruction issue, not by the dependent chain)
template <bool COLS, int T = PAR_T, int LPW = 8>
__global__ __launch_bounds__(T) void k_iir_seg(PlaneSet ps, const double *src0, int H, int W, int P, IIRPair cf, SegPow sp, int SL, int cum_mask)
{
    if (src0) src0 += (size_t)blockIdx.z * ps.zs;
    constexpr int NSEG = T / LPW, ZN = T + 8, GN = T / PAR_G + 16;
    // lanes -> (line, segment): along columns consecutive lanes take consecutive segments of ONE column (contiguous memory: a wave reads
    // 64 x SL consecutive samples); along rows consecutive lanes take the same segment of consecutive rows.  The [segment][line] LDS
    // arrays are laid out to match (segment-fastest with a pad / line-fastest): conflict-free either way
    constexpr int ZG = COLS ? 1 : LPW, ZL = COLS ? NSEG + 1 : 1, GG = COLS ? 1 : LPW, GL = COLS ? NSEG / PAR_G + 2 : 1;
    __shared__ double Z[3][ZN];
    __shared__ double GT[3][GN];
    __shared__ double Fin[3][LPW];
    const int t = threadIdx.x, l = COLS ? t / NSEG : t % LPW, g = COLS ? t % NSEG : t / LPW, pl = blockIdx.y;
    const int nlines = COLS ? W : H, n = COLS ? H : W;
    const int line = blockIdx.x * LPW + l;
    const bool valid = line < nlines;
    const int lc = valid ? line : nlines - 1;              // idle lanes shadow the last line (reads only)
    const long stride = COLS ? 1 : P;
    const size_t off = COLS ? (size_t)lc * P : (size_t)lc;
    double *dstp = ps_plane(ps, pl) + off;
    const double *srcp = ((pl == 0 && src0) ? src0 : ps_plane(ps, pl)) + off;
    const int cs = ps_coef(ps, pl);
    const IIRCoef &k = cf.c[cs];
    const double a1 = k.a1, a2 = k.a2, a3 = k.a3, scale = k.scale;
    const int nseg = (n + SL - 1) / SL;
    const bool has = g < nseg;
    const int b = g * SL, len = has ? min(n, b + SL) - b : 0;
    const bool lastseg = has && g == nseg - 1;
    double x[PAR_SLMAX];
#pragma unroll
    for (int j = 0; j < PAR_SLMAX; j++) x[j] = (j < len) ? srcp[(long)(b + j) * stride] : 0.0;
    const double x0 = srcp[0], xlast = srcp[(long)(n - 1) * stride];
    const double uminus = x0 / k.inv1masum;
    // ---------------- forward: A ----------------
    {
        double w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll
        for (int j = 0; j < PAR_SLMAX; j++) if (j < len) { const double tt = iir3<true>(x[j], a1, w1, a2, w2, a3, w3); w3 = w2; w2 = w1; w1 = tt; }
        if (has) { Z[0][g * ZG + l * ZL] = w1; Z[1][g * ZG + l * ZL] = w2; Z[2][g * ZG + l * ZL] = w3; }
    }
    __syncthreads();
    // ---------------- forward: B, C ----------------
    {
        double w1, w2, w3;
        fold_entry<ZG, ZL, GG, GL, ZN, GN>(sp, cs, Z, GT, l, g, has, [](int kk) { return kk; }, uminus, uminus, uminus, w1, w2, w3);
#pragma unroll
        for (int j = 0; j < PAR_SLMAX; j++) if (j < len) { const double tt = iir3<true>(x[j], a1, w1, a2, w2, a3, w3); w3 = w2; w2 = w1; w1 = tt; x[j] = tt; }
        if (lastseg) { Fin[0][l] = w1; Fin[1][l] = w2; Fin[2][l] = w3; }
    }
    __syncthreads();
    // ---------------- Triggs-Sdika right boundary ----------------
    const double f1 = Fin[0][l], f2 = Fin[1][l], f3 = Fin[2][l];
    const double uplus = xlast / k.inv1masum, vplus = uplus / k.inv1mbsum;
    const double d0 = f1 - uplus, d1 = f2 - uplus, d2 = f3 - uplus;
    const double vr0 = ((k.M[0] * d0 + k.M[1] * d1) + k.M[2] * d2) + vplus;
    const double vr1 = ((k.M[3] * d0 + k.M[4] * d1) + k.M[5] * d2) + vplus;
    const double vr2 = ((k.M[6] * d0 + k.M[7] * d1) + k.M[8] * d2) + vplus;
    // ---------------- backward: A' (sample n-1 is not part of the recurrence: v[n-1] = vr0) ----------------
    const int blen = lastseg ? len - 1 : len;              // samples of this segment that the backward recurrence visits
    {
        double v1 = 0.0, v2 = 0.0, v3 = 0.0;
#pragma unroll
        for (int j = PAR_SLMAX - 1; j >= 0; j--) if (j < blen) { const double tt = iir3<true>(x[j], a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; }
        if (has) { Z[0][g * ZG + l * ZL] = v1; Z[1][g * ZG + l * ZL] = v2; Z[2][g * ZG + l * ZL] = v3; }
    }
    __syncthreads();
    // ---------------- backward: B', C' ----------------
    {
        // state after the (short) last segment, then fold the full segments right-to-left
        double s0a = vr0, s0b = vr1, s0c = vr2;
        mv3(sp.Plast[cs], s0a, s0b, s0c, Z[0][(nseg - 1) * ZG + l * ZL], Z[1][(nseg - 1) * ZG + l * ZL], Z[2][(nseg - 1) * ZG + l * ZL]);
        double v1, v2, v3;
        const int kq = nseg - 2 - g;                        // fold order of the full segments
        const bool hasq = has && !lastseg;
        fold_entry<ZG, ZL, GG, GL, ZN, GN>(sp, cs, Z, GT, l, hasq ? kq : 0, hasq, [nseg](int kk) { return nseg - 2 - kk; }, s0a, s0b, s0c, v1, v2, v3);
        if (lastseg) { v1 = vr0; v2 = vr1; v3 = vr2; }
#pragma unroll
        for (int j = PAR_SLMAX - 1; j >= 0; j--) if (j < blen) { const double tt = iir3<true>(x[j], a1, v1, a2, v2, a3, v3); v3 = v2; v2 = v1; v1 = tt; x[j] = tt * scale; }
#pragma unroll
        for (int j = 0; j < PAR_SLMAX; j++) if (lastseg && j == len - 1) x[j] = vr0 * scale;
    }
    // planes of cum_mask: the running sum along the line follows in registers (k_cum_seg's arithmetic on the same values: segment totals
    // folded in two levels through LDS) -- the product planes of a single image leave with their dim-1 integral, one kernel and one
    // read + write of three planes less per level
    if ((cum_mask >> pl) & 1) {                                   // workgroup-uniform
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < PAR_SLMAX; j++) if (j < len) acc = acc + x[j];
        __syncthreads();                                          // the backward fold's LDS arrays are free
        if (has) Z[0][g * ZG + l * ZL] = acc;
        __syncthreads();
        const int q = g % PAR_G, grp = g / PAR_G;
        double pre = 0.0;
        if (has) for (int i = 0; i < q; i++) pre = pre + Z[0][(grp * PAR_G + i) * ZG + l * ZL];
        if (has && q == PAR_G - 1) GT[0][grp * GG + l * GL] = pre + Z[0][g * ZG + l * ZL];
        __syncthreads();
        double base = 0.0;
        if (has) for (int h = 0; h < grp; h++) base = base + GT[0][h * GG + l * GL];
        acc = base + pre;
#pragma unroll
        for (int j = 0; j < PAR_SLMAX; j++) if (j < len) { acc = acc + x[j]; x[j] = acc; }
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < PAR_SLMAX; j++) if (j < len) dstp[(long)(b + j) * stride] = x[j];
    }
}

template <bool COLS, int T = PAR_T, int LPW = 8>
__global__ __launch_bounds__(T) void k_cum_seg(PlaneSet ps, int H, int W, int P, int SL)
{
    constexpr int NSEG = T / LPW;
    constexpr int ZG = COLS ? 1 : LPW, ZL = COLS ? NSEG + 1 : 1, GG = COLS ? 1 : LPW, GL = COLS ? NSEG / PAR_G + 2 : 1;   // k_iir_seg's lane mapping and LDS layouts
    __shared__ double Zs[T + 8];
    __shared__ double Gs[T / PAR_G + 16];
    const int t = threadIdx.x, l = COLS ? t / NSEG : t % LPW, g = COLS ? t % NSEG : t / LPW, pl = blockIdx.y;
    const int nlines = COLS ? W : H, n = COLS ? H : W;
    const int line = blockIdx.x * LPW + l;
    const bool valid = line < nlines;
    const int lc = valid ? line : nlines - 1;
    const long stride = COLS ? 1 : P;
    double *p = ps_plane(ps, pl) + (COLS ? (size_t)lc * P : (size_t)lc);
    const int nseg = (n + SL - 1) / SL;
    const bool has = g < nseg;
    const int b = g * SL, len = has ? min(n, b + SL) - b : 0;
    double x[PAR_SLMAX];
#pragma unroll
    for (int j = 0; j < PAR_SLMAX; j++) x[j] = (j < len) ? p[(long)(b + j) * stride] : 0.0;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < PAR_SLMAX; j++) if (j < len) acc = acc + x[j];
    if (has) Zs[g * ZG + l * ZL] = acc;
    __syncthreads();
    const int q = g % PAR_G, grp = g / PAR_G;
    double pre = 0.0;
    if (has) for (int i = 0; i < q; i++) pre = pre + Zs[(grp * PAR_G + i) * ZG + l * ZL];
    if (has && q == PAR_G - 1) Gs[grp * GG + l * GL] = pre + Zs[g * ZG + l * ZL];
    __syncthreads();
    double base = 0.0;
    if (has) for (int h = 0; h < grp; h++) base = base + Gs[h * GG + l * GL];
    acc = base + pre;
    if (valid) {
#pragma unroll
        for (int j = 0; j < PAR_SLMAX; j++) if (j < len) { acc = acc + x[j]; p[(long)(b + j) * stride] = acc; }
    }
}

// ---- tolerance build (mode 3, batches): the whole dim-2 stage of a level in ONE kernel, 1 R + 1 W per plane ----------------------------
// k_iir_rows_ck + k_cum_fused move 8 R + 3.25 W + 3 R + 3 W plane passes per level because a bit-identical recurrence needs a line's
// samples twice, a line apart.  Here a workgroup owns 16 ROWS of one plane (16 consecutive y = one aligned 128-byte line per
// column) and cuts them into NS column segments of exactly SL samples, one per thread (NS = 32: 512 threads, two workgroups per CU --
// their load / compute / store phases overlap; NS = 64 for rows wider than 1280 samples): the thread loads its samples
// ONCE into registers, runs the forward recurrence from a zero state, the true entry states follow from a two-level fold of the affine
// maps s -> M^SL s + z through LDS (k_iir_seg's scheme), the thread re-runs the recurrence from its true entry state, the same right to
// left for the backward recurrence on the register-resident forward values; then, still in registers,
//   * product planes: the running sum along x (local prefix, segment totals folded through LDS) -- the finished integral image.  Their
//     input is k_cols_fused<TOL>'s output: exclusive suffix sums E along y of the dim-1-filtered products + the column totals, i.e. the
//     dim-1 running sum C1[y] = tot - E[y] (a running sum along y and a filter along x commute: different dimensions, both linear);
//   * the blurred layer: imresize! into the next level's layer (k_resize's arithmetic: horizontal interpolation per lane; the row pairs
//     were already averaged by k_cols_fused<TOL, DEC> -- the plane arrives at half height -- or are averaged here through DPP; even
//     heights), or a plain store (odd heights: k_resize follows).
// The fold's uniform power matrices are SGPR operands read from the kernel-argument segment (as LDS copies they cost 36+ vector registers
// and the 40-sample variant spilled); only the per-lane power M^(SL q) comes from LDS.
// Inside a segment the arithmetic is the sequential one; the entry states, the order of the two running sums and the suffix-sum form
// of the dim-1 sum round differently: planes within 1e-11 of the exact mode relative to the plane's magnitude (tests/test_gpu_tol_batch.py).
#define RT_R 16                         // rows per workgroup
#define RT_NS 32                        // column segments per row (the default variant)
struct RowsTolArgs {
    double *p[4];           // [blurred layer (dim 1 done)], Qyy, Qxx, Qyx (image 0 of the batch)
    int coef[4];            // IIRCoef index
    int kind[4];            // 0: blurred layer, 1: product plane as suffix sums E + totals (from k_cols_fused<TOL>), 2: product plane holding its dim-1 running sum already (single images: k_iir_seg + k_cum_seg along y)
    int n, nq0;             // planes; index of the first product plane
    size_t zs;
    const double *tot; int tot_stride;
    RowResize rz;           // kind 0: next level's layer if rz.dst
    int dec;                // kind 0: the plane is the half-height Th of k_cols_fused<TOL> (H / 2 rows, pitch P / 2): no row pairing left to do
    int dbg;                // SLAMHIP_RT_DBG (timing experiments only): 1 = loads + stores without the arithmetic (invalid planes)
};
// (this kernel is tolerance mode by definition: its multiply-adds are fused -- half the f64 instructions of the -ffp-contract=off form)
__device__ __forceinline__ double rt_step(double x, double a1, double a2, double a3, double w1, double w2, double w3)
{
    return __builtin_fma(a3, w3, __builtin_fma(a2, w2, __builtin_fma(a1, w1, x)));
}
__device__ __forceinline__ void mv3f(const double *P, double &a, double &b, double &c, double za, double zb, double zc)
{
    const double n1 = __builtin_fma(P[0], a, __builtin_fma(P[1], b, __builtin_fma(P[2], c, za)));
    const double n2 = __builtin_fma(P[3], a, __builtin_fma(P[4], b, __builtin_fma(P[5], c, zb)));
    const double n3 = __builtin_fma(P[6], a, __builtin_fma(P[7], b, __builtin_fma(P[8], c, zc)));
    a = n1; b = n2; c = n3;
}
// entry state of segment `k` (0-based in fold order) of row l: two-level fold over groups of PAR_G segments (fold_entry's scheme).
// Z: zero-state end states [3][NS][RT_R] indexed through slot(fold index); GT: group totals [3][NS / PAR_G + 1][RT_R]
template <int NS, int R, class Slot>
__device__ __forceinline__ void rt_fold(const SegPow &sp, int cs, const double *PW, double (*Z)[NS][R], double (*GT)[NS / PAR_G + 1][R], int l, int k, bool has, Slot slot,
                                        double s0a, double s0b, double s0c, double &ea, double &eb, double &ec)
{
    const int q = k % PAR_G, grp = k / PAR_G;
    // M^SL and M^(8 SL) are the same for every lane: read from the kernel-argument segment (cs is wave-uniform -> scalar loads, SGPR
    // operands of the multiply-adds, no vector registers); only the per-lane power M^(SL q) comes from the LDS copy PW
    const double *P1 = sp.P[cs][0], *P8 = sp.P[cs][PAR_G - 1];
    double a = 0.0, b = 0.0, c = 0.0;
    if (has) {
#pragma unroll 1
        for (int i = 0; i < q; i++) { const int s = slot(grp * PAR_G + i); mv3f(P1, a, b, c, Z[0][s][l], Z[1][s][l], Z[2][s][l]); }
    }
    if (has && q == PAR_G - 1) {
        double ta = a, tb = b, tc = c; const int s = slot(k);
        mv3f(P1, ta, tb, tc, Z[0][s][l], Z[1][s][l], Z[2][s][l]);
        GT[0][grp][l] = ta; GT[1][grp][l] = tb; GT[2][grp][l] = tc;
    }
    __syncthreads();
    double Sa = s0a, Sb = s0b, Sc = s0c;
    if (has) {
#pragma unroll 1
        for (int h = 0; h < grp; h++) mv3f(P8, Sa, Sb, Sc, GT[0][h][l], GT[1][h][l], GT[2][h][l]);
        if (q > 0) mv3f(PW + 9 * (q - 1), Sa, Sb, Sc, 0.0, 0.0, 0.0);
    }
    ea = Sa + a; eb = Sb + b; ec = Sc + c;
}
// Every segment has exactly SL samples (no per-sample predicates in the recurrences): the line is padded on the LEFT with
// pad = nseg SL - n virtual samples equal to x[0] -- under the replicate border the forward state before sample 0 is the steady
// state of a constant input x[0], which those samples leave unchanged; their outputs are never stored and count as zeros in the
// running sum.
template <int SL, int NS, int R>
__global__ __launch_bounds__(R * NS, (R * NS >= 1024 ? 4 : 4)) void k_rows_tol(RowsTolArgs A, int H, int W, int P, IIRPair cf, SegPow sp)
{
    constexpr int LPW = R;
    __shared__ double Z[3][NS][LPW];
    __shared__ double GT[3][NS / PAR_G + 1][LPW];
    __shared__ double Fin[3][LPW];
    __shared__ double PW[9 * (PAR_G + 1)];                 // M^(SL q), q = 1 .. PAR_G; M^(SL - 1)
    __shared__ double TOT[SL * NS];                        // product planes: the column totals of this plane's image (tot - E = running sum along y)
    // (8 rows per workgroup -- 64-byte half lines, the two halves of a line placed on one XCD -- was measured: 984 us against 650 with 16 rows)
    const int bx = blockIdx.x;
    const int t = threadIdx.x, l = t % LPW, g = t / LPW, pl = blockIdx.y;
    const int n = W;
    const int kind = PS_PICK(A, kind, pl), cs = PS_PICK(A, coef, pl);
    const bool hd = kind == 0 && A.dec != 0;               // the half-height blurred layer
    if (hd) { H >>= 1; P >>= 1; }
    if (bx * LPW >= H) return;                             // (the grid covers the full-height planes)
    const int y = bx * LPW + l;
    const bool valid = y < H;
    const int yc = valid ? y : H - 1;                      // idle lanes shadow the last row (reads only)
    double *plane = PS_PICK(A, p, pl) + (size_t)blockIdx.z * A.zs;
    const IIRCoef &k = cf.c[cs];
    const double a1 = k.a1, a2 = k.a2, a3 = k.a3, scale = k.scale;
    const int nseg = (n + SL - 1) / SL, pad = nseg * SL - n;
    const bool has = g < nseg;
    const int b = g * SL - pad;                            // column of the segment's first sample (negative inside the padding)
    const bool lastseg = g == nseg - 1;
    if (t < 9 * PAR_G) PW[t] = sp.P[cs][t / 9][t % 9];
    else if (t < 9 * PAR_G + 9) PW[t] = sp.Plast[cs][t - 9 * PAR_G];
    // addressing: uniform plane / column base (SGPRs) + one 32-bit byte offset per lane -- no per-load 64-bit address arithmetic in VGPRs
    const char *pb = (const char *)plane;
    const int bl = b < n - SL ? b : n - SL;               // idle segments (g >= nseg) shadow the last one (reads only)
    double x[SL];
    const double x0r = *(const double *)(pb + (unsigned)yc * 8u), xlr = *(const double *)(pb + ((size_t)(n - 1) * P) * 8 + (unsigned)yc * 8u);
    {
#pragma unroll
        for (int j = 0; j < SL; j++) {
            const int col = bl + j > 0 ? bl + j : 0;       // the padding of segment 0 shadows column 0 (replaced by x0 below)
            x[j] = __builtin_nontemporal_load((const double *)(pb + (unsigned)(((unsigned)col * (unsigned)P + (unsigned)yc) * 8u)));
        }
    }
    double x0 = x0r, xlast = xlr;
    if (kind == 1) {
        const double *tp = A.tot + ((size_t)blockIdx.z * 3 + (pl - A.nq0)) * A.tot_stride;
        for (int i = t; i < n; i += R * NS) TOT[i] = tp[i];
        __syncthreads();
        if (has) {
#pragma unroll
            for (int j = 0; j < SL; j++) x[j] = TOT[b + j >= 0 ? b + j : 0] - x[j];
        }
        x0 = TOT[0] - x0; xlast = TOT[n - 1] - xlast;
    } else __syncthreads();                                // PW
    if (g == 0) {
#pragma unroll
        for (int j = 0; j < SL; j++) if (j < pad) x[j] = x0;
    }
    if (A.dbg & 1) {
        char *wb = (char *)plane;
        if (valid && has && kind == 1) {
#pragma unroll
            for (int j = 0; j < SL; j++) if (j >= pad || g != 0) __builtin_nontemporal_store(x[j], (double *)(wb + (unsigned)(((unsigned)(b + j) * (unsigned)P + (unsigned)yc) * 8u)));
        }
        return;
    }
    const double uminus = x0 / k.inv1masum;
    // ---------------- forward: zero-state pass, fold, true pass ----------------
    {
        double w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll
        for (int j = 0; j < SL; j++) { const double tt = rt_step(x[j], a1, a2, a3, w1, w2, w3); w3 = w2; w2 = w1; w1 = tt; }
        if (has) { Z[0][g][l] = w1; Z[1][g][l] = w2; Z[2][g][l] = w3; }
    }
    __syncthreads();
    {
        double w1, w2, w3;
        rt_fold<NS, R>(sp, cs, PW, Z, GT, l, g, has, [](int kk) { return kk; }, uminus, uminus, uminus, w1, w2, w3);
#pragma unroll
        for (int j = 0; j < SL; j++) { const double tt = rt_step(x[j], a1, a2, a3, w1, w2, w3); w3 = w2; w2 = w1; w1 = tt; x[j] = tt; }
        if (lastseg) { Fin[0][l] = w1; Fin[1][l] = w2; Fin[2][l] = w3; }
    }
    __syncthreads();
    // ---------------- Triggs-Sdika right boundary ----------------
    const double f1 = Fin[0][l], f2 = Fin[1][l], f3 = Fin[2][l];
    const double uplus = xlast / k.inv1masum, vplus = uplus / k.inv1mbsum;
    const double d0 = f1 - uplus, d1 = f2 - uplus, d2 = f3 - uplus;
    const double vr0 = ((k.M[0] * d0 + k.M[1] * d1) + k.M[2] * d2) + vplus;
    const double vr1 = ((k.M[3] * d0 + k.M[4] * d1) + k.M[5] * d2) + vplus;
    const double vr2 = ((k.M[6] * d0 + k.M[7] * d1) + k.M[8] * d2) + vplus;
    // ---------------- backward (sample n-1, the last segment's last, is not part of the recurrence: v[n-1] = vr0) ----------------
    {
        double v1 = 0.0, v2 = 0.0, v3 = 0.0;
        if (!lastseg) { const double tt = rt_step(x[SL - 1], a1, a2, a3, v1, v2, v3); v3 = v2; v2 = v1; v1 = tt; }
#pragma unroll
        for (int j = SL - 2; j >= 0; j--) { const double tt = rt_step(x[j], a1, a2, a3, v1, v2, v3); v3 = v2; v2 = v1; v1 = tt; }
        if (has) { Z[0][g][l] = v1; Z[1][g][l] = v2; Z[2][g][l] = v3; }
    }
    __syncthreads();
    {
        double s0a = vr0, s0b = vr1, s0c = vr2;
        mv3f(sp.Plast[cs], s0a, s0b, s0c, Z[0][nseg - 1][l], Z[1][nseg - 1][l], Z[2][nseg - 1][l]);
        double v1, v2, v3;
        const int kq = nseg - 2 - g;
        const bool hasq = has && !lastseg;
        rt_fold<NS, R>(sp, cs, PW, Z, GT, l, hasq ? kq : 0, hasq, [nseg](int kk) { return nseg - 2 - kk; }, s0a, s0b, s0c, v1, v2, v3);
        if (lastseg) { v1 = vr0; v2 = vr1; v3 = vr2; x[SL - 1] = vr0 * scale; }
        else { const double tt = rt_step(x[SL - 1], a1, a2, a3, v1, v2, v3); v3 = v2; v2 = v1; v1 = tt; x[SL - 1] = tt * scale; }
#pragma unroll
        for (int j = SL - 2; j >= 0; j--) { const double tt = rt_step(x[j], a1, a2, a3, v1, v2, v3); v3 = v2; v2 = v1; v1 = tt; x[j] = tt * scale; }
    }
    __syncthreads();                                       // every thread is done with Z / GT: they are reused below
    double (*Zs)[LPW] = Z[0];                              // [NS][LPW]
    double (*Gs)[LPW] = GT[0];                             // [NS / PAR_G + 1][LPW]
    if (kind >= 1) {
        // ---------------- running sum along x (k_cum_seg's scheme on the register-resident values) ----------------
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < SL; j++) if (j < pad) x[j] = 0.0;
        }
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < SL; j++) acc = acc + x[j];
        if (has) Zs[g][l] = acc;
        __syncthreads();
        const int q = g % PAR_G, grp = g / PAR_G;
        double pre = 0.0;
        if (has) for (int i = 0; i < q; i++) pre = pre + Zs[grp * PAR_G + i][l];
        if (has && q == PAR_G - 1) Gs[grp][l] = pre + Zs[g][l];
        __syncthreads();
        double base = 0.0;
        if (has) for (int h = 0; h < grp; h++) base = base + Gs[h][l];
        acc = base + pre;
        char *wb = (char *)plane;
        if (valid && has && g != 0) {
#pragma unroll
            for (int j = 0; j < SL; j++) { acc = acc + x[j]; __builtin_nontemporal_store(acc, (double *)(wb + (unsigned)(((unsigned)(b + j) * (unsigned)P + (unsigned)yc) * 8u))); }
        } else if (valid && has) {
#pragma unroll
            for (int j = 0; j < SL; j++) { acc = acc + x[j]; if (j >= pad) __builtin_nontemporal_store(acc, (double *)(wb + (unsigned)(((unsigned)(b + j) * (unsigned)P + (unsigned)yc) * 8u))); }
        }
        return;
    }
    if (A.rz.dst == nullptr) {                             // blurred layer, plain store (k_resize follows)
        char *wb = (char *)plane;
        if (valid && has) {
#pragma unroll
            for (int j = 0; j < SL; j++) if (j >= pad || g != 0) *(double *)(wb + (unsigned)(((unsigned)(b + j) * (unsigned)P + (unsigned)yc) * 8u)) = x[j];
        }
        return;
    }
    // ---------------- imresize! of the blurred layer into the next level's layer (k_resize's arithmetic, exact 2:1 row ratio) ----------------
    if (has) Zs[g][l] = x[0];
    __syncthreads();
    if (!has) return;
    const double nxt = g + 1 < nseg ? Zs[g + 1][l] : 0.0;
    const int Wd = A.rz.Wd, Hd = A.rz.Hd;
    const double sx = (double)W / (double)Wd, ox = 1 - 0.5 - sx * (1 - 0.5);
    // first output column (1-based) whose left source sample (1-based: floor(c)) is >= max(b, 0) + 1
    const int b0 = b > 0 ? b : 0;
    int xo = (int)ceil(((double)(b0 + 1) - ox) / sx);
    if (xo < 1) xo = 1;
    while (xo > 1 && (int)floor(sx * (xo - 1) + ox) >= b0 + 1) xo--;
    while (xo <= Wd && (int)floor(sx * xo + ox) < b0 + 1) xo++;
    double *dbase = A.rz.dst + (size_t)blockIdx.z * A.zs + (size_t)(hd ? y : y >> 1);
    const bool writer = hd ? (valid && y < Hd) : (valid && (l & 1) == 0 && (y >> 1) < Hd);
#pragma unroll
    for (int j = 0; j < SL; j++) {                         // (the conditions are uniform over the 16 rows of a segment: the DPP pairs stay together)
        const double c = sx * xo + ox;
        int ixx = (int)floor(c);
        if (ixx > W - 1) ixx = W - 1;
        if (ixx < 1) ixx = 1;
        if (xo <= Wd && ixx == b + j + 1) {
            const double fx = c - ixx;
            const double bb = j + 1 < SL ? x[j + 1 < SL ? j + 1 : j] : nxt;
            const double h = (1 - fx) * x[j] + fx * bb;
            const double hn = dpp_pair_next(h);            // lanes 2k, 2k+1 <- lane 2k+1
            const double fy = 0.5;                          // r = 2 y' - 0.5: exact
            if (writer) dbase[(size_t)(xo - 1) * A.rz.Pd] = hd ? h : (1 - fy) * h + fy * hn;
            xo++;
        }
    }
}

// imgradients (KernelFactors.scharr, separable: derivative (-1,0,1)/2, smoothing
// (3,10,3)/16; first factor along dim 1 first) + the three gradient products.
// border 0: replicate (update!, pyramid.jl:98-103); 1: Fill(0) (ctor, pyramid.jl:51,59).
// One thread per row y of a strip of SCH_XC columns: the per-column terms (derivative and smoothing
// factor applied along dim 1) slide along x in registers, so each layer sample is fetched from HBM
// once per strip (+2 halo columns) instead of once per output column and XCD.
#define SCH_XC 16
struct ColTerm { double d, s; };
__device__ __forceinline__ ColTerm scharr_col(const double *L, int H, int W, int P, int y, int xx, int border)
{
    const double dk[3] = {-1.0 / 2, 0.0 / 2, 1.0 / 2}, sk[3] = {3.0 / 16, 10.0 / 16, 3.0 / 16};
    ColTerm t; t.d = 0.0; t.s = 0.0;
    if (border == 0) xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
    else if (xx < 0 || xx >= W) return t;
    const double *c0 = L + (size_t)xx * P;
    double a, b, c;
    b = c0[y];
    if (border == 0) { a = c0[y > 0 ? y - 1 : 0]; c = c0[y + 1 < H ? y + 1 : H - 1]; }
    else { a = y > 0 ? c0[y - 1] : 0.0; c = y + 1 < H ? c0[y + 1] : 0.0; }
    t.d += a * dk[0]; t.d += b * dk[1]; t.d += c * dk[2];
    t.s += a * sk[0]; t.s += b * sk[1]; t.s += c * sk[2];
    return t;
}
template <int XC>      // columns per thread: 16 for batches (each layer sample fetched 1.1 times), 4 for a single image (four times the waves: the kernel is latency-bound there)
__global__ __launch_bounds__(64) void k_scharr_products(LevelView v, int border, size_t zs, int squares)
{
    { const size_t z = (size_t)blockIdx.z * zs; v.L += z; v.Iy += z; v.Ix += z; v.Iyy += z; v.Ixx += z; v.Iyx += z; }
    const int H = v.H, W = v.W, P = v.P;
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= H) return;
    const int xs = blockIdx.y * XC, xe = min(W, xs + XC);
    const double dk[3] = {-1.0 / 2, 0.0 / 2, 1.0 / 2}, sk[3] = {3.0 / 16, 10.0 / 16, 3.0 / 16};
    ColTerm t0 = scharr_col(v.L, H, W, P, y, xs - 1, border), t1 = scharr_col(v.L, H, W, P, y, xs, border);
    for (int x = xs; x < xe; x++) {
        const ColTerm t2 = scharr_col(v.L, H, W, P, y, x + 1, border);
        double iy = 0.0, ix = 0.0;
        iy += t0.d * sk[0]; iy += t1.d * sk[1]; iy += t2.d * sk[2];
        ix += t0.s * dk[0]; ix += t1.s * dk[1]; ix += t2.s * dk[2];
        const size_t i = (size_t)y + (size_t)x * P;
        v.Iy[i] = iy; v.Ix[i] = ix;
        if (squares) { v.Iyy[i] = iy * iy; v.Ixx[i] = ix * ix; }      // 0: the checkpointed column kernel squares Iy / Ix itself
        v.Iyx[i] = iy * ix;
        t0 = t1; t1 = t2;
    }
}

// ImageTransformations.imresize!(dst, interpolate!(src, BSpline(Linear())))
__global__ __launch_bounds__(256) void k_resize(double *dst, int Hd, int Wd, int Pd, const double *src, int Hs, int Ws, int Ps, size_t zs)
{
    dst += (size_t)blockIdx.z * zs; src += (size_t)blockIdx.z * zs;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Hd * Wd) return;
    const int y = (int)(i % Hd) + 1, x = (int)(i / Hd) + 1;
    const double sy = (double)Hs / (double)Hd, sx = (double)Ws / (double)Wd;
    const double oy = 1 - 0.5 - sy * (1 - 0.5), ox = 1 - 0.5 - sx * (1 - 0.5);
    double r = sy * y + oy, c = sx * x + ox;
    if (sy < 1) r = r < 1 ? 1 : (r > Hs ? Hs : r);
    if (sx < 1) c = c < 1 ? 1 : (c > Ws ? Ws : c);
    int iy = (int)floor(r), ixx = (int)floor(c);
    if (iy > Hs - 1) iy = Hs - 1;
    if (ixx > Ws - 1) ixx = Ws - 1;
    if (iy < 1) iy = 1;
    if (ixx < 1) ixx = 1;
    const double fy = r - iy, fx = c - ixx;
    const double *p = src + (size_t)(iy - 1) + (size_t)(ixx - 1) * Ps;
    const int dy = Hs > 1 ? 1 : 0; const size_t dx = Ws > 1 ? (size_t)Ps : 0;
    const double r0 = (1 - fx) * p[0] + fx * p[dx];
    const double r1 = (1 - fx) * p[dy] + fx * p[dy + dx];
    dst[(size_t)(y - 1) + (size_t)(x - 1) * Pd] = (1 - fy) * r0 + fy * r1;
}

// Gray{Float64}.(img::Matrix{Gray{N0f8}}) of the KITTI reader (example/kitty/main.jl:39-41):
// Float64(::N0f8) = raw / 255 (FixedPointNumbers >= 0.8 divides; correctly rounded)
__global__ __launch_bounds__(256) void k_u8_to_f64(double *dst, const unsigned char *src, int H, int W, int P)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)H * W) dst[i % H + (i / H) * P] = (double)src[i] / 255.0;
}

// ingest: the dense H x W image z (its own device pointer) is copied into the pitched layer 0 of pyramid z
#define BATCH_MAX 128
struct ImgPtrs { const double *p[BATCH_MAX]; };
__global__ __launch_bounds__(256) void k_gather_images(ImgPtrs src, double *dst, int H, int W, int P, size_t zs)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)H * W) dst[(size_t)blockIdx.z * zs + i % H + (i / H) * P] = src.p[blockIdx.z][i];
}
// same from 8-bit frames: Float64(::N0f8) = raw / 255 on the way into the pitched layer
struct ImgPtrsU8 { const unsigned char *p[BATCH_MAX]; };
__global__ __launch_bounds__(256) void k_gather_images_u8(ImgPtrsU8 src, double *dst, int H, int W, int P, size_t zs)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)H * W) dst[(size_t)blockIdx.z * zs + i % H + (i / H) * P] = (double)src.p[blockIdx.z][i] / 255.0;
}
// device -> host staging of one plane: pitched -> dense
__global__ __launch_bounds__(256) void k_unpitch(double *dst, const double *src, int H, int W, int P)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)H * W) dst[i] = src[i % H + (i / H) * P];
}

__global__ __launch_bounds__(256) void k_fill(double *p, size_t n, double v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// smallest batch whose tolerance build (mode 3) takes the batch kernels (k_cols_fused<TOL> + k_rows_tol) instead of the segmented single-image ones
static inline int tol_batch_min_s()
{
    static const int v = [] { const char *e = getenv("SLAMHIP_TOL_BATCH_MIN_S"); return e ? atoi(e) : 4; }();
    return v;
}
static inline size_t ck_min_bytes()
{
    const char *e = getenv("SLAMHIP_CK_MIN_MB");          // test / tuning hook; read per build (graphs are keyed on it below)
    return e ? (size_t)atol(e) << 20 : (size_t)40 << 20;   // (96 MB until round 2: level 2 of 48+ streams and level 1 of 16+ now take the fused kernels too: +1.6 % / +5 % frames/s)
}
static inline dim3 lines_grid(int nlines, int nplanes, int S = 1) { return dim3((nlines + LINE_THREADS - 1) / LINE_THREADS, nplanes, S); }

static void make_view(slam_pyr *p)
{
    p->view.levels = p->levels;
    for (int l = 0; l < p->levels; l++) {
        LevelView &v = p->view.lv[l];
        v.L = p->plane(0, l); v.Iy = p->plane(1, l); v.Ix = p->plane(2, l);
        v.Iyy = p->plane(3, l); v.Ixx = p->plane(4, l); v.Iyx = p->plane(5, l);
        v.H = p->H[l]; v.W = p->W[l]; v.P = p->P[l];
    }
}

// NA() normaliser: the Fill(0)-filtered indicator of valid pixels, per level.
static int build_norm(slam_ctx *ctx, slam_pyr *p, double sigma)
{
    if (p->norm && p->norm_sigma == sigma) return SLAM_OK;
    if (!p->norm) HIP_TRY(ctx, hipMalloc((void **)&p->norm, ((size_t)p->off[p->levels] + (size_t)64 * p->P[0]) * 8));
    IIRPair cf; cf.c[0] = slam_iir_coef(sigma); cf.c[1] = cf.c[0];
    for (int l = 0; l + 1 < p->levels; l++) {
        double *N = p->norm + p->off[l];
        size_t n = (size_t)p->P[l] * p->W[l];
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, N, n, 1.0);
        PlaneSet ps = {}; ps.p[0] = N; ps.coef[0] = 0; ps.fill0[0] = 1; ps.n = 1;
        hipLaunchKernelGGL(k_iir_cols<3>, lines_grid(p->W[l], 1), dim3(LINE_THREADS), 0, ctx->stream, ps, (const double *)nullptr, p->H[l], p->W[l], p->P[l], cf);
        hipLaunchKernelGGL(k_iir_rows, lines_grid(p->H[l], 1), dim3(LINE_THREADS), 0, ctx->stream, ps, p->H[l], p->W[l], p->P[l], cf);
    }
    HIP_TRY(ctx, hipGetLastError());
    p->norm_sigma = sigma;
    return SLAM_OK;
}

// Enqueue the whole pyramid build on ctx->stream; layer 0 must already hold the image.
// Launch every kernel of one pyramid build.  `st` carries the dependent chain
// (gradients -> IIR dim 1 -> IIR dim 2 -> resize -> next level); the integral-image
// passes of level l only feed the LK kernel, so they run on `aux`, concurrently
// with level l+1 (fork after the row pass, one join at the end).  With
// aux == st everything is serial on one stream (profiling / fallback path).
static void mat3_pow(const IIRCoef &k, int e, double P[9])
{
    double M[9] = {k.a1, k.a2, k.a3, 1, 0, 0, 0, 1, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[9];
    auto mul = [&](const double *A, const double *B, double *C) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int m = 0; m < 3; m++) t += A[3 * i + m] * B[3 * m + j]; C[3 * i + j] = t; } };
    while (e > 0) {
        if (e & 1) { mul(R, M, T); memcpy(R, T, sizeof R); }
        mul(M, M, T); memcpy(M, T, sizeof M);
        e >>= 1;
    }
    memcpy(P, R, sizeof R);
}
static inline int seg_len(int n, int nseg_max) { int sl = (n + nseg_max - 1) / nseg_max; return sl < 4 ? 4 : sl; }
static void seg_pow(const IIRPair &cf, int n, int SL, SegPow &sp)
{
    const int nseg = (n + SL - 1) / SL, len_last = n - (nseg - 1) * SL;
    for (int c = 0; c < 2; c++) {
        for (int q = 1; q <= PAR_G; q++) mat3_pow(cf.c[c], SL * q, sp.P[c][q - 1]);
        mat3_pow(cf.c[c], len_last - 1, sp.Plast[c]);
    }
}

// k_rows_tol: samples per segment from a fixed menu (one instantiation each), the smallest that covers a row with RT_NS = 32 segments
// (512 threads: two workgroups per CU); rows wider than 40 x 32 samples (the 1920-wide level 0 of configs[4]) take 64 segments of 24 / 32
// samples (1024 threads, one workgroup per CU: 40+ samples per thread would spill at 128 registers).  Returns 0 when no variant fits.
static int rt_seg_len(int n, int *ns)
{
    static const int menu[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 40};
    *ns = RT_NS;
    for (int m : menu) if ((n + m - 1) / m <= RT_NS && n >= 2 * m) return m;
    *ns = 2 * RT_NS;
    for (int m : {24, 32}) if ((n + m - 1) / m <= 2 * RT_NS && n >= 2 * m) return m;
    return 0;
}
static void rt_seg_pow(const IIRPair &cf, int SL, SegPow &sp)     // M^(SL q), q = 1 .. PAR_G; M^(SL - 1): all segments are full (left padding)
{
    for (int c = 0; c < 2; c++) {
        for (int q = 1; q <= PAR_G; q++) mat3_pow(cf.c[c], SL * q, sp.P[c][q - 1]);
        mat3_pow(cf.c[c], SL - 1, sp.Plast[c]);
    }
}
// Launch every kernel of one pyramid build.  `st` carries the dependent chain
// (gradients -> IIR dim 1 -> IIR dim 2 -> resize -> next level); the integral-image
// passes of level l only feed the LK kernel, so they run on `aux`, concurrently
// with level l+1 (fork after the row pass, one join at the end).  With
// aux == st everything is serial on one stream (profiling / fallback path).
// mode 3 ("fast") swaps the sequential line kernels for the segmented ones.
// Where the kernels of one build go: straight onto a stream (profiling spans, SLAMHIP_NO_GRAPH), or into an EXPLICITLY constructed
// hipGraph -- hipGraphAddKernelNode with the dependencies of a two-lane DAG (lane 0: the dependent chain gradients -> dim 1 -> dim 2
// -> resize -> next level; lane 1: the integral-image passes of level l, which only feed the tracking kernels and run beside level
// l + 1).  Rounds 1-3 recorded the DAG with hipStreamBeginCapture on the caller's stream + a forked side stream: a capture is
// process-visible runtime state (another thread's or library's event queries and pinned allocations fail with
// hipErrorStreamCaptureUnsupported / hipErrorCapturedEvent while one is open, and a stream that was a capture origin left state
// behind when destroyed: DESIGN 6.6).  Building the graph node by node touches no stream at all.
enum { LN_MAIN = 0, LN_AUX = 1, LN_SIDE0 = 2, LN_COUNT = 2 + SLAM_MAX_LEVELS };   // LN_SIDE0 + l: the side branch of level l (single-image latency topology)
struct BuildSink {
    hipStream_t st = nullptr;                          // direct mode: every kernel on this stream, in program order
    hipGraph_t graph = nullptr;                        // graph mode
    bool forked = false;                               // graph mode: lanes >= 1 are branches of their own (else one chain)
    hipGraphNode_t last[LN_COUNT] = {}, fork_dep[LN_COUNT] = {};
    hipError_t err = hipSuccess;
    template <typename Tup, size_t... I> static void addr_of(Tup &t, void **out, std::index_sequence<I...>) { ((out[I] = (void *)&std::get<I>(t)), ...); }
    template <typename... KA, typename... A>
    void launch(void (*kern)(KA...), dim3 grid, dim3 block, size_t lds, int lane, A &&...a)
    {
        if (!graph) { hipLaunchKernelGGL(kern, grid, block, lds, st, std::forward<A>(a)...); return; }
        std::tuple<KA...> args(std::forward<A>(a)...);  // the kernel's own parameter types (the node copies them)
        void *ptrs[sizeof...(KA)];
        addr_of(args, ptrs, std::index_sequence_for<KA...>{});
        hipKernelNodeParams np = {};
        np.func = (void *)kern; np.gridDim = grid; np.blockDim = block; np.sharedMemBytes = (unsigned)lds; np.kernelParams = ptrs; np.extra = nullptr;
        const int ln = forked ? lane : LN_MAIN;
        hipGraphNode_t deps[2]; int nd = 0;
        if (last[ln]) deps[nd++] = last[ln];
        if (ln != LN_MAIN && fork_dep[ln] && fork_dep[ln] != last[ln]) deps[nd++] = fork_dep[ln];
        hipGraphNode_t node = nullptr;
        const hipError_t e = hipGraphAddKernelNode(&node, graph, nd ? deps : nullptr, (size_t)nd, &np);
        if (e != hipSuccess) { if (err == hipSuccess) err = e; return; }
        last[ln] = node;
        if (ln != LN_MAIN) fork_dep[ln] = nullptr;
    }
    // lane `ln`'s next kernel also waits for everything lane 0 has issued so far
    void fork(int ln = LN_AUX) { if (graph && forked) fork_dep[ln] = last[LN_MAIN]; }
    bool lanes() const { return graph && forked; }
};

// One level of the build for the images [z0, z0 + S) of a batch of Sall (kernel selection follows Sall: a sub-batch runs the kernels the
// whole batch would).
static void launch_level(slam_ctx *ctx, slam_pyr *p, int mode, const IIRPair &cf, BuildSink &B, bool spans, int S, int src_kind, bool target, int l, int z0, int Sall, bool fast)
{
    const size_t zs = p->zstride, zo = (size_t)z0 * zs;
    const int border_mode = (mode == 0) ? 1 : 0;
    {
        const int H = p->H[l], W = p->W[l], P = p->P[l];
        LevelView v = p->view.lv[l];
        v.L += zo; v.Iy += zo; v.Ix += zo; v.Iyy += zo; v.Ixx += zo; v.Iyx += zo;
        const bool has_next = l + 1 < p->levels;
        double *T = p->tmp + p->off[l] + zo;
        double *nextL = has_next ? p->view.lv[l + 1].L + zo : nullptr;
        if (target && l >= 1) {                                   // the layer chain only: blur (dim 1, dim 2) -> resize
            if (!has_next) return;
            PlaneSet pt = {}; pt.p[0] = T; pt.coef[0] = 0; pt.fill0[0] = (mode == 0); pt.nrm[0] = nullptr; pt.n = 1; pt.zs = zs;
            const bool ckr = (mode != 3 || Sall >= tol_batch_min_s()) && p->ck != nullptr && mode != 0 && W >= 64 && H >= 64 && (size_t)Sall * 4 * H * W * 8 >= ck_min_bytes();
            if (ckr) B.launch(k_iir_cols_ck, lines_grid(W, 1, S), dim3(LINE_THREADS), 0, LN_MAIN, pt, (const double *)v.L, H, W, P, cf, p->ck);
            else B.launch(k_iir_cols<2>, lines_grid(W, 1, S), dim3(LINE_THREADS), 0, LN_MAIN, pt, (const double *)v.L, H, W, P, cf);
            static const bool no_rr = getenv("SLAMHIP_NO_ROWS_RESIZE") != nullptr;
            const bool rr = ckr && (H & 1) == 0 && !no_rr;
            RowResize rzt = {};
            if (rr) { rzt.dst = nextL; rzt.Hd = p->H[l + 1]; rzt.Wd = p->W[l + 1]; rzt.Pd = p->P[l + 1]; }
            if (ckr) B.launch(k_iir_rows_ck, lines_grid(H, 1, S), dim3(LINE_THREADS), 0, LN_MAIN, pt, H, W, P, cf, p->ck, rzt);
            else B.launch(k_iir_rows, lines_grid(H, 1, S), dim3(LINE_THREADS), 0, LN_MAIN, pt, H, W, P, cf);
            if (!rr)
                B.launch(k_resize, dim3(((size_t)p->H[l + 1] * p->W[l + 1] + 255) / 256, 1, S), dim3(256), 0, LN_MAIN,
                                   nextL, p->H[l + 1], p->W[l + 1], p->P[l + 1], (const double *)T, H, W, P, zs);
            return;
        }
        // bandwidth-bound launches (many images x a large level) take the checkpointed IIR kernels; the column one squares
        // Iy / Ix itself, so the gradient kernel does not write (and the filter does not re-read) the Iyy / Ixx inputs
        const int np_ = has_next ? 4 : 3;
        static const bool no_ck_cols = getenv("SLAMHIP_NO_CK_COLS") != nullptr;     // tuning toggles are read once per process: a cached graph never disagrees with them
        const bool ck_cols = !fast && p->ck != nullptr && mode != 0 && H >= 64 && (size_t)Sall * np_ * H * W * 8 >= ck_min_bytes() && !no_ck_cols;
        static const bool no_sq = getenv("SLAMHIP_NO_SQ_FUSE") != nullptr;
        const bool fuse_sq = ck_cols && !no_sq;
        // ... and the fully fused dim-1 stage (Scharr + products + the four dim-1 recurrences straight from the layer)
        static const bool no_cols_fused = getenv("SLAMHIP_NO_COLS_FUSED") != nullptr;
        const bool cols_fused = fuse_sq && !no_cols_fused;
        // LATENCY TOPOLOGY of a single image (round 5): the only thing level l + 1 waits for is level l's blurred, resized layer.  The main
        // chain carries exactly that -- the dim-1 filter of the layer alone, the dim-2 filter (+ imresize!): 2-3 one-plane kernels per level;
        // the gradients, the three product planes and their integral images of level l are a side branch of their own (lane LN_SIDE0 + l),
        // which starts as soon as layer l exists and runs beside the chain and beside the other levels' branches.  Same kernels, same
        // arithmetic per plane (a launch over 4 planes and two launches over 1 + 3 planes do the same per-line work).
        // MEASURED (round 5, one 370 x 1226 image, builds back to back): tolerance mode 243 us per build with the topology, 218 without; bit-exact
        // 450 vs 369 -- the split launches and the extra graph branches cost more than the shorter chain saves (the runtime serialises
        // branch nodes onto few hardware queues), as the round-1 experiment on the bit-exact kernels had found.  Off unless SLAMHIP_TOPOLOGY=1.
        // A two-lane variant (layer chain of every level on the main lane, the gradient / product work of levels 0-1 on the aux lane, of the
        // coarser levels behind the chain) was measured too: one isolated build 181 vs 198 us under the profiler, but 178-220 vs 152 us back to
        // back -- every cross-queue dependency of the graph costs ~10 us, and consecutive builds no longer overlap.  Not kept.
        static const bool no_topo = getenv("SLAMHIP_TOPOLOGY") == nullptr;
        const bool topo = S == 1 && Sall == 1 && has_next && !cols_fused && B.lanes() && !no_topo;
        const int side = topo ? LN_SIDE0 + l : LN_MAIN;            // lane of the gradients / products (non-topology: the main chain)
        const int side_cum = topo ? LN_SIDE0 + l : LN_AUX;         // lane of the integral images
        if (topo) B.fork(side);
        auto emit_scharr = [&](int lane_) {
            static const int sch1 = [] { const char *e = getenv("SLAMHIP_SCHARR_XC1"); return e ? atoi(e) : 2; }();      // (measurement knob) columns per thread for a single image: 16 / 8 / 4 / 2 / 1 -> 216 / 200 / 192 / 189 / 189 us per tolerance-mode build
            if (!cols_fused && S == 1 && sch1 == 4) B.launch(k_scharr_products<4>, dim3((H + 63) / 64, (W + 3) / 4, S), dim3(64), 0, lane_, v, border_mode, zs, fuse_sq ? 0 : 1);
            else if (!cols_fused && S == 1 && sch1 == 2) B.launch(k_scharr_products<2>, dim3((H + 63) / 64, (W + 1) / 2, S), dim3(64), 0, lane_, v, border_mode, zs, fuse_sq ? 0 : 1);
            else if (!cols_fused && S == 1 && sch1 == 1) B.launch(k_scharr_products<1>, dim3((H + 63) / 64, W, S), dim3(64), 0, lane_, v, border_mode, zs, fuse_sq ? 0 : 1);
            else if (!cols_fused && S == 1 && sch1 == 8) B.launch(k_scharr_products<8>, dim3((H + 63) / 64, (W + 7) / 8, S), dim3(64), 0, lane_, v, border_mode, zs, fuse_sq ? 0 : 1);
            else if (!cols_fused) B.launch(k_scharr_products<SCH_XC>, dim3((H + 63) / 64, (W + SCH_XC - 1) / SCH_XC, S), dim3(64), 0, lane_, v, border_mode, zs, fuse_sq ? 0 : 1);
        };
        emit_scharr(side);
        // dim-1 IIR: [blur: L -> T], Iyy, Ixx, Iyx in place
        PlaneSet ps = {};
        int np = 0;
        if (has_next) { ps.p[np] = T; ps.coef[np] = 0; ps.fill0[np] = (mode == 0); ps.nrm[np] = (mode == 0) ? p->norm + p->off[l] : nullptr; np++; }
        ps.p[np] = v.Iyy; ps.coef[np] = 1; ps.sq[np] = fuse_sq ? v.Iy : nullptr; np++;
        ps.p[np] = v.Ixx; ps.coef[np] = 1; ps.sq[np] = fuse_sq ? v.Ix : nullptr; np++;
        ps.p[np] = v.Iyx; ps.coef[np] = 1; np++;
        ps.n = np; ps.zs = zs;
        const double *src0 = has_next ? (const double *)v.L : (const double *)nullptr;
        PlaneSet psT = {}, psQ = {};                                // topology: the blurred layer alone / the three product planes
        if (topo) {
            psT.p[0] = ps.p[0]; psT.coef[0] = ps.coef[0]; psT.fill0[0] = ps.fill0[0]; psT.nrm[0] = ps.nrm[0]; psT.n = 1; psT.zs = zs;
            for (int q = 0; q < 3; q++) { psQ.p[q] = ps.p[q + 1]; psQ.coef[q] = ps.coef[q + 1]; psQ.fill0[q] = ps.fill0[q + 1]; psQ.nrm[q] = ps.nrm[q + 1]; psQ.sq[q] = ps.sq[q + 1]; }
            psQ.n = 3; psQ.zs = zs;
        }
        PlaneSet pc = {};
        pc.p[0] = v.Iyy; pc.p[1] = v.Ixx; pc.p[2] = v.Iyx; pc.n = 3; pc.zs = zs;
        if (fast) {
            const int slc = seg_len(H, PAR_T / 8), slr = seg_len(W, PAR_T / 8);
            SegPow spc, spr;
            seg_pow(cf, H, slc, spc); seg_pow(cf, W, slr, spr);
            const dim3 gc((W + 7) / 8, np, S), gr((H + 7) / 8, np, S), gc3((W + 7) / 8, 3, S), gr3((H + 7) / 8, 3, S);
            // columns of up to 512 samples: 32 segments of <= 16 samples per column in 256-thread workgroups (see k_iir_seg); SLAMHIP_SEG_WIDE=1: the 128-segment variant
            static const bool seg_wide = getenv("SLAMHIP_SEG_WIDE") != nullptr;
            const int slc32 = seg_len(H, 32);
            const bool c32 = slc32 <= PAR_SLMAX && !seg_wide;
            SegPow spc32;
            if (c32) seg_pow(cf, H, slc32, spc32);
            int rns = RT_NS;
            const int slt = rt_seg_len(W, &rns);
            static const bool no_rt1 = getenv("SLAMHIP_NO_ROWS_TOL_SINGLE") != nullptr;
            const bool rt1 = slt > 0 && rns == RT_NS && !no_rt1;      // the dim-2 stage runs through k_rows_tol (below)
            // ... then the product planes take their running sum along y inside the dim-1 kernel (SLAMHIP_NO_SEG_CUM=1: a k_cum_seg launch of their own)
            static const bool no_seg_cum = getenv("SLAMHIP_NO_SEG_CUM") != nullptr;
            const bool seg_cum = rt1 && !topo && !no_seg_cum;
            const int cmask = seg_cum ? (has_next ? 0xE : 0x7) : 0;
            if (topo) {
                B.launch(k_iir_seg<true>, dim3((W + 7) / 8, 1, S), dim3(PAR_T), 0, LN_MAIN, psT, src0, H, W, P, cf, spc, slc, 0);
                B.launch(k_iir_seg<true>, gc3, dim3(PAR_T), 0, side, psQ, (const double *)nullptr, H, W, P, cf, spc, slc, 0);
            }
            else if (c32) B.launch((k_iir_seg<true, 256, 8>), gc, dim3(256), 0, LN_MAIN, ps, src0, H, W, P, cf, spc32, slc32, cmask);
            else B.launch(k_iir_seg<true>, gc, dim3(PAR_T), 0, LN_MAIN, ps, src0, H, W, P, cf, spc, slc, cmask);
            // round 4: the dim-2 stage of a single image through k_rows_tol as well -- the blurred layer (filter along x + imresize!, one launch on
            // the main chain instead of k_iir_seg + k_resize) and, on the branch, the product planes (running sum along y first: the two
            // directions commute; then filter + running sum along x in one launch instead of k_iir_seg + k_cum_seg): 5 launches per level, 3 of
            // them on the chain the next level waits for (6 / 4 before)
            if (rt1) {
                SegPow spt; rt_seg_pow(cf, slt, spt);
                auto rows_tol = [&](RowsTolArgs &ra, int nplanes, int lane_) {
                    const dim3 g2((H + RT_R - 1) / RT_R, nplanes, S), b2(RT_R * RT_NS);
#define RT_GO1(SLV) B.launch((k_rows_tol<SLV, RT_NS, RT_R>), g2, b2, 0, lane_, ra, H, W, P, cf, spt)
                    switch (slt) { case 4: RT_GO1(4); break; case 6: RT_GO1(6); break; case 8: RT_GO1(8); break; case 10: RT_GO1(10); break; case 12: RT_GO1(12); break;
                                   case 16: RT_GO1(16); break; case 20: RT_GO1(20); break; case 24: RT_GO1(24); break; case 32: RT_GO1(32); break; default: RT_GO1(40); break; }
#undef RT_GO1
                };
                if (!topo) B.fork();
                if (has_next) {
                    RowsTolArgs rt = {};
                    rt.p[0] = T; rt.coef[0] = 0; rt.kind[0] = 0; rt.n = 1; rt.nq0 = 1; rt.zs = zs;
                    const bool rzf = (H & 1) == 0;
                    if (rzf) { rt.rz.dst = nextL; rt.rz.Hd = p->H[l + 1]; rt.rz.Wd = p->W[l + 1]; rt.rz.Pd = p->P[l + 1]; }
                    if (spans) { ProfScope span(ctx, "k_iir_rows"); rows_tol(rt, 1, LN_MAIN); } else rows_tol(rt, 1, LN_MAIN);
                    if (!rzf)
                        B.launch(k_resize, dim3(((size_t)p->H[l + 1] * p->W[l + 1] + 255) / 256, 1, S), dim3(256), 0, LN_MAIN,
                                           nextL, p->H[l + 1], p->W[l + 1], p->P[l + 1], (const double *)T, H, W, P, zs);
                }
                if (seg_cum) {}
                else if (c32) B.launch((k_cum_seg<true, 256, 8>), gc3, dim3(256), 0, side_cum, pc, H, W, P, slc32);
                else B.launch(k_cum_seg<true>, gc3, dim3(PAR_T), 0, side_cum, pc, H, W, P, slc);
                RowsTolArgs rq = {};
                rq.p[0] = v.Iyy; rq.p[1] = v.Ixx; rq.p[2] = v.Iyx;
                for (int q = 0; q < 3; q++) { rq.coef[q] = 1; rq.kind[q] = 2; }
                rq.n = 3; rq.nq0 = 0; rq.zs = zs;
                rows_tol(rq, 3, side_cum);
                return;
            }
            if (topo) {
                B.launch(k_iir_seg<false>, dim3((H + 7) / 8, 1, S), dim3(PAR_T), 0, LN_MAIN, psT, (const double *)nullptr, H, W, P, cf, spr, slr, 0);
                B.launch(k_iir_seg<false>, gr3, dim3(PAR_T), 0, side, psQ, (const double *)nullptr, H, W, P, cf, spr, slr, 0);
            }
            else if (spans) { ProfScope span(ctx, "k_iir_rows");
                B.launch(k_iir_seg<false>, gr, dim3(PAR_T), 0, LN_MAIN, ps, (const double *)nullptr, H, W, P, cf, spr, slr, 0); }
            else B.launch(k_iir_seg<false>, gr, dim3(PAR_T), 0, LN_MAIN, ps, (const double *)nullptr, H, W, P, cf, spr, slr, 0);
            if (!topo) B.fork();
            if (has_next)
                B.launch(k_resize, dim3(((size_t)p->H[l + 1] * p->W[l + 1] + 255) / 256, 1, S), dim3(256), 0, LN_MAIN,
                                   nextL, p->H[l + 1], p->W[l + 1], p->P[l + 1], (const double *)T, H, W, P, zs);
            if (c32) B.launch((k_cum_seg<true, 256, 8>), gc3, dim3(256), 0, side_cum, pc, H, W, P, slc32);
            else B.launch(k_cum_seg<true>, gc3, dim3(PAR_T), 0, side_cum, pc, H, W, P, slc);
            B.launch(k_cum_seg<false>, gr3, dim3(PAR_T), 0, side_cum, pc, H, W, P, slr);
            return;
        }
        // tolerance build of a batch (mode 3, S >= 4): the dim-1 stage leaves the product planes as suffix sums along y, ONE row kernel
        // finishes the level (dim-2 filter + running sum along x + imresize!): 1 R + 1 W per plane instead of 11 R + 6.25 W
        static const bool no_tol_batch = getenv("SLAMHIP_NO_TOL_BATCH") != nullptr;
        int rt_ns = RT_NS;
        const int slr_t = rt_seg_len(W, &rt_ns);
        const bool tolb = mode == 3 && Sall >= tol_batch_min_s() && cols_fused && p->alloc->tot != nullptr && slr_t > 0 && !no_tol_batch;
        if (cols_fused) {
            ColsFusedArgs ca;
            ca.L = v.L; ca.T = has_next ? T : nullptr; ca.Iy = v.Iy; ca.Ix = v.Ix; ca.Qyy = v.Iyy; ca.Qxx = v.Ixx; ca.Qyx = v.Iyx;
            ca.H = H; ca.W = W; ca.P = P; ca.zs = zs;
            ca.src_kind = l == 0 ? src_kind : 0; ca.srctab = (const void *const *)p->alloc->srctab + z0;
            size_t toff = 0;
            for (int q = 0; q < l; q++) toff += (size_t)3 * Sall * p->W[q];
            toff += (size_t)3 * z0 * W;
            ca.tot = tolb ? p->alloc->tot + toff : nullptr; ca.tot_stride = W;
            static const bool no_dec = getenv("SLAMHIP_NO_TOL_DEC") != nullptr;
            ca.dec = (tolb && has_next && (H & 1) == 0 && (P & 31) == 0 && !no_dec) ? 1 : 0;
            if (tolb && ca.dec) B.launch(k_cols_fused<true, true>, dim3((W + CF4_COLS - 1) / CF4_COLS, 1, S), dim3(256), 0, LN_MAIN, ca, cf, p->ck);
            else if (tolb) B.launch(k_cols_fused<true, false>, dim3((W + CF4_COLS - 1) / CF4_COLS, 1, S), dim3(256), 0, LN_MAIN, ca, cf, p->ck);
            else B.launch(k_cols_fused<false, false>, dim3((W + CF4_COLS - 1) / CF4_COLS, 1, S), dim3(256), 0, LN_MAIN, ca, cf, p->ck);
            if (tolb) {
                RowsTolArgs ra = {};
                int nr = 0;
                if (has_next) { ra.p[nr] = T; ra.coef[nr] = 0; ra.kind[nr] = 0; nr++; }
                ra.nq0 = nr;
                ra.p[nr] = v.Iyy; ra.coef[nr] = 1; ra.kind[nr] = 1; nr++;
                ra.p[nr] = v.Ixx; ra.coef[nr] = 1; ra.kind[nr] = 1; nr++;
                ra.p[nr] = v.Iyx; ra.coef[nr] = 1; ra.kind[nr] = 1; nr++;
                ra.n = nr; ra.zs = zs; ra.tot = ca.tot; ra.tot_stride = W;
                const bool rzf = has_next && (H & 1) == 0;
                ra.dec = ca.dec;
                if (rzf) { ra.rz.dst = nextL; ra.rz.Hd = p->H[l + 1]; ra.rz.Wd = p->W[l + 1]; ra.rz.Pd = p->P[l + 1]; }
                SegPow spr; rt_seg_pow(cf, slr_t, spr);
                static const int rt_dbg = getenv("SLAMHIP_RT_DBG") ? atoi(getenv("SLAMHIP_RT_DBG")) : 0; ra.dbg = rt_dbg;
                const dim3 gr((H + RT_R - 1) / RT_R, nr, S), bd(RT_R * rt_ns);
                auto go = [&]() {
                    if (rt_ns != RT_NS) {                          // wide rows: 64 segments
                        if (slr_t == 24) B.launch((k_rows_tol<24, 2 * RT_NS, RT_R>), gr, bd, 0, LN_MAIN, ra, H, W, P, cf, spr);
                        else B.launch((k_rows_tol<32, 2 * RT_NS, RT_R>), gr, bd, 0, LN_MAIN, ra, H, W, P, cf, spr);
                        return;
                    }
#define RT_GO(SLV) B.launch((k_rows_tol<SLV, RT_NS, RT_R>), gr, bd, 0, LN_MAIN, ra, H, W, P, cf, spr)
                    switch (slr_t) { case 4: RT_GO(4); break; case 6: RT_GO(6); break; case 8: RT_GO(8); break; case 10: RT_GO(10); break; case 12: RT_GO(12); break;
                                     case 16: RT_GO(16); break; case 20: RT_GO(20); break; case 24: RT_GO(24); break; case 32: RT_GO(32); break; default: RT_GO(40); break; }
#undef RT_GO
                };
                if (spans) { ProfScope span(ctx, "k_iir_rows"); go(); } else go();
                if (has_next && !rzf)
                    B.launch(k_resize, dim3(((size_t)p->H[l + 1] * p->W[l + 1] + 255) / 256, 1, S), dim3(256), 0, LN_MAIN,
                                       nextL, p->H[l + 1], p->W[l + 1], p->P[l + 1], (const double *)T, H, W, P, zs);
                return;
            }
        }
        else if (ck_cols) B.launch(k_iir_cols_ck, lines_grid(W, np, S), dim3(LINE_THREADS), 0, LN_MAIN, ps, src0, H, W, P, cf, p->ck);
        else if (topo) {
            B.launch(k_iir_cols<3>, lines_grid(W, 1, S), dim3(LINE_THREADS), 0, LN_MAIN, psT, src0, H, W, P, cf);
            B.launch(k_iir_cols<3>, lines_grid(W, 3, S), dim3(LINE_THREADS), 0, side, psQ, (const double *)nullptr, H, W, P, cf);
        }
        else if (S == 1) B.launch(k_iir_cols<3>, lines_grid(W, np, S), dim3(LINE_THREADS), 0, LN_MAIN, ps, src0, H, W, P, cf);
        else B.launch(k_iir_cols<2>, lines_grid(W, np, S), dim3(LINE_THREADS), 0, LN_MAIN, ps, src0, H, W, P, cf);
        // bandwidth-bound launches (many images x a large level) take the checkpointed row kernel (2R+1W instead of 2R+2W)
        const bool ck_rows = p->ck != nullptr && mode != 0 && W >= 64 && (size_t)Sall * np * H * W * 8 >= ck_min_bytes();
        // ... with the imresize! into the next level fused into the blurred layer's backward sweep when the row ratio is exactly 2:1
        static const bool no_rows_resize = getenv("SLAMHIP_NO_ROWS_RESIZE") != nullptr;
        const bool rows_resize = ck_rows && has_next && (H & 1) == 0 && !no_rows_resize;
        RowResize rz = {};
        if (rows_resize) { rz.dst = nextL; rz.Hd = p->H[l + 1]; rz.Wd = p->W[l + 1]; rz.Pd = p->P[l + 1]; }
        if (topo) {
            B.launch(k_iir_rows, lines_grid(H, 1, S), dim3(LINE_THREADS), 0, LN_MAIN, psT, H, W, P, cf);
            B.launch(k_iir_rows, lines_grid(H, 3, S), dim3(LINE_THREADS), 0, side, psQ, H, W, P, cf);
        }
        else if (spans) { ProfScope span(ctx, "k_iir_rows");
            if (ck_rows) B.launch(k_iir_rows_ck, lines_grid(H, np, S), dim3(LINE_THREADS), 0, LN_MAIN, ps, H, W, P, cf, p->ck, rz);
            else B.launch(k_iir_rows, lines_grid(H, np, S), dim3(LINE_THREADS), 0, LN_MAIN, ps, H, W, P, cf); }
        else if (ck_rows) B.launch(k_iir_rows_ck, lines_grid(H, np, S), dim3(LINE_THREADS), 0, LN_MAIN, ps, H, W, P, cf, p->ck, rz);
        else B.launch(k_iir_rows, lines_grid(H, np, S), dim3(LINE_THREADS), 0, LN_MAIN, ps, H, W, P, cf);
        if (!topo) B.fork();
        if (has_next && !rows_resize)
            B.launch(k_resize, dim3(((size_t)p->H[l + 1] * p->W[l + 1] + 255) / 256, 1, S), dim3(256), 0, LN_MAIN,
                               nextL, p->H[l + 1], p->W[l + 1], p->P[l + 1], (const double *)T, H, W, P, zs);
        static const bool no_fused_cum = getenv("SLAMHIP_NO_FUSED_CUM") != nullptr;
        const int cf_bands = (H + 63) / 64;
        const int cf_seg = cf_bands <= CF_MAXW ? 1 : (cf_bands + CF_SEGW - 1) / CF_SEGW;      // taller planes: row segments of CF_SEGW bands, one workgroup each, chained through p->alloc->xc
        if (Sall >= 8 && !no_fused_cum && (cf_seg == 1 || (p->alloc->xc != nullptr && cf_seg <= p->alloc->xseg))) {      // batches: one-pass integral image
            const int nw = cf_seg == 1 ? cf_bands : CF_SEGW;
            const size_t lds = ((size_t)nw * CF_W * CF_LS + (size_t)nw * 2 * CF_W) * sizeof(double);
            const size_t xo = (size_t)z0 * 3 * (size_t)(cf_seg > 1 ? cf_seg - 1 : 1);
            if (cf_seg == 1) B.launch(k_cum_fused<false>, dim3(1, 3, S), dim3(nw * 64), lds, LN_AUX, pc, H, W, P, (double *)nullptr, (int *)nullptr);
            else B.launch(k_cum_fused<true>, dim3(cf_seg, 3, S), dim3(nw * 64), lds, LN_AUX, pc, H, W, P, p->alloc->xc + xo * W, p->alloc->xf + xo);
            return;
        }
        if (S == 1) B.launch(k_cum_cols<3>, lines_grid(W, 3, S), dim3(LINE_THREADS), 0, side_cum, pc, H, W, P);
        else B.launch(k_cum_cols<2>, lines_grid(W, 3, S), dim3(LINE_THREADS), 0, LN_AUX, pc, H, W, P);
        B.launch(k_cum_rows, lines_grid(H, 3, S), dim3(LINE_THREADS), 0, S == 1 ? side_cum : LN_AUX, pc, H, W, P);
    }
}


// Launch every kernel of one pyramid build (see BuildSink above for where they go).  A large batch is built in SUB-BATCHES of the big
// levels: the 3.5 intermediate planes a level's dim-1 stage leaves behind (12.7 MB per 370 x 1226 image) are re-read by its dim-2 stage
// while they still sit in the 256 MiB Infinity Cache, instead of after the whole batch's 1.6 GB have gone through it.
static void launch_build(slam_ctx *ctx, slam_pyr *p, int mode, const IIRPair &cf, BuildSink &B, bool spans, int S = 1, int src_kind = 0, bool target = false)
{
    bool fast = mode == 3;
    if (fast && (seg_len(p->H[0], PAR_T / 8) > PAR_SLMAX || seg_len(p->W[0], PAR_T / 8) > PAR_SLMAX)) fast = false;   // lines > 2048 samples: exact kernels
    if (fast && S >= tol_batch_min_s()) fast = false;   // the segmented kernels buy latency for ONE image; with several images per launch the exact kernels are faster (and trivially within the tolerance)
    static const int chunk_mb = [] { const char *v = getenv("SLAMHIP_PYR_CHUNK_MB"); return v ? atoi(v) : 0; }();      // intermediate planes of a sub-batch, MB (0: no sub-batches)
    for (int l = 0; l < p->levels; l++) {
        int C = S;
        if (chunk_mb > 0 && S >= 8 && mode != 0) {
            const size_t per_image = (size_t)4 * p->P[l] * p->W[l] * 8;                  // blurred layer + three product planes between the two stages
            const size_t want = (size_t)chunk_mb << 20;
            if (per_image * S > 2 * want) { C = (int)std::max<size_t>(4, want / per_image); C = std::min(C, S); }
        }
        for (int z0 = 0; z0 < S; z0 += C) launch_level(ctx, p, mode, cf, B, spans, std::min(C, S - z0), src_kind, target, l, z0, S, fast);
    }
}

// level 0 of an S-image build runs k_cols_fused (which can ingest the source images itself): launch_build's conditions
static bool level0_fused(const slam_pyr *p, int mode, int S)
{
    static const bool off = getenv("SLAMHIP_NO_COLS_FUSED") != nullptr || getenv("SLAMHIP_NO_SQ_FUSE") != nullptr || getenv("SLAMHIP_NO_CK_COLS") != nullptr ||
                            getenv("SLAMHIP_NO_FUSED_INGEST") != nullptr;
    if (off || mode == 0 || p->ck == nullptr || p->alloc->srctab == nullptr || p->H[0] < 64) return false;
    if (mode == 3 && S < tol_batch_min_s() && !(seg_len(p->H[0], PAR_T / 8) > PAR_SLMAX || seg_len(p->W[0], PAR_T / 8) > PAR_SLMAX)) return false;   // the segmented kernels
    const int np_ = p->levels > 1 ? 4 : 3;
    return (size_t)S * np_ * p->H[0] * p->W[0] * 8 >= ck_min_bytes();
}
__global__ void k_set_ptrs(const void **tab, ImgPtrs src)
{
    if (threadIdx.x < BATCH_MAX) tab[threadIdx.x] = src.p[threadIdx.x];
}

// Enqueue the whole pyramid build on ctx->stream; layer 0 must already hold the image.
// Normal path: one hipGraph replay (captured once per (mode, sigma): two-stream
// fork/join DAG, ~25 kernel nodes) -> one host launch instead of ~25.  With
// profiling spans enabled (or SLAMHIP_NO_GRAPH=1) the same kernels are launched
// directly, serially, so that per-kernel hipEvent spans are meaningful.
static int enqueue_build(slam_ctx *ctx, slam_pyr *p, int mode_flags, double sigma, int S = 1, int src_kind = 0)
{
    // SLAM_PYR_TARGET_ONLY: the pyramid will only be the TARGET of matches (the mapper's right pyramid, mapper.jl:51-56): optflow!
    // samples a target's layers at every level, and fb_tracking!'s backward pass (pyramid_levels = 0, tracker.jl:51-57) takes it as
    // template at level 1 only -- the gradient, product and integral planes of the coarser levels are never read.  Level 0 is
    // built in full, levels >= 1 get their blurred / resized layers and nothing else.
    const bool target = (mode_flags & SLAM_PYR_TARGET_ONLY) != 0;
    // SLAM_PYR_CHAIN: the build as ONE chain on the calling context's stream (no forked integral-image branch).  A single build takes
    // ~15 % longer, but it occupies one hardware queue instead of two and its replay costs the host a third (31 vs 84 us): the
    // choice when several builds of consecutive frames are kept in flight on several contexts (3 in flight: 170 us per build, 376 forked)
    static const bool linear_env = getenv("SLAMHIP_LINEAR_GRAPH") != nullptr;
    const bool chain = (mode_flags & SLAM_PYR_CHAIN) != 0 || linear_env;
    const int mode = mode_flags & ~SLAM_PYR_FLAGS;
    p->target_only = target;                                      // (the other members of a batch are marked by the batch entry points)
    p->tol_planes = mode == 3;
    if (target) src_kind |= 16;                                   // part of the graph key
    if (chain) src_kind |= 32;
    IIRPair cf; cf.c[0] = slam_iir_coef(sigma); cf.c[1] = slam_iir_coef(4.0);   // lucas_kanade.jl:112
    if (mode == 0) { int rc = build_norm(ctx, p, sigma); if (rc) return rc; }
    hipStream_t st = ctx->stream;
    static const bool no_graph = getenv("SLAMHIP_NO_GRAPH") != nullptr;
    if (ctx->prof_on || no_graph) {
        ProfScope span_all(ctx, "pyr_update");
        BuildSink B; B.st = st;
        launch_build(ctx, p, mode, cf, B, ctx->prof_on, S, src_kind & 15, target);
        HIP_TRY(ctx, hipGetLastError());
        return SLAM_OK;
    }
    hipGraphExec_t exec = nullptr;
    const size_t ckmin = ck_min_bytes();
    {
        std::lock_guard<std::mutex> lk(p->alloc->graph_mu);       // two contexts (threads) may request the first build of one pyramid at once
        for (auto &g : p->graphs) if (g.mode == mode && g.sigma == sigma && g.S == S && g.ckmin == ckmin && g.src_kind == src_kind) exec = g.exec;
        if (!exec && !p->graph_failed) {
            hipGraph_t graph = nullptr;
            hipError_t e = hipGraphCreate(&graph, 0);
            if (e == hipSuccess) {
                BuildSink B; B.graph = graph; B.forked = !chain;
                launch_build(ctx, p, mode, cf, B, false, S, src_kind & 15, target);
                e = B.err;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { (void)hipGetLastError(); p->graph_failed = true; exec = nullptr; }
            else { slam_pyr::Graph g; g.mode = mode; g.sigma = sigma; g.S = S; g.ckmin = ckmin; g.src_kind = src_kind; g.exec = exec; p->graphs.push_back(g); }
        }
        // (launched under the lock: one executable graph is not launched from two threads at the same moment)
        if (exec) HIP_TRY(ctx, hipGraphLaunch(exec, st));
    }
    if (!exec) { BuildSink B; B.st = st; launch_build(ctx, p, mode, cf, B, false, S, src_kind & 15, target); }
    HIP_TRY(ctx, hipGetLastError());
    return SLAM_OK;
}

// dense column-major H x W image (device) -> pitched layer 0
static void ingest_dense(slam_ctx *ctx, slam_pyr *p, const double *image_dev)
{
    ImgPtrs ip = {}; ip.p[0] = image_dev;
    const size_t n = (size_t)p->H[0] * p->W[0];
    hipLaunchKernelGGL(k_gather_images, dim3((n + 255) / 256, 1, 1), dim3(256), 0, ctx->stream, ip, p->plane(0, 0), p->H[0], p->W[0], p->P[0], (size_t)0);
}

extern "C" {

// One allocation holds S pyramids back to back: per image 6 planes + the blur scratch plane
// (zstride = 7 * sum_l P_l W_l doubles, P_l = H_l rounded up to 16).  S = 1 is the ordinary single pyramid.
static int pyr_create_n(slam_ctx *ctx, int H, int W, int pyramid_levels, int S, slam_pyr **out)
{
    ARG_TRY(ctx, ctx != nullptr && out != nullptr);
    ARG_TRY(ctx, H >= 4 && W >= 4 && pyramid_levels >= 0 && pyramid_levels + 1 <= SLAM_MAX_LEVELS && S >= 1 && S <= BATCH_MAX);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int Hs[SLAM_MAX_LEVELS], Ws[SLAM_MAX_LEVELS], Ps[SLAM_MAX_LEVELS]; int64_t off[SLAM_MAX_LEVELS + 1];
    int64_t o = 0; int h = H, w = W;
    const int levels = pyramid_levels + 1;
    for (int l = 0; l < levels; l++) {
        if (h < 4 || w < 4) return slam_fail(ctx, SLAM_ERR_ARG, "slam_pyr_create: level %d is %dx%d, too small for the IIR kernel (needs > 3)", l, h, w);
        Hs[l] = h; Ws[l] = w; Ps[l] = (h + 15) & ~15; off[l] = o; o += (int64_t)Ps[l] * w;
        h = (h + 1) / 2; w = (w + 1) / 2;                         // ceil(size / 2)
    }
    off[levels] = o;
    slam_pyr::Alloc *al = new slam_pyr::Alloc();
    const size_t tail = (size_t)64 * Ps[0];                                  // column tiles of the last workgroup read up to 63 columns past W
    hipError_t e = hipMalloc((void **)&al->base, ((size_t)o * 7 * S + tail) * 8);
    if (e != hipSuccess) { delete al; return slam_fail(ctx, SLAM_ERR_HIP, "slam_pyr_create: hipMalloc: %s", hipGetErrorString(e)); }
    // pitch padding rows are never used; keep them defined.  Creation is not on the hot path: the memset is complete before the
    // handle exists, so a build enqueued through ANOTHER context's (non-blocking) stream can never be overtaken by it.
    e = hipMemsetAsync(al->base, 0, ((size_t)o * 7 * S + tail) * 8, ctx->stream);
    if (e == hipSuccess) e = slam_stream_wait(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(al->base); delete al; return slam_fail(ctx, SLAM_ERR_HIP, "slam_pyr_create: memset: %s", hipGetErrorString(e)); }
    al->refs = S;
    if (S > 1 && hipMalloc((void **)&al->srctab, BATCH_MAX * sizeof(void *)) != hipSuccess) { (void)hipGetLastError(); al->srctab = nullptr; }
    // once per creation, outside any stream capture: k_cum_fused needs the > 64 KB dynamic-LDS opt-in
    {
        hipError_t ea = hipFuncSetAttribute((const void *)k_cum_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (ea == hipSuccess) ea = hipFuncSetAttribute((const void *)k_cum_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (ea != hipSuccess) { (void)hipFree(al->base); delete al; return slam_fail(ctx, SLAM_ERR_HIP, "slam_pyr_create: hipFuncSetAttribute(k_cum_fused): %s", hipGetErrorString(ea)); }
    }
    double *ckbuf = nullptr;
    if (S > 1) {   // checkpoint scratch of k_iir_rows_ck: (blocks x 3) doubles per line of the widest launch (level 0, 4 planes)
        const size_t lines = (size_t)S * 4 * (((size_t)Hs[0] + LINE_THREADS - 1) / LINE_THREADS) * LINE_THREADS;
        const size_t nbk = ((size_t)Ws[0] + CK_B - 1) / CK_B;
        const size_t lines_c = (size_t)S * 4 * (((size_t)Ws[0] + CF4_COLS - 1) / CF4_COLS) * LINE_THREADS;      // column pass: lines = columns (62 own columns per wave in k_cols_fused)
        const size_t nbk_c = ((size_t)Hs[0] + 31) / 32 + 1;
        const size_t need = lines * nbk > lines_c * nbk_c ? lines * nbk : lines_c * nbk_c;
        if (hipMalloc((void **)&ckbuf, need * 3 * 8) != hipSuccess) { (void)hipGetLastError(); ckbuf = nullptr; }
        al->ck = ckbuf;
        // column totals of the tolerance build's suffix-sum planes: 3 planes x S images x W_l doubles per level
        size_t wsum = 0; for (int l = 0; l < levels; l++) wsum += (size_t)Ws[l];
        if (hipMalloc((void **)&al->tot, (size_t)3 * S * wsum * 8) != hipSuccess) { (void)hipGetLastError(); al->tot = nullptr; }
        // frames taller than one k_cum_fused workgroup (512 rows): carry rows + flags of its row segments (level 0 sizes them; the coarser levels fit)
        const int xbands = (Hs[0] + 63) / 64, xseg = xbands <= CF_MAXW ? 1 : (xbands + CF_SEGW - 1) / CF_SEGW;
        if (xseg > 1 && ctx->arch_ok) {                          // (elsewhere: k_cum_cols + k_cum_rows, no cross-workgroup hand-over)
            const size_t nslot = (size_t)S * 3 * (xseg - 1);
            if (hipMalloc((void **)&al->xc, nslot * Ws[0] * 8) != hipSuccess) { (void)hipGetLastError(); al->xc = nullptr; }
            if (al->xc && (hipMalloc((void **)&al->xf, nslot * 4) != hipSuccess || hipMemset(al->xf, 0, nslot * 4) != hipSuccess)) { (void)hipGetLastError(); (void)hipFree(al->xc); al->xc = nullptr; al->xf = nullptr; }
            if (al->xc) al->xseg = xseg;
        }
    }
    for (int s = 0; s < S; s++) {
        slam_pyr *p = new slam_pyr();
        p->device = ctx->device; p->levels = levels;
        memcpy(p->H, Hs, sizeof Hs); memcpy(p->W, Ws, sizeof Ws); memcpy(p->P, Ps, sizeof Ps); memcpy(p->off, off, sizeof off);
        p->alloc = al;
        p->zstride = (size_t)o * 7;
        p->planes = al->base + (size_t)s * p->zstride;
        p->tmp = p->planes + (size_t)o * 6;
        p->batch_index = s; p->batch_size = S; p->ck = ckbuf;
        make_view(p);
        out[s] = p;
    }
    return SLAM_OK;
}

int slam_pyr_create(slam_ctx *ctx, int H, int W, int pyramid_levels, slam_pyr **out)
{
    return pyr_create_n(ctx, H, W, pyramid_levels, 1, out);
}

int slam_pyr_create_batch(slam_ctx *ctx, int H, int W, int pyramid_levels, int S, slam_pyr **out)
{
    return pyr_create_n(ctx, H, W, pyramid_levels, S, out);
}

// pyrs[0..S) must be the members of one slam_pyr_create_batch call, in order
int slam_pyr_update_batch_dev(slam_ctx *ctx, slam_pyr *const *pyrs, const double *const *images_dev, int S, int mode, double sigma, int sync)
{
    ARG_TRY(ctx, ctx != nullptr && pyrs != nullptr && images_dev != nullptr && S >= 1 && S <= BATCH_MAX && ((mode & ~SLAM_PYR_FLAGS) == 1 || (mode & ~SLAM_PYR_FLAGS) == 3) && sigma > 0);
    slam_pyr *p0 = pyrs[0];
    ARG_TRY(ctx, p0 != nullptr && p0->batch_index == 0 && p0->batch_size == S);
    for (int s = 0; s < S; s++) ARG_TRY(ctx, pyrs[s] != nullptr && pyrs[s]->alloc == p0->alloc && pyrs[s]->batch_index == s && images_dev[s] != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ImgPtrs ip;
    for (int s = 0; s < BATCH_MAX; s++) ip.p[s] = s < S ? images_dev[s] : nullptr;
    const size_t n = (size_t)p0->H[0] * p0->W[0];
    const bool fused_ingest = level0_fused(p0, mode & ~SLAM_PYR_FLAGS, S);          // the level-0 kernel reads the source images itself and writes the layer
    if (fused_ingest) hipLaunchKernelGGL(k_set_ptrs, dim3(1), dim3(BATCH_MAX), 0, ctx->stream, p0->alloc->srctab, ip);
    else hipLaunchKernelGGL(k_gather_images, dim3((n + 255) / 256, 1, S), dim3(256), 0, ctx->stream, ip, p0->plane(0, 0), p0->H[0], p0->W[0], p0->P[0], p0->zstride);
    int rc = enqueue_build(ctx, p0, mode, sigma, S, fused_ingest ? 1 : 0);
    for (int s = 0; s < S; s++) { pyrs[s]->target_only = (mode & SLAM_PYR_TARGET_ONLY) != 0; pyrs[s]->tol_planes = (mode & ~SLAM_PYR_FLAGS) == 3; }
    if (rc) return rc;
    if (sync) HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

// 8-bit frames already in HBM (column-major H x W bytes, as the KITTI reader decodes them): converted on the device
int slam_pyr_update_batch_u8_dev(slam_ctx *ctx, slam_pyr *const *pyrs, const uint8_t *const *images_u8_dev, int S, int mode, double sigma, int sync)
{
    ARG_TRY(ctx, ctx != nullptr && pyrs != nullptr && images_u8_dev != nullptr && S >= 1 && S <= BATCH_MAX && ((mode & ~SLAM_PYR_FLAGS) == 1 || (mode & ~SLAM_PYR_FLAGS) == 3) && sigma > 0);
    slam_pyr *p0 = pyrs[0];
    ARG_TRY(ctx, p0 != nullptr && p0->batch_index == 0 && p0->batch_size == S);
    for (int s = 0; s < S; s++) ARG_TRY(ctx, pyrs[s] != nullptr && pyrs[s]->alloc == p0->alloc && pyrs[s]->batch_index == s && images_u8_dev[s] != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ImgPtrsU8 ip;
    for (int s = 0; s < BATCH_MAX; s++) ip.p[s] = s < S ? images_u8_dev[s] : nullptr;
    const size_t n = (size_t)p0->H[0] * p0->W[0];
    const bool fused_ingest = level0_fused(p0, mode & ~SLAM_PYR_FLAGS, S);
    if (fused_ingest) { ImgPtrs iq; for (int s = 0; s < BATCH_MAX; s++) iq.p[s] = (const double *)ip.p[s]; hipLaunchKernelGGL(k_set_ptrs, dim3(1), dim3(BATCH_MAX), 0, ctx->stream, p0->alloc->srctab, iq); }
    else hipLaunchKernelGGL(k_gather_images_u8, dim3((n + 255) / 256, 1, S), dim3(256), 0, ctx->stream, ip, p0->plane(0, 0), p0->H[0], p0->W[0], p0->P[0], p0->zstride);
    int rc = enqueue_build(ctx, p0, mode, sigma, S, fused_ingest ? 2 : 0);
    for (int s = 0; s < S; s++) { pyrs[s]->target_only = (mode & SLAM_PYR_TARGET_ONLY) != 0; pyrs[s]->tol_planes = (mode & ~SLAM_PYR_FLAGS) == 3; }
    if (rc) return rc;
    if (sync) HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_pyr_destroy(slam_pyr *p)
{
    if (!p) return SLAM_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    if (p->alloc && --p->alloc->refs == 0) { (void)hipFree(p->alloc->base); if (p->alloc->ck) (void)hipFree(p->alloc->ck); if (p->alloc->tot) (void)hipFree(p->alloc->tot); if (p->alloc->xc) (void)hipFree(p->alloc->xc); if (p->alloc->xf) (void)hipFree(p->alloc->xf); if (p->alloc->srctab) (void)hipFree(p->alloc->srctab); delete p->alloc; }
    if (p->norm) (void)hipFree(p->norm);
    for (auto &g : p->graphs) (void)hipGraphExecDestroy(g.exec);
    delete p;
    return SLAM_OK;
}

int slam_pyr_update_dev(slam_ctx *ctx, slam_pyr *p, const double *image_dev, int mode, double sigma, int sync)
{
    ARG_TRY(ctx, ctx != nullptr && p != nullptr && image_dev != nullptr && (mode == 0 || (mode & ~SLAM_PYR_FLAGS) == 1 || (mode & ~SLAM_PYR_FLAGS) == 3) && sigma > 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ingest_dense(ctx, p, image_dev);
    int rc = enqueue_build(ctx, p, mode, sigma);
    if (rc) return rc;
    if (sync) HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_pyr_update(slam_ctx *ctx, slam_pyr *p, const double *image, int mode, double sigma)
{
    ARG_TRY(ctx, ctx != nullptr && p != nullptr && image != nullptr && (mode == 0 || (mode & ~SLAM_PYR_FLAGS) == 1 || (mode & ~SLAM_PYR_FLAGS) == 3) && sigma > 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *stage;
    int rc = slam_scratch2(ctx, (size_t)p->H[0] * p->W[0] * 8, &stage);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(stage, image, (size_t)p->H[0] * p->W[0] * 8, hipMemcpyHostToDevice, ctx->stream));
    ingest_dense(ctx, p, (const double *)stage);
    rc = enqueue_build(ctx, p, mode, sigma);
    if (rc) return rc;
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_pyr_update_u8(slam_ctx *ctx, slam_pyr *p, const uint8_t *image_u8, int mode, double sigma)
{
    ARG_TRY(ctx, ctx != nullptr && p != nullptr && image_u8 != nullptr && (mode == 0 || (mode & ~SLAM_PYR_FLAGS) == 1 || (mode & ~SLAM_PYR_FLAGS) == 3) && sigma > 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)p->H[0] * p->W[0];
    void *d8;
    int rc = slam_scratch2(ctx, n, &d8);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d8, image_u8, n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_u8_to_f64, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, p->plane(0, 0), (const unsigned char *)d8, p->H[0], p->W[0], p->P[0]);
    rc = enqueue_build(ctx, p, mode, sigma);
    if (rc) return rc;
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

// the same from an 8-bit image already in HBM (e.g. copied there by the caller from pinned memory on ctx's stream); sync == 0: enqueue only
int slam_pyr_update_u8_dev(slam_ctx *ctx, slam_pyr *p, const uint8_t *image_u8_dev, int mode, double sigma, int sync)
{
    ARG_TRY(ctx, ctx != nullptr && p != nullptr && image_u8_dev != nullptr && (mode == 0 || (mode & ~SLAM_PYR_FLAGS) == 1 || (mode & ~SLAM_PYR_FLAGS) == 3) && sigma > 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)p->H[0] * p->W[0];
    hipLaunchKernelGGL(k_u8_to_f64, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, p->plane(0, 0), (const unsigned char *)image_u8_dev, p->H[0], p->W[0], p->P[0]);
    int rc = enqueue_build(ctx, p, mode, sigma);
    if (rc) return rc;
    if (sync) HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

int slam_pyr_copy(slam_ctx *ctx, slam_pyr *dst, const slam_pyr *src)
{
    ARG_TRY(ctx, ctx != nullptr && dst != nullptr && src != nullptr);
    ARG_TRY(ctx, dst->levels == src->levels && dst->H[0] == src->H[0] && dst->W[0] == src->W[0]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst->planes, src->planes, (size_t)src->off[src->levels] * 6 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    dst->target_only = src->target_only;       // what the planes are travels with them
    dst->tol_planes = src->tol_planes;
    return SLAM_OK;
}

int slam_pyr_clone(slam_ctx *ctx, const slam_pyr *src, slam_pyr **out)
{
    ARG_TRY(ctx, ctx != nullptr && src != nullptr && out != nullptr);
    int rc = slam_pyr_create(ctx, src->H[0], src->W[0], src->levels - 1, out);
    if (rc) return rc;
    return slam_pyr_copy(ctx, *out, src);
}

int slam_pyr_shape(const slam_pyr *p, int level, int *H, int *W)
{
    if (!p || level < 0 || level >= p->levels) return SLAM_ERR_ARG;
    if (H) *H = p->H[level];
    if (W) *W = p->W[level];
    return SLAM_OK;
}

int slam_pyr_levels(const slam_pyr *p) { return p ? p->levels : SLAM_ERR_ARG; }

int slam_pyr_download(slam_ctx *ctx, const slam_pyr *p, int plane, int level, double *out)
{
    ARG_TRY(ctx, ctx != nullptr && p != nullptr && out != nullptr);
    ARG_TRY(ctx, plane >= 0 && plane < 6 && level >= 0 && level < p->levels);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)p->H[level] * p->W[level];
    void *stage;
    int rc = slam_scratch2(ctx, n * 8, &stage);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpitch, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (double *)stage, (const double *)p->plane(plane, level), p->H[level], p->W[level], p->P[level]);
    HIP_TRY(ctx, hipMemcpyAsync(out, stage, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

} // extern "C"
