// Hand-written batched complex128 FFT along z for power-of-two lengths (64 .. 4096) and for
// 9 x 2^k (576, 1152, 2304: a power-of-two slab plus its 2 x 64 guard cells), on the
// strided (Nz, ncols) view of the z-major field slabs: element (iz, col) at base + iz*stride
// + col.  rocFFT's strided-batch kernel reaches ~1.0-1.1 TB/s on this layout (measured,
// tools/fft_stride_probe.py); the grids of the PIC cycle are cache-resident, so a transform is
// really bound by LDS traffic and fp64 VALU work.  Here:
//
//   * one workgroup (256 lanes) transforms a tile of C = 4096/Nz adjacent columns, i.e. 4096
//     points = 16 per lane, whatever Nz: 64 B (Nz = 1024) .. 1 KiB contiguous per z row, so
//     the first pass reads the grid and the last pass writes it directly, coalesced, and
//     in-place transforms are safe (a tile is read completely before it is written);
//   * Stockham autosort passes of radix 16 / 8 (compile-time plan per Nz; three passes for
//     1024..4096: 1024 x 1536 columns 13.0 -> 12.3 us against the radix 8,8,4,4 plan): every lane
//     keeps its 16 points in registers, does its butterflies, and exchanges through a 66 KiB
//     LDS tile between passes (row index padded by row/8: conflict-free 16-B accesses for
//     both the strided writes of a pass and the contiguous reads of the next);
//   * twiddles exp(-2 pi i m / Nz) come from a table built on the host in extended
//     precision (one per Nz, cached for the life of the process); the backward transform
//     uses the conjugates and folds the 1/Nz of the reference (fourier.py:150-160) into its
//     last pass.
// Same conventions as np.fft.fft / ifft along axis 0 (fbpic/fields/spectral_transform/
// fourier.py:104-168).
#include "fb_common.h"
#include <cmath>
#include <map>
#include <vector>

namespace fb {

typedef double2 cx;
// non-temporal load of a complex value (the big-grid transforms read every element once: what they WRITE is what
// the next launch reads, and only that should stay in the 256 MB Infinity Cache - fb_common.h, FB_NT_LD)
typedef double zf_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cx zf_ld_nt(const cx *p)
{
#ifndef FB_NO_NT
    const zf_v2d v = __builtin_nontemporal_load((const zf_v2d *)p);
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}

__device__ __forceinline__ cx cadd(cx a, cx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cx csub(cx a, cx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cx cmul(cx a, cx b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i (FWD) or +i (backward)
template <bool FWD> __device__ __forceinline__ cx mul_mi(cx a)
{
    return FWD ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
}

template <int R, bool FWD> struct Dft;
template <bool FWD> struct Dft<2, FWD> {
    __device__ __forceinline__ static void run(cx *v)
    {
        const cx a = v[0], b = v[1];
        v[0] = cadd(a, b); v[1] = csub(a, b);
    }
};
template <bool FWD> struct Dft<4, FWD> {
    __device__ __forceinline__ static void run(cx *v)
    {
        const cx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        const cx t2 = cadd(v[1], v[3]), t3 = mul_mi<FWD>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2); v[2] = csub(t0, t2);
        v[1] = cadd(t1, t3); v[3] = csub(t1, t3);
    }
};
template <bool FWD> struct Dft<8, FWD> {
    __device__ __forceinline__ static void run(cx *v)
    {
        cx e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4, FWD>::run(e);
        Dft<4, FWD>::run(o);
        constexpr double h = 0.70710678118654752440;
        // o[k] *= W8^k, W8 = exp(-+ 2 pi i / 8)
        const cx o1 = FWD ? make_double2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x))
                          : make_double2(h * (o[1].x - o[1].y), h * (o[1].y + o[1].x));
        const cx o2 = mul_mi<FWD>(o[2]);
        const cx o3 = FWD ? make_double2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y))
                          : make_double2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y));
        v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
        v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
        v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
    }
};

template <bool FWD> struct Dft<16, FWD> {
    __device__ __forceinline__ static void run(cx *v)
    {
        cx e[8] = {v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14]};
        cx o[8] = {v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15]};
        Dft<8, FWD>::run(e);
        Dft<8, FWD>::run(o);
        // o[k] *= W16^k, W16 = exp(-+ 2 pi i / 16)
        constexpr double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178;
        constexpr double h = 0.70710678118654752440;
        const double sg = FWD ? -1. : 1.;
        o[1] = cmul(o[1], make_double2(c1, sg * s1));
        o[2] = cmul(o[2], make_double2(h, sg * h));
        o[3] = cmul(o[3], make_double2(s1, sg * c1));
        o[4] = mul_mi<FWD>(o[4]);
        o[5] = cmul(o[5], make_double2(-s1, sg * c1));
        o[6] = cmul(o[6], make_double2(-h, sg * h));
        o[7] = cmul(o[7], make_double2(-c1, sg * s1));
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = cadd(e[k], o[k]); v[k + 8] = csub(e[k], o[k]); }
    }
};

template <bool FWD> struct Dft<3, FWD> {
    __device__ __forceinline__ static void run(cx *v)
    {
        constexpr double h = 0.86602540378443864676;      // sin(2 pi / 3)
        const cx t1 = cadd(v[1], v[2]);
        const cx t2 = make_double2(v[0].x - 0.5 * t1.x, v[0].y - 0.5 * t1.y);
        const cx d = csub(v[1], v[2]);
        // (-+ i h) d
        const cx t3 = FWD ? make_double2(h * d.y, -h * d.x) : make_double2(-h * d.y, h * d.x);
        v[0] = cadd(v[0], t1);
        v[1] = cadd(t2, t3);
        v[2] = csub(t2, t3);
    }
};
template <bool FWD> struct Dft<6, FWD> {
    // 6 = 2 x 3: X[k] = E[k mod 3] + W6^k O[k mod 3], E / O = 3-point DFTs of the even / odd
    // inputs, W6 = exp(-+ 2 pi i / 6) = (1/2, -+ h); W6^(k+3) = -W6^k
    __device__ __forceinline__ static void run(cx *v)
    {
        constexpr double h = 0.86602540378443864676;
        cx e[3] = {v[0], v[2], v[4]}, o[3] = {v[1], v[3], v[5]};
        Dft<3, FWD>::run(e);
        Dft<3, FWD>::run(o);
        const cx w1 = make_double2(0.5, FWD ? -h : h), w2 = make_double2(-0.5, FWD ? -h : h);
        const cx o1 = cmul(o[1], w1), o2 = cmul(o[2], w2);
        v[0] = cadd(e[0], o[0]); v[3] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);   v[4] = csub(e[1], o1);
        v[2] = cadd(e[2], o2);   v[5] = csub(e[2], o2);
    }
};
template <bool FWD> struct Dft<9, FWD> {
    // 9 = 3 x 3 Cooley-Tukey: X[k1 + 3 k2] = sum_n2 W3^(n2 k2) W9^(n2 k1) sum_n1 W3^(n1 k1) x[3 n1 + n2]
    __device__ __forceinline__ static void run(cx *v)
    {
        cx y[3][3];
#pragma unroll
        for (int n2 = 0; n2 < 3; n2++) {
            cx t[3] = {v[n2], v[3 + n2], v[6 + n2]};
            Dft<3, FWD>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < 3; k1++) y[n2][k1] = t[k1];
        }
        // W9^m = exp(-+ 2 pi i m / 9), m = 1, 2, 4
        constexpr double c1 = 0.76604444311897803520, s1 = 0.64278760968653932632;
        constexpr double c2 = 0.17364817766693034885, s2 = 0.98480775301220805937;
        constexpr double c4 = -0.93969262078590838405, s4 = 0.34202014332566873304;
        const cx w1 = make_double2(c1, FWD ? -s1 : s1), w2 = make_double2(c2, FWD ? -s2 : s2);
        const cx w4 = make_double2(c4, FWD ? -s4 : s4);
        y[1][1] = cmul(y[1][1], w1); y[1][2] = cmul(y[1][2], w2);
        y[2][1] = cmul(y[2][1], w2); y[2][2] = cmul(y[2][2], w4);
#pragma unroll
        for (int k1 = 0; k1 < 3; k1++) {
            cx t[3] = {y[0][k1], y[1][k1], y[2][k1]};
            Dft<3, FWD>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < 3; k2++) v[k1 + 3 * k2] = t[k2];
        }
    }
};

// Compile-time configuration of one length: N points per column, C adjacent columns per
// workgroup, NTHR lanes, up to 5 Stockham passes of radix R0..R4 (1 = no pass).  Every lane
// keeps N C / NTHR points in registers; each pass must give every lane a whole number of
// butterflies: (N / R) C / NTHR integer.
template <int N_, int C_, int NTHR_, int R0_, int R1_, int R2_, int R3_, int R4_>
struct ZCfg {
    static constexpr int N = N_, C = C_, NTHR = NTHR_;
    static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_, R4 = R4_;
    static constexpr int E = N * C / NTHR;                  // points per lane
    static constexpr int LDS_CX = N * C + N / 8 + 1;        // complex slots incl. the row padding
    static_assert(R0 * R1 * R2 * R3 * R4 == N, "pass plan");
    static_assert(N * C % NTHR == 0 && NTHR % C == 0, "tile shape");
};
// powers of two: 4096 points per 256-lane workgroup (16 per lane), radix 16 / 8
typedef ZCfg<64, 64, 256, 8, 8, 1, 1, 1> ZC64;
typedef ZCfg<128, 32, 256, 16, 8, 1, 1, 1> ZC128;
typedef ZCfg<256, 16, 256, 16, 16, 1, 1, 1> ZC256;
typedef ZCfg<512, 8, 256, 8, 8, 8, 1, 1> ZC512;
typedef ZCfg<1024, 4, 256, 16, 8, 8, 1, 1> ZC1024;
// (round 6, measured at C2 - 1024 rows x 1024 / 1536 columns: two columns per 128-lane workgroup, <1024, 2, 128, ...>:
// 24.4 / 20.0 us for the records / (p, m) launch against 22.7 / 18.8; eight columns per 512-lane workgroup: 28.3 / 19.5;
// MORE WAVES on the same four-column tile - 8 points per lane, 512 lanes, radix 8 8 4 4: 20.8 / 16.4 - 17.1 against 21.0 /
// 18.1; 4 points per lane, 1024 lanes, radix 4^5: 21.1 - 22.5 / 17.8 - the step unchanged, profiles/r06_zfft_waves_per_tile.txt)
typedef ZCfg<2048, 2, 256, 16, 16, 8, 1, 1> ZC2048;
typedef ZCfg<4096, 1, 256, 16, 16, 16, 1, 1> ZC4096;
// 9 x 2^k (a power-of-two slab plus 2 x 64 guard cells, e.g. 1024 + 128): 4608 points per
// 192-lane workgroup (24 per lane), radix 6, 6 then 8 / 4: four passes (measured against the
// 128-lane radix 9, 4, 4, 4, 2 plan with 36 points per lane and five passes)
// head of the lengths 192 x R (fb_fft_generic: 4416 = 192 x 23), 16 columns per workgroup
// (12 points per lane, four passes, 3 waves per SIMD: 0.60 ms per step at 4416 rows against 0.64
// for the 24-point, three-pass plan <192, 16, 128, 6, 8, 4>)
typedef ZCfg<192, 16, 256, 6, 4, 4, 2, 1> ZC192;
typedef ZCfg<576, 8, 192, 6, 6, 4, 4, 1> ZC576;
typedef ZCfg<1152, 4, 192, 6, 6, 8, 4, 1> ZC1152;
typedef ZCfg<2304, 2, 192, 6, 6, 8, 8, 1> ZC2304;
// (4608 = 9 x 512 - the 4096-cell laser-wakefield window with 2 x 256 guard / damping /
// injection cells - was tried as ZCfg<4608, 1, 192, 6, 6, 8, 4, 4>: one column per workgroup
// means 16-B pieces of 4608 different rows per pass; 0.99 ms per step for the 32 transforms of
// that configuration against 0.80 ms for the three global passes of fb_fft_generic at 4416
// rows.  Lengths beyond 4096 stay with the generic pass-per-launch FFT.)

// One Stockham pass of radix R over the tile (NS = product of the previous radices):
//   butterfly jj in [0, N/R): inputs rows jj + t N/R, twiddled by W_{NS R}^{t (jj mod NS)},
//   R-point DFT, outputs to rows (jj - jj mod NS) R + jj mod NS + t NS.
// The first pass reads the grid, the last one writes it; the others exchange in place
// through LDS: every lane reads all of its points, barrier, then writes them.
// pm: 0 = plain load; 1 = this column is an r slot, r = p + m with m one field (gin2) to the
// right; 2 = a t slot, t = i (p - m) with p one field to the left (numba_pm_to_rt,
// spectral_transformer.py:140-142, folded into the first pass of the backward transform)
template <class Z, int R, int NS, bool FWD, bool FIRST, bool LAST, bool NTIN = false>
__device__ __forceinline__ void zf_pass(cx *lds, const cx *__restrict__ tw,
        const cx *gin, long in_stride, cx *gout, long out_stride,
        int c, int jj0, bool col_ok, double scale, int pm = 0, const cx *gin2 = nullptr,
        cx *gclear = nullptr)
{
    constexpr int N = Z::N, C = Z::C, NB = Z::E / R, JSTEP = Z::NTHR / C;
    constexpr int NR = N / R;
    static_assert(Z::E % R == 0, "whole butterflies per lane");
    cx v[NB][R];
    if (!FIRST) __syncthreads();                       // the previous pass has written
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int jj = jj0 + b * JSTEP;
#pragma unroll
        for (int t = 0; t < R; t++) {
            const int row = jj + t * NR;
            if (FIRST) {
                cx a = col_ok ? (NTIN ? zf_ld_nt(gin + (long)row * in_stride) : gin[(long)row * in_stride]) : make_double2(0., 0.);
                // consume-and-clear (fb_zfft_from_records_consume): every element of the record
                // array is read by exactly one lane of one workgroup
                if (gclear && col_ok) gclear[(long)row * in_stride] = make_double2(0., 0.);
                if (pm != 0 && col_ok) {
                    const cx o = gin2[(long)row * in_stride];
                    if (pm == 1) a = cadd(a, o);                               // p + m
                    else { const cx d = csub(o, a); a = make_double2(-d.y, d.x); }  // i (p - m)
                }
                v[b][t] = a;
            } else v[b][t] = lds[row * C + c + (row >> 3)];
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int jj = jj0 + b * JSTEP;
        if (!FIRST) {
            const int k = jj % NS;
            constexpr int TSTEP = N / (NS * R);
#pragma unroll
            for (int t = 1; t < R; t++) {
                cx w = tw[t * k * TSTEP];
                if (!FWD) w.y = -w.y;
                v[b][t] = cmul(v[b][t], w);
            }
        }
        Dft<R, FWD>::run(v[b]);
    }
    if (!FIRST && !LAST) __syncthreads();              // every lane has read its inputs
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int jj = jj0 + b * JSTEP;
        const int k = jj % NS;
        const int j0 = (jj - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; t++) {
            const int row = j0 + t * NS;
            if (LAST) {
                if (col_ok)
                    gout[(long)row * out_stride] = make_double2(v[b][t].x * scale, v[b][t].y * scale);
            } else {
                lds[row * C + c + (row >> 3)] = v[b][t];
            }
        }
    }
}

// SUB: the launch transforms `nsub` interleaved sub-sequences of every column - the head of a
// longer transform of length nsub * N (fb_fft_generic): sub-sequence b reads rows b + nsub * u
// (in_stride = nsub x the row stride, + b * sub_in) and writes rows b * N + k (+ b * sub_out).
// The virtual column index runs over (b, column).
template <class Z, bool FWD, bool SUB = false>
__global__ __launch_bounds__(Z::NTHR) void k_zfft(long ncols, const cx *in, long in_stride,
        cx *out, long out_stride, const cx *__restrict__ tw, double scale, int ntiles, int pm_Nr,
        int aos_Nr, int aos_rec, int aos_clear, long sub_in = 0, long sub_out = 0, int nsub = 1)
{
    constexpr int C = Z::C;
    extern __shared__ double2 zf_lds[];
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs; give each XCD a
    // contiguous range of tiles so that neighbouring column tiles share its L2
    const int nb = gridDim.x;
    int tile = blockIdx.x;
    if ((nb & 7) == 0) tile = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int c = threadIdx.x % C, jj0 = threadIdx.x / C;
    long col = (long)tile * C + c;
    if (!SUB && aos_Nr > 0 && aos_Nr % C == 0) {
        // records input: order the tiles node-block major, field minor, so that the tiles that
        // read the same record lines (one 16-B field each) are neighbours on one XCD and the
        // line comes from HBM once, not once per field and XCD
        const int nfld = (int)(ncols / aos_Nr);
        col = (long)(tile % nfld) * aos_Nr + (long)(tile / nfld) * C + c;
    }
    long col_sub_in = 0, col_sub_out = 0;
    bool col_ok = col < ncols;
    if constexpr (SUB) {
        col_ok = col < ncols * nsub;
        long b = col / ncols;
        col -= b * ncols;
        if (aos_Nr > 0 && aos_Nr % C == 0) {
            // records input: (sub-sequence, node block) major, field minor - the tiles that read
            // the same 128-B records (one 16-B field each) are neighbours on one XCD
            const int nfld = (int)(ncols / aos_Nr), nblk = aos_Nr / C;
            const int q = tile / nfld;
            b = q / nblk;
            col = (long)(tile % nfld) * aos_Nr + (long)(q % nblk) * C + c;
            col_ok = b < nsub;
        }
        col_sub_in = b * sub_in; col_sub_out = b * sub_out;
    }
    // aos_Nr > 0: the input is node-major, in[iz * in_stride + ir * aos_rec + field] (the
    // deposition's record-per-node target); column (field, ir) of the transform gathers it
    const cx *gin = (aos_Nr > 0 ? in + (col % aos_Nr) * aos_rec + (col / aos_Nr) : in + col) + col_sub_in;
    cx *gout = out + col + col_sub_out;
    // pm_Nr > 0: the columns are groups of (p, m, z) fields of pm_Nr columns each
    int pm = 0;
    const cx *gin2 = nullptr;
    if (pm_Nr > 0) {
        const int slot = (int)((col / pm_Nr) % 3);
        if (slot == 0) { pm = 1; gin2 = gin + pm_Nr; }
        else if (slot == 1) { pm = 2; gin2 = gin - pm_Nr; }
    }
    constexpr int R0 = Z::R0, R1 = Z::R1, R2 = Z::R2, R3 = Z::R3, R4 = Z::R4;
    constexpr int NPASS = (R1 == 1) ? 1 : (R2 == 1) ? 2 : (R3 == 1) ? 3 : (R4 == 1) ? 4 : 5;
#define ZF_ARGS zf_lds, tw, gin, in_stride, gout, out_stride, c, jj0, col_ok, scale
    zf_pass<Z, R0, 1, FWD, true, NPASS == 1, SUB>(ZF_ARGS, pm, gin2,
                                             (aos_clear && aos_Nr > 0) ? const_cast<cx *>(gin) : nullptr);
    if constexpr (NPASS >= 2) zf_pass<Z, R1, R0, FWD, false, NPASS == 2>(ZF_ARGS);
    if constexpr (NPASS >= 3) zf_pass<Z, R2, R0 * R1, FWD, false, NPASS == 3>(ZF_ARGS);
    if constexpr (NPASS >= 4) zf_pass<Z, R3, R0 * R1 * R2, FWD, false, NPASS == 4>(ZF_ARGS);
    if constexpr (NPASS >= 5) zf_pass<Z, R4, R0 * R1 * R2 * R3, FWD, false, true>(ZF_ARGS);
#undef ZF_ARGS
}

// ---- twiddle tables, one per length, built once
static std::map<int, cx *> g_twiddles;

static int get_twiddles(int N, const cx **out)
{
    auto it = g_twiddles.find(N);
    if (it != g_twiddles.end()) { *out = it->second; return 0; }
    std::vector<double> h(2 * (size_t)N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < N; m++) {
        // exact octant symmetries keep cos/sin of the eighth-turns exact
        const long double a = two_pi * (long double)m / (long double)N;
        h[2 * m] = (double)cosl(a);
        h[2 * m + 1] = (double)(-sinl(a));
    }
    cx *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, 2 * sizeof(double) * (size_t)N);
    if (e != hipSuccess) return check(e, "fb_zfft(twiddles)");
    e = hipMemcpy(d, h.data(), 2 * sizeof(double) * (size_t)N, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return check(e, "fb_zfft(twiddles)"); }
    g_twiddles[N] = d;
    *out = d;
    return 0;
}

template <class Z>
static int zfft_launch(long ncols, const cx *in, long is, cx *out, long os, int direction,
                       const cx *tw, hipStream_t s, int pm_Nr = 0, int aos_Nr = 0, int aos_rec = 0,
                       int aos_clear = 0)
{
    constexpr int N = Z::N, C = Z::C;
    const int ntiles = (int)((ncols + C - 1) / C);
    const size_t lds_bytes = (size_t)Z::LDS_CX * sizeof(cx);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_zfft<Z, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_zfft<Z, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return check(e, "fb_zfft(attr)");
        attr_set = true;
    }
    const int nblocks = (ntiles + 7) & ~7;             // multiple of 8 for the XCD mapping
    if (direction < 0)
        hipLaunchKernelGGL((k_zfft<Z, true>), dim3(nblocks), dim3(Z::NTHR), lds_bytes, s, ncols, in,
                           is, out, os, tw, 1.0, ntiles, pm_Nr, aos_Nr, aos_rec, aos_clear);
    else
        hipLaunchKernelGGL((k_zfft<Z, false>), dim3(nblocks), dim3(Z::NTHR), lds_bytes, s, ncols, in,
                           is, out, os, tw, 1.0 / (double)N, ntiles, pm_Nr, aos_Nr, aos_rec, aos_clear);
    return check(hipGetLastError(), "fb_zfft");
}

// head of fb_fft_generic for Nz = 192 x nsub: all the 192-point sub-transforms in one launch
static int zfft_head192(int nsub, long ncols, const cx *in, long is, cx *out, long os, bool fwd,
                        hipStream_t s, int aos_Nr = 0, int aos_rec = 0, int aos_clear = 0)
{
    using Z = ZC192;
    const cx *tw = nullptr;
    int r = get_twiddles(Z::N, &tw);
    if (r) return r;
    const size_t lds_bytes = (size_t)Z::LDS_CX * sizeof(cx);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_zfft<Z, true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_zfft<Z, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return check(e, "fb_fft_generic(attr)");
        attr_set = true;
    }
    const long vcols = ncols * nsub;
    const int ntiles = (int)((vcols + Z::C - 1) / Z::C);
    const int nblocks = (ntiles + 7) & ~7;
    if (fwd)
        hipLaunchKernelGGL((k_zfft<Z, true, true>), dim3(nblocks), dim3(Z::NTHR), lds_bytes, s, ncols, in,
                           (long)nsub * is, out, os, tw, 1.0, ntiles, 0, aos_Nr, aos_rec, aos_clear, is,
                           (long)Z::N * os, nsub);
    else
        hipLaunchKernelGGL((k_zfft<Z, false, true>), dim3(nblocks), dim3(Z::NTHR), lds_bytes, s, ncols, in,
                           (long)nsub * is, out, os, tw, 1.0, ntiles, 0, 0, 0, 0, is, (long)Z::N * os, nsub);
    return check(hipGetLastError(), "fb_fft_generic(head)");
}

// Direct DFT of odd prime length R = 2 h + 1 with the conjugate symmetry of the roots folded
// in: X[u], X[R-u] = (v0 + sum_t c_tu S_t) -+ i (sum_t s_tu D_t) with S_t = v[t] + v[R-t],
// D_t = v[t] - v[R-t], c/s = cos/sin(2 pi (t u mod R) / R): 4 FMAs per (t, u) pair instead of
// the 8 flops x 4 pairs of the plain double loop.  cr/sr hold cos/sin(2 pi m / R), m < R.
template <int R, bool FWD>
__device__ __forceinline__ void dft_odd(cx *v, const double *cr, const double *sr)
{
    constexpr int H = R / 2;
    cx S[H + 1], D[H + 1];
    cx x0 = v[0];
#pragma unroll
    for (int t = 1; t <= H; t++) {
        S[t] = cadd(v[t], v[R - t]);
        D[t] = csub(v[t], v[R - t]);
        x0 = cadd(x0, S[t]);
    }
    cx out[R];
    out[0] = x0;
#pragma unroll
    for (int u = 1; u <= H; u++) {
        cx P = v[0], Q = make_double2(0., 0.);
#pragma unroll
        for (int t = 1; t <= H; t++) {
            const int m = (t * u) % R;
            P.x = __builtin_fma(cr[m], S[t].x, P.x); P.y = __builtin_fma(cr[m], S[t].y, P.y);
            Q.x = __builtin_fma(sr[m], D[t].x, Q.x); Q.y = __builtin_fma(sr[m], D[t].y, Q.y);
        }
        // forward: X[u] = P - i Q, X[R-u] = P + i Q ; backward: the other way round
        const cx a = make_double2(P.x + Q.y, P.y - Q.x), b = make_double2(P.x - Q.y, P.y + Q.x);
        out[u] = FWD ? a : b;
        out[R - u] = FWD ? b : a;
    }
#pragma unroll
    for (int u = 0; u < R; u++) v[u] = out[u];
}

// Composite radix 24 = 3 x 8 in registers (Cooley-Tukey): one pass instead of a radix-8 and a
// radix-3 pass, i.e. one sweep less over the slab for lengths like 4416 = 24 * 8 * 23.
//   X[k1 + 3 k2] = sum_n2 W8^(n2 k2) W24^(n2 k1) sum_n1 W3^(n1 k1) x[8 n1 + n2]
// w24[m] = exp(-+ 2 pi i m / 24) for m < 15 (largest n2 k1 = 7 * 2).
template <bool FWD>
__device__ __forceinline__ void dft24(cx *v, const cx *w24)
{
    cx y[8][3];
#pragma unroll
    for (int n2 = 0; n2 < 8; n2++) {
        cx t[3] = {v[n2], v[8 + n2], v[16 + n2]};
        Dft<3, FWD>::run(t);
        y[n2][0] = t[0];
        y[n2][1] = n2 ? cmul(t[1], w24[n2]) : t[1];
        y[n2][2] = n2 ? cmul(t[2], w24[2 * n2]) : t[2];
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
        cx t[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; n2++) t[n2] = y[n2][k1];
        Dft<8, FWD>::run(t);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) v[k1 + 3 * k2] = t[k2];
    }
}

// roots of unity of order R from the table of N-th roots (tw[j] = exp(-2 pi i j / N))
template <int R>
__device__ __forceinline__ void load_roots(const cx *__restrict__ tw, int NR, double *cr, double *sr)
{
#pragma unroll
    for (int m = 0; m < R; m++) {
        const cx w = tw[m * NR];
        cr[m] = w.x; sr[m] = -w.y;
    }
}

// ---- any other length: one Stockham pass per launch through global memory --------------
// rocFFT (ROCm 7.2) refuses some lengths outright (e.g. 4416 = 2^6 * 3 * 23, the local grid of
// the 4096-cell laser-wakefield run with its guard, damping and injection cells), so the
// fallback is self-contained: the same pass as zf_pass, one lane per (butterfly, column),
// radix 8 / 4 / 2 butterflies or a direct O(R^2) DFT for an odd prime R <= 31, ping-pong
// between the destination and a scratch slab.  Coalesced along the columns; every pass is a
// full sweep over the group, so the transform is bound by passes x 32 B per point (measured on
// the 4416-row laser-wakefield grid: 1.4 ms per step for its 44 field transforms, i.e. the
// HBM rate).  An LDS-resident variant (one column per workgroup, 2 x 70 KiB) was slower: one
// wave per SIMD and 16-B row pieces leave it latency-bound.
template <int R, bool FWD>
__global__ __launch_bounds__(256) void k_fft_pass(int N, int NS, long ncols, const cx *in, long is,
        cx *out, long os, const cx *__restrict__ tw, double scale)
{
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    const int NR = N / R;
    double cr[R], sr[R];
    if constexpr (!(R == 2 || R == 4 || R == 8 || R == 3 || R == 9 || R == 24)) load_roots<R>(tw, NR, cr, sr);
    for (int jj = blockIdx.y; jj < NR; jj += gridDim.y) {
        const int k = jj % NS;
        const int tstep = N / (NS * R);
        cx v[R];
#pragma unroll
        for (int t = 0; t < R; t++) {
            v[t] = zf_ld_nt(in + ((long)(jj + t * NR) * is + col));
            if (t > 0 && NS > 1) {
                cx w = tw[t * k * tstep];
                if (!FWD) w.y = -w.y;
                v[t] = cmul(v[t], w);
            }
        }
        cx X[R];
        if constexpr (R == 24) {
            cx w24[15];
#pragma unroll
            for (int m = 0; m < 15; m++) {
                w24[m] = tw[m * NR];
                if (!FWD) w24[m].y = -w24[m].y;
            }
            dft24<FWD>(v, w24);
#pragma unroll
            for (int u = 0; u < R; u++) X[u] = v[u];
        } else if constexpr (R == 2 || R == 4 || R == 8 || R == 3 || R == 9) {
            Dft<R, FWD>::run(v);
#pragma unroll
            for (int u = 0; u < R; u++) X[u] = v[u];
        } else {
            dft_odd<R, FWD>(v, cr, sr);
#pragma unroll
            for (int u = 0; u < R; u++) X[u] = v[u];
        }
        const long j0 = (long)(jj - k) * R + k;
#pragma unroll
        for (int u = 0; u < R; u++)
            out[(j0 + (long)u * NS) * os + col] = make_double2(X[u].x * scale, X[u].y * scale);
    }
}

__global__ __launch_bounds__(256) void k_fft_copy(int N, long ncols, const cx *in, long is, cx *out, long os)
{
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    for (int r = blockIdx.y; r < N; r += gridDim.y) out[(long)r * os + col] = in[(long)r * is + col];
}

template <int R>
static void pass_launch(bool fwd, dim3 grid, hipStream_t s, int N, int NS, long ncols, const cx *in, long is,
                        cx *out, long os, const cx *tw, double scale)
{
    if (fwd) hipLaunchKernelGGL((k_fft_pass<R, true>), grid, dim3(256), 0, s, N, NS, ncols, in, is, out, os, tw, scale);
    else hipLaunchKernelGGL((k_fft_pass<R, false>), grid, dim3(256), 0, s, N, NS, ncols, in, is, out, os, tw, scale);
}

// radices of the passes for length N (largest power-of-two radices first), 0 if a prime
// factor exceeds 31
static int factorize(int N, int *radix)
{
    int n = N, np = 0;
    if (n % 24 == 0 && n % 9 != 0) { radix[np++] = 24; n /= 24; }      // 3 x 8 in one pass
    while (n % 8 == 0) { radix[np++] = 8; n /= 8; }
    while (n % 4 == 0) { radix[np++] = 4; n /= 4; }
    while (n % 2 == 0) { radix[np++] = 2; n /= 2; }
    while (n % 9 == 0) { radix[np++] = 9; n /= 9; }
    const int primes[] = {3, 5, 7, 11, 13, 17, 19, 23, 29, 31};
    for (int p : primes)
        while (n % p == 0) { radix[np++] = p; n /= p; }
    return n == 1 ? np : 0;
}

// Nz = 192 x R with a single-pass R (4416 = 192 x 23, the 4096-cell laser-wakefield window):
// the passes of the 192-point factor run in ONE launch through LDS (k_zfft on the interleaved
// sub-sequences, 256-B row pieces) and only the radix-R pass sweeps the slab again: two sweeps
// instead of three (24, 8, 23).  The head can gather its input from the deposition's records
// (and clear them), like the single-launch LDS transform.
static bool fft_two_sweep_supported(int Nz)
{
    if (Nz % 192 != 0 || Nz <= 192) return false;
    int rr[40];
    return factorize(Nz / 192, rr) == 1;
}

static int fft_two_sweep(int Nz, long ncols, const cx *in, long in_stride, cx *out, long out_stride,
                         cx *scratch, long scratch_stride, bool fwd, const cx *tw, hipStream_t s,
                         int aos_Nr, int aos_rec, int aos_clear)
{
    const int R = Nz / 192;
    int r = zfft_head192(R, ncols, in, in_stride, scratch, scratch_stride, fwd, s, aos_Nr, aos_rec, aos_clear);
    if (r) return r;
    const dim3 g2((unsigned)((ncols + 255) / 256), 192u);
    const double scale = fwd ? 1.0 : 1.0 / (double)Nz;
#define FB_PASS(RR) case RR: pass_launch<RR>(fwd, g2, s, Nz, 192, ncols, scratch, scratch_stride, out, out_stride, tw, scale); break
    switch (R) {
    FB_PASS(2); FB_PASS(3); FB_PASS(4); FB_PASS(5); FB_PASS(7); FB_PASS(8); FB_PASS(9); FB_PASS(11);
    FB_PASS(13); FB_PASS(17); FB_PASS(19); FB_PASS(23); FB_PASS(24); FB_PASS(29); FB_PASS(31);
    }
#undef FB_PASS
    return check(hipGetLastError(), "fb_fft_generic");
}

}  // namespace fb

using namespace fb;

extern "C" int fb_fft_generic_supported(int Nz)
{
    int radix[40];
    return Nz >= 2 && factorize(Nz, radix) > 0;
}

extern "C" int fb_fft_generic(int Nz, long ncols, const void *in, long in_stride, void *out,
                              long out_stride, void *scratch, long scratch_stride, int direction,
                              void *stream)
{
    int radix[40];
    const int np = factorize(Nz, radix);
    if (Nz < 2 || np == 0) { set_error("fb_fft_generic", "Nz has a prime factor > 31"); return -1; }
    if (ncols <= 0) return 0;
    if (!scratch || scratch == in || scratch == out) { set_error("fb_fft_generic", "a distinct scratch slab is needed"); return -1; }
    const cx *tw = nullptr;
    int r = get_twiddles(Nz, &tw);
    if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    const bool fwd = direction < 0;
    // Nz = 192 x R with a single-pass R (4416 = 192 x 23, the 4096-cell laser-wakefield window):
    // the passes of the 192-point factor run in ONE launch through LDS (k_zfft on the
    // interleaved sub-sequences, 256-B row pieces) and only the radix-R pass sweeps the slab
    // again: two sweeps instead of three (24, 8, 23).
    if (fft_two_sweep_supported(Nz))
        return fft_two_sweep(Nz, ncols, (const cx *)in, in_stride, (cx *)out, out_stride, (cx *)scratch,
                             scratch_stride, fwd, tw, s, 0, 0, 0);
    const dim3 grid((unsigned)((ncols + 255) / 256), (unsigned)(Nz / radix[np - 1] < 1024 ? Nz / radix[np - 1] : 1024));
    // buffers alternate so that the last pass lands in `out`; pass 0 must not write what it
    // reads: in place with an odd number of passes starts with a copy to the scratch slab
    const cx *src = (const cx *)in;
    long src_stride = in_stride;
    cx *bufs[2] = {(cx *)out, (cx *)scratch};
    long strides[2] = {out_stride, scratch_stride};
    int which = (np % 2 == 1) ? 0 : 1;                 // destination of pass 0
    if (in == out && which == 0) {
        hipLaunchKernelGGL(k_fft_copy, grid, dim3(256), 0, s, Nz, ncols, src, src_stride, bufs[1], strides[1]);
        src = bufs[1]; src_stride = strides[1];
    }
    int NS = 1;
    for (int p = 0; p < np; p++) {
        const double scale = (p == np - 1 && !fwd) ? 1.0 / (double)Nz : 1.0;
        cx *dst = bufs[which];
        const long ds = strides[which];
#define FB_PASS(R) case R: pass_launch<R>(fwd, grid, s, Nz, NS, ncols, src, src_stride, dst, ds, tw, scale); break
        switch (radix[p]) {
        FB_PASS(2); FB_PASS(3); FB_PASS(4); FB_PASS(5); FB_PASS(7); FB_PASS(8); FB_PASS(9); FB_PASS(11);
        FB_PASS(13); FB_PASS(17); FB_PASS(19); FB_PASS(23); FB_PASS(24); FB_PASS(29); FB_PASS(31);
        }
#undef FB_PASS
        NS *= radix[p];
        src = dst; src_stride = ds;
        which ^= 1;
    }
    return check(hipGetLastError(), "fb_fft_generic");
}


extern "C" int fb_fft_generic_from_records_supported(int Nz) { return fft_two_sweep_supported(Nz) ? 1 : 0; }

extern "C" int fb_fft_generic_from_records_consume(int Nz, int nfields, int Nr, void *in, long in_stride,
                                                   int record, void *out, long out_stride, void *scratch,
                                                   long scratch_stride, void *stream)
{
    const char *who = "fb_fft_generic_from_records_consume";
    if (!fft_two_sweep_supported(Nz)) { set_error(who, "Nz must be 192 x R with a single-pass R"); return -1; }
    if (nfields != record || Nr <= 0) { set_error(who, "the whole record must be transformed (nfields == record)"); return -1; }
    if (!scratch || scratch == in || scratch == out || in == out) { set_error(who, "in, out and scratch must be distinct"); return -1; }
    const cx *tw = nullptr;
    int r = get_twiddles(Nz, &tw);
    if (r) return r;
    return fft_two_sweep(Nz, (long)nfields * Nr, (const cx *)in, in_stride, (cx *)out, out_stride,
                         (cx *)scratch, scratch_stride, true, tw, (hipStream_t)stream, Nr, record, 1);
}

extern "C" int fb_zfft_supported(int Nz)
{
    switch (Nz) {
    case 64: case 128: case 256: case 512: case 1024: case 2048: case 4096:
    case 576: case 1152: case 2304:
        return 1;
    default:
        return 0;
    }
}

static int zfft_dispatch(const char *who, int Nz, long ncols, const void *in, long in_stride, void *out,
                         long out_stride, int direction, int pm_Nr, void *stream, int aos_Nr = 0,
                         int aos_rec = 0, int aos_clear = 0);

extern "C" int fb_zfft_from_records(int Nz, int nfields, int Nr, const void *in, long in_stride,
                                    int record, void *out, long out_stride, void *stream)
{
    if (nfields <= 0 || nfields > record || Nr <= 0) {
        set_error("fb_zfft_from_records", "need 0 < nfields <= record and Nr > 0");
        return -1;
    }
    if (in == out) { set_error("fb_zfft_from_records", "out of place only"); return -1; }
    return zfft_dispatch("fb_zfft_from_records", Nz, (long)nfields * Nr, in, in_stride, out, out_stride,
                         -1, 0, stream, Nr, record);
}

extern "C" int fb_zfft_from_records_consume(int Nz, int nfields, int Nr, void *in, long in_stride,
                                            int record, void *out, long out_stride, void *stream)
{
    if (nfields != record || Nr <= 0) {
        set_error("fb_zfft_from_records_consume", "the whole record must be transformed (nfields == record)");
        return -1;
    }
    if (in == out) { set_error("fb_zfft_from_records_consume", "out of place only"); return -1; }
    return zfft_dispatch("fb_zfft_from_records_consume", Nz, (long)nfields * Nr, in, in_stride, out,
                         out_stride, -1, 0, stream, Nr, record, 1);
}

extern "C" int fb_zfft_pm_to_rt(int Nz, long ncols, const void *in, long in_stride, void *out,
                                long out_stride, int Nr, void *stream)
{
    if (Nr <= 0 || ncols % (3L * Nr) != 0) {
        set_error("fb_zfft_pm_to_rt", "ncols must be a multiple of 3 * Nr");
        return -1;
    }
    if (in == out) { set_error("fb_zfft_pm_to_rt", "out of place only (a column reads its neighbour field)"); return -1; }
    return zfft_dispatch("fb_zfft_pm_to_rt", Nz, ncols, in, in_stride, out, out_stride, +1, Nr, stream);
}

extern "C" int fb_zfft(int Nz, long ncols, const void *in, long in_stride, void *out,
                       long out_stride, int direction, void *stream)
{
    return zfft_dispatch("fb_zfft", Nz, ncols, in, in_stride, out, out_stride, direction, 0, stream);
}

static int zfft_dispatch(const char *who, int Nz, long ncols, const void *in, long in_stride, void *out,
                         long out_stride, int direction, int pm_Nr, void *stream, int aos_Nr, int aos_rec,
                         int aos_clear)
{
    if (!fb_zfft_supported(Nz)) {
        set_error(who, "unsupported Nz (2^k in [64, 4096] or 9 * 2^k in [576, 2304])");
        return -1;
    }
    if (ncols <= 0) return 0;
    if (in == out && in_stride != out_stride) { set_error(who, "in-place needs equal strides"); return -1; }
    const cx *tw = nullptr;
    int r = get_twiddles(Nz, &tw);
    if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    const cx *a = (const cx *)in;
    cx *b = (cx *)out;
#define ZF_CASE(n, Z) case n: return zfft_launch<Z>(ncols, a, in_stride, b, out_stride, direction, tw, s, pm_Nr, aos_Nr, aos_rec, aos_clear)
    switch (Nz) {
    ZF_CASE(64, ZC64); ZF_CASE(128, ZC128); ZF_CASE(256, ZC256); ZF_CASE(512, ZC512);
    ZF_CASE(1024, ZC1024); ZF_CASE(2048, ZC2048); ZF_CASE(4096, ZC4096);
    ZF_CASE(576, ZC576); ZF_CASE(1152, ZC1152); ZF_CASE(2304, ZC2304);
    }
#undef ZF_CASE
    return -1;
}
