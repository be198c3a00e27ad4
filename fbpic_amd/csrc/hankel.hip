// Discrete Hankel transform along r as an fp64 MFMA GEMM on gfx950.
//
// out[iz, n] = alpha * fz[iz] * fr[n] * sum_k ( in[iz, k] * sk[k] ) * mat[k, n]
//   (complex row x real (Nr,Nr) matrix; the optional real scalings sk / fz / fr fuse the
//    reference's divide-by-volume pass -- it commutes with the z-FFT -- and its spectral
//    filter pass into the transform; all three default to 1)
//
// The reference splits the complex (Nz,Nr) array into a real (2Nz,Nr) one, calls cuBLAS
// dgemm and re-interleaves (fbpic/fields/spectral_transform/hankel.py:196-205).  Here the
// split is free: a lane loads one complex element (16 B) and feeds its real part to one
// v_mfma_f64_16x16x4_f64 accumulator chain and its imaginary part to a second chain that
// shares the same B operand; the two accumulators hold re/im of the same output element
// in the same lane/register, so the store is again one 16-B complex write.
//
// Operands are staged through LDS (see k_hankel): fragment-shaped loads straight from global
// memory touch 16 half-used cache lines per instruction and saturate the texture-address path
// long before the MFMA pipe (measured in round 1: 12% of peak).
// Jobs (field x mode) are batched along gridDim.z so one launch fills 256 CUs.
//
// Fragment layouts (cdna_hip_programming.md section 3, f64 row formula):
//   A: lane l -> A[i = l & 15][k = l >> 4]      B: lane l -> B[k = l >> 4][j = l & 15]
//   D: reg r of lane l -> D[i = (l >> 4) + 4 r][j = l & 15]
#include <cstdlib>
#include "fb_common.h"

namespace fb {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int HK_MAXJOBS = 48;
struct HankelJobs {
    const cplx *in[HK_MAXJOBS];
    cplx *out[HK_MAXJOBS];
    const double *mat[HK_MAXJOBS];
};
// Optional (r, t) -> (p | m) combination fused into the operand load (rt_to_pm of
// spectral_transformer.py:208-210 without a sweep of its own): the input of job j is
// 0.5 * (in[j] + sgn[j] * i * in2[j]) when in2[j] != 0, i.e. p for sgn = -1, m for sgn = +1.
struct HankelPairs {
    const cplx *in2[HK_MAXJOBS];
    double sgn[HK_MAXJOBS];
};
struct HankelScales {
    const double *sk[HK_MAXJOBS];     // per input column k (e.g. 1/volume), or null
    const double *fz[HK_MAXJOBS];     // per output row iz (filter along z), or null
    const double *fr[HK_MAXJOBS];     // per output column n (filter along r), or null
};

// Tiling.  Workgroup = 4 waves (2 x 2) = 32 z rows x 64 output columns; wave (wz, wn) owns rows
// [16 wz, +16) and columns [32 wn, +32): 2 column sub-tiles x (re, im) = 4 accumulator chains.
// K is walked in chunks of 16 through double-buffered LDS panels (A[row][k] complex, row
// stride 2*16+2 doubles: the ds_read_b128 of 16 rows hits 64 distinct banks; B[k][n] doubles,
// row stride 64+16: the four k rows of a fragment alternate between the two bank halves), with
// TWO register stages in front of them: the coalesced global loads (256-B row segments of the
// input, full rows of the matrix panel) of chunk c+2 are issued before the MFMAs of chunk c,
// those of chunk c+1 - issued one chunk earlier - are written to the other LDS buffer after
// them; one barrier per chunk.
//
// What limits the transform at the headline size (1024 x 128: 67 MFLOP and 4 MB per job) is not
// the matrix pipe but keeping it fed (rocprofv3 SQ counters, profiles/): the round-1 kernel
// (K-chunks of 32, one register stage, 75 KB of LDS -> two of the three workgroups of a CU
// resident) ran its load / compute / store phases in lockstep across the chip, 1.1 waves
// resident per SIMD on average, MFMA pipe 31 % busy.  With 38 KB of LDS all workgroups of a
// launch are resident at once (768 for E + B: every SIMD holds exactly 3 waves, no second
// round), and two chunks of global traffic are in flight per workgroup: 27.1 -> 22.6 us for
// the 12 transforms of E + B at 1024 x 128, 1080 -> 867 us for the 24 of 2048 x 512 (Nm = 4),
// 283 -> 224 us at 4096 x 256.  The wave tile stays 16 x 32; the workgroup is WZ x WN waves.
// For the large grids a 64 x 128 workgroup (16 waves, 128 VGPRs each, one per CU) re-reads the
// matrix 4x less often: 867 -> 815 us at 2048 x 512.  At the headline size no tile shape
// (16 x 128, 32 x 128, 64 x 128; 4 / 8 / 16 waves; wave tiles of 16 x 64 and 16 x 128) beats
// 32 x 64: with only 4 MB per transform the launch is one "generation" of workgroups whose
// first loads (nothing to compute yet) and final stores (nothing left to compute) do not
// overlap any MFMA work - ~8 us of its ~21 us (MFMA pipe busy 42 % over the kernel's
// duration; rocprofv3 MfmaUtil reads 29 % because GRBM_GUI_ACTIVE also counts ~7 us around
// the dispatch).
//
// DUAL (backward transform of a vector field, (p, m) -> (r, t) folded into the GEMM,
// spectral_transformer.py:89-155): a job with in2 != 0 computes BOTH p' = in . mat and
// m' = in2 . mat2 (K walked over the first product, then over the second, two accumulator
// sets - the same MFMA work as two plain jobs) and writes
//   out = p' + m' (the r slot)  and  out2 = i (p' - m') (the t slot).
// in2 comes from Pr.in2, mat2 travels in Sc.sk and out2 in Sc.fz; jobs with in2 == 0 are
// plain transforms (the z components).
constexpr int H2_KC = 16;
constexpr int H2_RSA = 2 * H2_KC + 2;
// WZ x WN waves (rows x columns) of wave tiles of 16 rows x WC columns
template <int WC, int WN, int WZ> struct H2Cfg {
    static constexpr int NTHR = 64 * WZ * WN;
    static constexpr int TZ = 16 * WZ, TN = WC * WN;        // rows / columns per workgroup
    static constexpr int RSB = TN + 16;
    static constexpr int ABUF = TZ * H2_RSA, BBUF = H2_KC * RSB;
    static constexpr size_t LDS_BYTES = (size_t)2 * (ABUF + BBUF) * 8;
};

template <bool SCALED, bool PAIRED, bool DUAL, int WC, int WN, int WZ, int WPE, bool FULL>
__global__ __launch_bounds__(64 * WZ * WN) __attribute__((amdgpu_waves_per_eu(WPE, 4))) void k_hankel(HankelJobs J, HankelScales Sc, HankelPairs Pr,
                                                         long irs, long ors, double alpha, int Nz, int Nr)
{
    using C = H2Cfg<WC, WN, WZ>;
    constexpr int TZ = C::TZ, TN = C::TN, NT = WC / 16;     // NT: column tiles per wave
    constexpr int NTHR = C::NTHR;
    constexpr int H2_RSB = C::RSB;
    constexpr int NAE = TZ * H2_KC, NBE = H2_KC * TN / 2;   // A elements / matrix pairs per chunk
    constexpr int NA = (NAE + NTHR - 1) / NTHR;             // ... per thread
    constexpr int NB = (NBE + NTHR - 1) / NTHR;
    extern __shared__ double hk_lds[];
    const int job = blockIdx.z;
    const cplx *__restrict__ in = J.in[job];
    const cplx *__restrict__ in2 = (PAIRED || DUAL) ? Pr.in2[job] : nullptr;
    const double psgn = PAIRED ? Pr.sgn[job] : 0.;
    const double *__restrict__ mat2 = DUAL ? Sc.sk[job] : nullptr;
    cplx *__restrict__ out2 = DUAL ? (cplx *)const_cast<double *>(Sc.fz[job]) : nullptr;
    cplx *__restrict__ out = J.out[job];
    const double *__restrict__ mat = J.mat[job];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wrow = wave / WN;                             // 16-row block of this wave
    const int wcol = WC * (wave % WN);                      // first column of this wave in the tile
    const int zb = blockIdx.x * TZ, n0 = blockIdx.y * TN;
    const double *sk = SCALED ? Sc.sk[job] : nullptr;

    double4_t acc_re[NT], acc_im[NT];
    double4_t acc2_re[DUAL ? NT : 1], acc2_im[DUAL ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        acc_re[t] = (double4_t){0., 0., 0., 0.};
        acc_im[t] = (double4_t){0., 0., 0., 0.};
    }
#pragma unroll
    for (int t = 0; t < (DUAL ? NT : 1); t++) {
        acc2_re[t] = (double4_t){0., 0., 0., 0.};
        acc2_im[t] = (double4_t){0., 0., 0., 0.};
    }
    // two register stages: the loads of chunk c+2 are issued before the MFMAs of chunk c (the
    // loads of chunk c+1 were issued one chunk earlier and are written to LDS after them), so a
    // single resident workgroup per CU keeps two chunks of global traffic in flight
    double2 ra0[NA], ra1[NA];
    double2 rb0[NB], rb1[NB];
    // FULL: the partner (PAIRED) and the column scale (SCALED) of a stage, combined when the stage goes to LDS
    double2 rw0[(FULL && PAIRED) ? NA : 1], rw1[(FULL && PAIRED) ? NA : 1];
    double rs0[(FULL && SCALED) ? NA : 1], rs1[(FULL && SCALED) ? NA : 1];
    const int nchunks = (Nr + H2_KC - 1) / H2_KC;
    const int ntot = (DUAL && in2) ? 2 * nchunks : nchunks;
    // FULL (Nr a multiple of two K chunks and of the tile width, every SCALED job with its column scale): the
    // loads of a chunk are STRAIGHT-LINE code - rows beyond Nz and chunks beyond the last one read a clamped
    // address (their sums are never stored / they are never written to LDS) -, so the compiler knows how many
    // loads are in flight at every point and the wait in front of the LDS writes of chunk c + 1 is
    // s_waitcnt vmcnt(loads of the younger chunk).  With a branch around a load (the general path
    // below) that wait is vmcnt(0): it also waits for the chunk requested a moment ago, and the register
    // stages hide nothing (round 6: MFMA pipe 59 % busy at 4416 x 256 whatever the workgroup shape).
    const cplx *__restrict__ pin2 = (PAIRED && in2) ? in2 : in;
    const double phalf = (PAIRED && in2) ? 0.5 : 1.0, psg = (PAIRED && in2) ? psgn : 0.;
    auto gload = [&](int c, double2 *ra, double2 *rb, double2 *rw, double *rs) {
        if constexpr (FULL) {
            const int cl = min(c, ntot - 1);
            const bool second = DUAL && cl >= nchunks;
            const int k0 = (second ? cl - nchunks : cl) * H2_KC;
            const cplx *__restrict__ src = second ? in2 : in;
            const double *__restrict__ mm = second ? mat2 : mat;
#pragma unroll
            for (int j = 0; j < NA; j++) {
                const int idx = j * NTHR + tid;
                const int row = idx >> 4, kk = idx & 15;
                const long o = (long)min(zb + row, Nz - 1) * irs + (k0 + kk);
                // (raw values: the combination below needs them, i.e. would wait for them here)
                ra[j] = *(const double2 *)(src + o);
                if (PAIRED) rw[j] = *(const double2 *)(pin2 + o);
                if (SCALED) rs[j] = sk[k0 + kk];
            }
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int idx = j * NTHR + tid;
                const int kr = idx / (TN / 2), nn = 2 * (idx % (TN / 2));
                const double *mrow = mm + (long)min(k0 + kr, Nr - 1) * Nr + (n0 + nn);
                const double2 v = *(const double2 *)mrow;
                rb[j] = v;
            }
            return;
        }
        const bool second = DUAL && c >= nchunks;
        const int k0 = (second ? c - nchunks : c) * H2_KC;
        const cplx *__restrict__ src = second ? in2 : in;
        const double *__restrict__ mm = second ? mat2 : mat;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int idx = j * NTHR + tid;
            const int row = idx >> 4, kk = idx & 15;        // 16 lanes = 256 contiguous bytes of a row
            const int zz = zb + row, k = k0 + kk;
            double2 v = make_double2(0., 0.);
            if ((NAE % NTHR == 0 || idx < NAE) && zz < Nz && k < Nr) {
                v = *(const double2 *)(src + (long)zz * irs + k);
                if (PAIRED && in2) {
                    // numba_rt_to_pm: p = 0.5 (r - i t), m = 0.5 (r + i t)
                    const double2 w_ = *(const double2 *)(in2 + (long)zz * irs + k);
                    v.x = 0.5 * (v.x - psgn * w_.y);
                    v.y = 0.5 * (v.y + psgn * w_.x);
                }
                if (SCALED && sk) { const double s_ = sk[k]; v.x *= s_; v.y *= s_; }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int idx = j * NTHR + tid;                 // pair index: TN/2 pairs per k row
            const int kr = idx / (TN / 2), nn = 2 * (idx % (TN / 2));
            const int k = k0 + kr, n = n0 + nn;
            double2 v = make_double2(0., 0.);
            if ((NBE % NTHR == 0 || idx < NBE) && k < Nr) {
                const double *mrow = mm + (long)k * Nr;
                if (n + 1 < Nr && ((Nr & 1) == 0)) v = *(const double2 *)(mrow + n);
                else { if (n < Nr) v.x = mrow[n]; if (n + 1 < Nr) v.y = mrow[n + 1]; }
            }
            rb[j] = v;
        }
    };
    auto lstore = [&](int buf, const double2 *ra, const double2 *rb, const double2 *rw, const double *rs) {
        double *A = hk_lds + buf * (C::ABUF + C::BBUF);
        double *B = A + C::ABUF;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int idx = j * NTHR + tid;
            double2 v = ra[j];
            if constexpr (FULL) {
                if (PAIRED) {
                    // numba_rt_to_pm: p = 0.5 (r - i t), m = 0.5 (r + i t); a job without a partner: 1.0 (v - 0 w)
                    const double2 w_ = rw[j];
                    v.x = phalf * (v.x - psg * w_.y);
                    v.y = phalf * (v.y + psg * w_.x);
                }
                if (SCALED) { const double s_ = rs[j]; v.x *= s_; v.y *= s_; }
            }
            if (NAE % NTHR == 0 || idx < NAE)
                *(double2 *)(A + (idx >> 4) * H2_RSA + 2 * (idx & 15)) = v;
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int idx = j * NTHR + tid;
            if (NBE % NTHR == 0 || idx < NBE)
                *(double2 *)(B + (idx / (TN / 2)) * H2_RSB + 2 * (idx % (TN / 2))) = rb[j];
        }
    };
        auto compute = [&](int c) {
        const int cur = c & 1;
        const double *A = hk_lds + cur * (C::ABUF + C::BBUF) + (wrow * 16 + li) * H2_RSA;
        const double *B = hk_lds + cur * (C::ABUF + C::BBUF) + C::ABUF + wcol + li;
        auto mma_chunk = [&](double4_t *are, double4_t *aim) {
#pragma unroll
            for (int s = 0; s < H2_KC / 4; s++) {
                const double2 a = *(const double2 *)(A + 2 * (4 * s + lk));
                const double *brow = B + (4 * s + lk) * H2_RSB;
                double b[NT];
#pragma unroll
                for (int t = 0; t < NT; t++) b[t] = brow[16 * t];
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    are[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b[t], are[t], 0, 0, 0);
                    aim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b[t], aim[t], 0, 0, 0);
                }
            }
        };
        if (DUAL && c >= nchunks) mma_chunk(acc2_re, acc2_im);
        else mma_chunk(acc_re, acc_im);
    };
    // Two register stages: chunk c + 2 is requested in front of the MFMAs of chunk c; chunk c + 1 - requested one
    // chunk of MFMA work earlier - goes to the other LDS buffer behind them; one barrier per chunk.
    gload(0, ra0, rb0, rw0, rs0);
    lstore(0, ra0, rb0, rw0, rs0);
    if constexpr (FULL) {
        // an even number of chunks, no branch in the loop: the requests beyond the last chunk re-read it, the
        // last LDS write lands in the buffer nobody reads any more
        gload(1, ra1, rb1, rw1, rs1);
        __syncthreads();
        // (sched_barrier: the requests stay in front of the MFMAs and the LDS writes behind them - left alone
        // the scheduler writes chunk c + 1 to LDS early in chunk c, i.e. waits for it half a chunk after its
        // request)
#ifndef HK_KNOCK
#define HK_KNOCK 0             // timing experiments (tools/variant.sh): 1 no requests in the loop, 2 nor LDS writes, 3 nor barriers
#endif
        for (int c = 0; c < ntot; c += 2) {
            if (HK_KNOCK < 1) gload(c + 2, ra0, rb0, rw0, rs0);
            __builtin_amdgcn_sched_barrier(0);
            compute(c);
            __builtin_amdgcn_sched_barrier(0);
            if (HK_KNOCK < 2) lstore(1, ra1, rb1, rw1, rs1);
            if (HK_KNOCK < 3) __syncthreads();
            if (HK_KNOCK < 1) gload(c + 3, ra1, rb1, rw1, rs1);
            __builtin_amdgcn_sched_barrier(0);
            compute(c + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (HK_KNOCK < 2) lstore(0, ra0, rb0, rw0, rs0);
            if (HK_KNOCK < 3) __syncthreads();
        }
    } else {
        if (ntot > 1) gload(1, ra1, rb1, rw1, rs1);
        __syncthreads();
        for (int c = 0; c < ntot; c += 2) {
            if (c + 2 < ntot) gload(c + 2, ra0, rb0, rw0, rs0);
            compute(c);
            if (c + 1 < ntot) lstore(1, ra1, rb1, rw1, rs1);
            __syncthreads();
            if (c + 1 >= ntot) break;
            if (c + 3 < ntot) gload(c + 3, ra1, rb1, rw1, rs1);
            compute(c + 1);
            if (c + 2 < ntot) lstore(0, ra0, rb0, rw0, rs0);
            __syncthreads();
        }
    }
    const bool pair = DUAL && in2;
    const int z0 = zb + wrow * 16;
    const double *fz = SCALED ? Sc.fz[job] : nullptr;
    const double *fr = SCALED ? Sc.fr[job] : nullptr;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + wcol + 16 * t + li;
        if (n >= Nr) continue;
        double cn = alpha;
        if (SCALED && fr) cn *= fr[n];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int zz = z0 + lk + 4 * r;
            if (zz < Nz) {
                double cz = cn;
                if (SCALED && fz) cz = fz[zz] * cn;     // fz[iz]*fr[ir]*F as in numba_filter_*
                if (pair) {
                    const double pr = cz * acc_re[t][r], pi = cz * acc_im[t][r];
                    const double mr = cz * acc2_re[DUAL ? t : 0][r], mi = cz * acc2_im[DUAL ? t : 0][r];
                    *(double2 *)(out + (long)zz * ors + n) = make_double2(pr + mr, pi + mi);
                    *(double2 *)(out2 + (long)zz * ors + n) = make_double2(-(pi - mi), pr - mr);
                } else {
                    *(double2 *)(out + (long)zz * ors + n) =
                        make_double2(cz * acc_re[t][r], cz * acc_im[t][r]);
                }
            }
        }
    }
}

template <bool SCALED, bool PAIRED, bool DUAL, int WC, int WN, int WZ, int WPE, bool FULL>
static int launch_tile_(const HankelJobs &J, const HankelScales &Sc, const HankelPairs &Pr, int nj, long irs,
                   long ors, double alpha, int Nz, int Nr, hipStream_t s)
{
    using C = H2Cfg<WC, WN, WZ>;
    auto kern = k_hankel<SCALED, PAIRED, DUAL, WC, WN, WZ, WPE, FULL>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)C::LDS_BYTES);
        if (e1 != hipSuccess) return check(e1, "fb_hankel(attr)");
        attr_done = true;
    }
    dim3 grid((Nz + C::TZ - 1) / C::TZ, (Nr + C::TN - 1) / C::TN, nj);
    hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), C::LDS_BYTES, s, J, Sc, Pr, irs, ors, alpha, Nz, Nr);
    return check(hipGetLastError(), "fb_hankel");
}

// straight-line loads (FULL) where the sizes allow it, see gload
template <bool SCALED, bool PAIRED, bool DUAL, int WC, int WN, int WZ, int WPE>
static int launch_tile(const HankelJobs &J, const HankelScales &Sc, const HankelPairs &Pr, int nj, long irs,
                   long ors, double alpha, int Nz, int Nr, hipStream_t s)
{
    bool full = (Nr % (2 * H2_KC) == 0) && (Nr % (WC * WN) == 0) && Nz >= 1;
    if (SCALED)
        for (int j = 0; j < nj; j++) full = full && Sc.sk[j] != nullptr;
#ifdef FB_HANKEL_TILE_PROBE
    static const int nofull = getenv("FBPIC_AMD_HANKEL_NOFULL") ? atoi(getenv("FBPIC_AMD_HANKEL_NOFULL")) : 0;
    if (nofull) full = false;
#endif
    if (full) return launch_tile_<SCALED, PAIRED, DUAL, WC, WN, WZ, WPE, true>(J, Sc, Pr, nj, irs, ors, alpha, Nz, Nr, s);
    return launch_tile_<SCALED, PAIRED, DUAL, WC, WN, WZ, WPE, false>(J, Sc, Pr, nj, irs, ors, alpha, Nz, Nr, s);
}

static int launch(int njobs, const void *const *in, long irs, void *const *out, long ors,
                  const double *const *mat, const double *const *sk, const double *const *fz,
                  const double *const *fr, double alpha, int Nz, int Nr, hipStream_t s,
                  const void *const *in2 = nullptr, const double *pair_sign = nullptr,
                  const double *const *mat2 = nullptr, void *const *out2 = nullptr)
{
    const bool dual = mat2 != nullptr;             // (p, m) -> (r, t) on the output side
    const bool scaled = !dual && (sk || fz || fr || in2);
    const bool paired = !dual && in2 != nullptr;
    for (int j0 = 0; j0 < njobs; j0 += HK_MAXJOBS) {
        const int nj = njobs - j0 < HK_MAXJOBS ? njobs - j0 : HK_MAXJOBS;
        HankelJobs J;
        HankelScales Sc;
        HankelPairs Pr;
        for (int j = 0; j < HK_MAXJOBS; j++) {
            const bool v = j < nj;
            Pr.in2[j] = (v && (paired || dual)) ? (const cplx *)in2[j0 + j] : nullptr;
            Pr.sgn[j] = (v && (paired || dual) && pair_sign) ? pair_sign[j0 + j] : 0.;
            J.in[j] = v ? (const cplx *)in[j0 + j] : nullptr;
            J.out[j] = v ? (cplx *)out[j0 + j] : nullptr;
            J.mat[j] = v ? mat[j0 + j] : nullptr;
            Sc.sk[j] = (v && dual) ? mat2[j0 + j] : ((v && sk) ? sk[j0 + j] : nullptr);
            Sc.fz[j] = (v && dual) ? (const double *)out2[j0 + j] : ((v && fz) ? fz[j0 + j] : nullptr);
            Sc.fr[j] = (v && fr) ? fr[j0 + j] : nullptr;
        }
        // the dual (two accumulator sets) variant always uses the 32 x 64 workgroup: at 4416 x 256
        // the 64 x 128 one spills (128 VGPRs at 16 waves: 0.33 -> 0.79 ms) and a 32 x 128 one
        // with 8 waves is no faster (0.35 ms)
        int r;
        // Tile choice (measured, see the comment on top of k_hankel): 64 x 128 with 16 waves when
        // that still gives every CU two workgroups' worth of tiles (2048 x 512, 4096 x 256: the
        // matrix is re-read 4x less often), else 32 x 64 with 4 waves, three workgroups
        // co-resident per CU (the headline size)
        const long wg_big = (long)((Nz + 63) / 64) * ((Nr + 127) / 128) * nj;
        const bool big = wg_big >= 2 * 256;
#define H2(SC, PA, DU, WN_, WZ_, WP) launch_tile<SC, PA, DU, 32, WN_, WZ_, WP>(J, Sc, Pr, nj, irs, ors, alpha, Nz, Nr, s)
        // (round 6, with the straight-line loads: 32 x 128 with 8 waves beats 64 x 128 at Nr = 256 - 195 against
        // 213 us for the 8 forward transforms of 4416 x 256, 308 against 315 for 12 -, equal at Nr = 512)
        const bool mid = big && Nr < 512;
#define H2V(SC, PA) (mid ? H2(SC, PA, false, 4, 2, 2) : big ? H2(SC, PA, false, 4, 4, 4) : H2(SC, PA, false, 2, 2, 3))
#ifdef FB_HANKEL_TILE_PROBE
        // developer A/B builds (tools/variant.sh ... -DFB_HANKEL_TILE_PROBE): other workgroup shapes by environment
        static const int probe = getenv("FBPIC_AMD_HANKEL_TILE") ? atoi(getenv("FBPIC_AMD_HANKEL_TILE")) : 0;
#define H2W(SC, PA, DU, WC_, WN_, WZ_, WP) launch_tile<SC, PA, DU, WC_, WN_, WZ_, WP>(J, Sc, Pr, nj, irs, ors, alpha, Nz, Nr, s)
        if (probe && !dual && !paired && !scaled) {
            switch (probe) {
            case 1: return H2W(false, false, false, 64, 2, 4, 2);      // 64 x 128, 8 waves, wave tile 16 x 64
            case 2: return H2W(false, false, false, 64, 4, 4, 4);      // 64 x 256, 16 waves
            case 3: return H2W(false, false, false, 64, 4, 2, 2);      // 32 x 256, 8 waves
            case 4: return H2W(false, false, false, 32, 4, 2, 2);      // 32 x 128, 8 waves
            case 5: return H2W(false, false, false, 32, 4, 4, 4);      // 64 x 128, 16 waves (the big default)
            case 6: return H2W(false, false, false, 32, 2, 2, 3);      // 32 x 64, 4 waves (the small default)
            case 7: return H2W(false, false, false, 32, 8, 2, 4);      // 32 x 256, 16 waves
            case 8: return H2W(false, false, false, 64, 4, 1, 1);      // 16 x 256, 4 waves
            case 9: return H2W(false, false, false, 32, 4, 3, 3);      // 48 x 128, 12 waves
            case 11: return H2W(false, false, false, 32, 2, 3, 3);     // 48 x 64, 6 waves
            case 12: return H2W(false, false, false, 32, 2, 4, 2);     // 64 x 64, 8 waves
            }
        }
        if (probe && dual) {
            switch (probe) {
            case 1: return H2W(false, false, true, 32, 4, 2, 2);       // 32 x 128, 8 waves
            case 2: return H2W(false, false, true, 32, 2, 4, 2);       // 64 x 64, 8 waves
            case 3: return H2W(false, false, true, 32, 8, 1, 2);       // 16 x 256, 8 waves
            case 4: return H2W(false, false, true, 32, 8, 2, 4);       // 32 x 256, 16 waves
            case 5: return H2W(false, false, true, 32, 4, 4, 4);       // 64 x 128, 16 waves (spills)
            case 6: return H2W(false, false, true, 32, 2, 2, 2);       // default
            }
        }
#undef H2W
#endif
        if (dual) r = H2(false, false, true, 2, 2, 2);
        else if (paired) r = H2V(true, true);
        else if (scaled) r = H2V(true, false);
        else r = H2V(false, false);
#undef H2V
#undef H2
        if (r) return r;
    }
    return 0;
}

}  // namespace fb

using namespace fb;

extern "C" int fb_hankel(int njobs, const void *const *in, long in_row_stride, void *const *out,
                         long out_row_stride, const double *const *mat, double alpha, int Nz,
                         int Nr, void *stream)
{
    if (njobs <= 0) return 0;
    return launch(njobs, in, in_row_stride, out, out_row_stride, mat, nullptr, nullptr, nullptr,
                  alpha, Nz, Nr, (hipStream_t)stream);
}

extern "C" int fb_hankel_scaled(int njobs, const void *const *in, long in_row_stride,
                                void *const *out, long out_row_stride, const double *const *mat,
                                const double *const *in_col_scale, const double *const *out_row_scale,
                                const double *const *out_col_scale, double alpha, int Nz, int Nr,
                                void *stream)
{
    if (njobs <= 0) return 0;
    return launch(njobs, in, in_row_stride, out, out_row_stride, mat, in_col_scale, out_row_scale,
                  out_col_scale, alpha, Nz, Nr, (hipStream_t)stream);
}

extern "C" int fb_hankel_rt_to_pm_scaled(int njobs, const void *const *in, const void *const *in2,
                                         const double *pair_sign, long in_row_stride,
                                         void *const *out, long out_row_stride,
                                         const double *const *mat, const double *const *in_col_scale,
                                         const double *const *out_row_scale,
                                         const double *const *out_col_scale, double alpha, int Nz,
                                         int Nr, void *stream)
{
    if (njobs <= 0) return 0;
    if (!in2 || !pair_sign) { set_error("fb_hankel_rt_to_pm_scaled", "in2 and pair_sign are required"); return -1; }
    return launch(njobs, in, in_row_stride, out, out_row_stride, mat, in_col_scale, out_row_scale,
                  out_col_scale, alpha, Nz, Nr, (hipStream_t)stream, in2, pair_sign);
}

extern "C" int fb_hankel_pm_to_rt(int njobs, const void *const *in, const void *const *in2,
                                  long in_row_stride, void *const *out, void *const *out2,
                                  long out_row_stride, const double *const *mat,
                                  const double *const *mat2, double alpha, int Nz, int Nr,
                                  void *stream)
{
    if (njobs <= 0) return 0;
    if (!in2 || !out2 || !mat2) {
        set_error("fb_hankel_pm_to_rt", "in2, out2 and mat2 are required");
        return -1;
    }
    for (int j = 0; j < njobs; j++)
        if (in2[j] && (!mat2[j] || !out2[j])) {
            set_error("fb_hankel_pm_to_rt", "pair job without mat2 / out2");
            return -1;
        }
    return launch(njobs, in, in_row_stride, out, out_row_stride, mat, nullptr, nullptr, nullptr,
                  alpha, Nz, Nr, (hipStream_t)stream, in2, nullptr, mat2, out2);
}
