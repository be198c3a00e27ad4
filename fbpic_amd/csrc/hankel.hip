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
// Tiling: workgroup = 4 waves = 64 z rows x 64 output columns; wave w owns rows
// [16w,16w+16) and 4 column sub-tiles -> 8 independent accumulator chains (64 VGPRs): one
// wave per SIMD keeps the 64-cycle f64 MFMA pipe busy as long as its operands arrive.
// Operands are double-buffered through LDS (see k_hankel): fragment-shaped loads straight
// from global memory touch 16 half-used cache lines per instruction and saturate the
// texture-address path long before the MFMA pipe (measured: 12% of peak).
// Jobs (field x mode) are batched along gridDim.z so one launch fills 256 CUs.
//
// Fragment layouts (cdna_hip_programming.md section 3, f64 row formula):
//   A: lane l -> A[i = l & 15][k = l >> 4]      B: lane l -> B[k = l >> 4][j = l & 15]
//   D: reg r of lane l -> D[i = (l >> 4) + 4 r][j = l & 15]
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

constexpr int HK_KC = 32;                 // k-chunk staged per pipeline stage
constexpr int HK_RSA = HK_KC * 2 + 2;     // A panel row stride in doubles (16-B pad: rows rotate banks)
constexpr int HK_RSB = 64 + 16;           // B panel row stride in doubles (+128 B: k rows alternate bank halves)
constexpr int HK_TZ = 32;                 // z rows per workgroup
constexpr int HK_ABUF = HK_TZ * HK_RSA;   // doubles per A buffer
constexpr int HK_BBUF = HK_KC * HK_RSB;   // doubles per B buffer
constexpr size_t HK_LDS_BYTES = (size_t)2 * (HK_ABUF + HK_BBUF) * 8;

// Workgroup = 4 waves (2 x 2) = 32 z rows x 64 output columns; wave (wz, wn) owns rows
// [16 wz, +16) and columns [32 wn, +32): 2 column sub-tiles x (re, im) = 4 accumulator
// chains.  The small tile keeps the launch balanced over 256 CUs (C2: 768 workgroups for
// E+B, 3 per CU, two co-resident so one's load latency hides under the other's MFMAs).
// Measured limits (rocprofv3 + tools/hankel_probe.py): 55-58 % MfmaUtil at 2048 x 512 and
// 4096 x 256, 30-40 % at 1024 x 128 where a launch holds only 2-12 transforms of 67 MFLOP
// (LdsUtil 7-20 %, no bank conflicts, L2 hit rate 83 %).  Variants tried and discarded
// because they changed nothing at the headline size: all operands of a chunk requested up
// front, XCD-aware tile order, matrix column block stationary in LDS with fragment-shaped
// global loads of the input (with and without split-K across waves), matrix columns
// stationary in registers with the input tile streamed through LDS by a persistent
// workgroup.  All land on ~2 us per transform = its memory phase (~1 us for 4 MB in + out,
// measured with the MFMAs removed) plus its MFMA phase (~0.9 us): at this size the two do
// not overlap whatever the tiling.  A register-only loop of the same MFMA sustains 66-72
// TFLOP/s (tools/mfma4_probe.hip), so the instruction itself is not the limit.
// K is walked in chunks of 32:
//   global -> registers (coalesced: full 512-B row segments of the complex input and of the
//   matrix) for chunk c+1 is issued BEFORE the 32 MFMAs (2048 cycles) of chunk c, then
//   written to the other LDS buffer; one barrier per chunk.  MFMA operands come from LDS
//   (padded panels, conflict-free ds_read_b128 / ds_read_b64), so the vector-memory path
//   only sees coalesced traffic and each matrix element is fetched once per workgroup.
// DUAL (backward transform of a vector field, (p, m) -> (r, t) folded into the GEMM,
// spectral_transformer.py:89-155): a job with in2 != 0 computes BOTH p' = in . mat and
// m' = in2 . mat2 (K walked over the first product, then over the second, two accumulator
// sets - the same MFMA work as two plain jobs) and writes
//   out = p' + m' (the r slot)  and  out2 = i (p' - m') (the t slot).
// in2 comes from Pr.in2, mat2 travels in Sc.sk and out2 in Sc.fz; jobs with in2 == 0 are
// plain transforms (the z components).
template <bool SCALED, bool PAIRED, bool DUAL = false>
__global__ __launch_bounds__(256) void k_hankel(HankelJobs J, HankelScales Sc, HankelPairs Pr, long irs,
                                                long ors, double alpha, int Nz, int Nr)
{
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
    const int wz = wave >> 1, wn = wave & 1;
    const int zb = blockIdx.x * HK_TZ, n0 = blockIdx.y * 64;
    const double *sk = SCALED ? Sc.sk[job] : nullptr;

    double4_t acc_re[2], acc_im[2], acc2_re[2], acc2_im[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        acc_re[t] = (double4_t){0., 0., 0., 0.};
        acc_im[t] = (double4_t){0., 0., 0., 0.};
        acc2_re[t] = (double4_t){0., 0., 0., 0.};
        acc2_im[t] = (double4_t){0., 0., 0., 0.};
    }
    // staging registers: A chunk = 32 rows x 32 complex (4 per thread), B chunk = 32 x 64
    // doubles (8 per thread)
    double2 ra[4];
    double rb[8];
    const int nchunks = (Nr + HK_KC - 1) / HK_KC;
    // chunk index -> which of the two products it belongs to (DUAL), and its k offset
    auto gload = [&](int c) {
        const bool second = DUAL && c >= nchunks;
        const int k0 = (second ? c - nchunks : c) * HK_KC;
        const cplx *__restrict__ src = second ? in2 : in;
        const double *__restrict__ mm = second ? mat2 : mat;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = j * 256 + tid;
            const int row = idx >> 5, kk = idx & 31;
            const int zz = zb + row, k = k0 + kk;
            double2 v = make_double2(0., 0.);
            if (zz < Nz && k < Nr) {
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
        for (int j = 0; j < 8; j++) {
            const int idx = j * 256 + tid;
            const int kr = idx >> 6, nn = idx & 63;
            const int k = k0 + kr, n = n0 + nn;
            rb[j] = (k < Nr && n < Nr) ? mm[(long)k * Nr + n] : 0.;
        }
    };
    auto lstore = [&](int buf) {
        double *A = hk_lds + buf * (HK_ABUF + HK_BBUF);
        double *B = A + HK_ABUF;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = j * 256 + tid;
            *(double2 *)(A + (idx >> 5) * HK_RSA + 2 * (idx & 31)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int idx = j * 256 + tid;
            B[(idx >> 6) * HK_RSB + (idx & 63)] = rb[j];
        }
    };
    const int ntot = (DUAL && in2) ? 2 * nchunks : nchunks;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < ntot; c++) {
        const int cur = c & 1;
        if (c + 1 < ntot) gload(c + 1);
        const double *A = hk_lds + cur * (HK_ABUF + HK_BBUF) + (wz * 16 + li) * HK_RSA;
        const double *B = hk_lds + cur * (HK_ABUF + HK_BBUF) + HK_ABUF + 32 * wn;
        auto mma_chunk = [&](double4_t *are, double4_t *aim) {
#pragma unroll
            for (int s = 0; s < HK_KC / 4; s++) {
                const double2 a = *(const double2 *)(A + 2 * (4 * s + lk));
                const double *brow = B + (4 * s + lk) * HK_RSB + li;
                double b[2];
#pragma unroll
                for (int t = 0; t < 2; t++) b[t] = brow[16 * t];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    are[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b[t], are[t], 0, 0, 0);
                    aim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b[t], aim[t], 0, 0, 0);
                }
            }
        };
        if (DUAL && c >= nchunks) mma_chunk(acc2_re, acc2_im);
        else mma_chunk(acc_re, acc_im);
        if (c + 1 < ntot) lstore(cur ^ 1);
        __syncthreads();
    }
    const bool pair = DUAL && in2;
    const int z0 = zb + wz * 16;
    const double *fz = SCALED ? Sc.fz[job] : nullptr;
    const double *fr = SCALED ? Sc.fr[job] : nullptr;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int n = n0 + 32 * wn + 16 * t + li;
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
                    const double mr = cz * acc2_re[t][r], mi = cz * acc2_im[t][r];
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
        dim3 grid((Nz + HK_TZ - 1) / HK_TZ, (Nr + 63) / 64, nj);
        static bool attr_done = false;
        if (!attr_done) {
            const void *ks[4] = {(const void *)k_hankel<true, true>, (const void *)k_hankel<true, false>,
                                 (const void *)k_hankel<false, false>,
                                 (const void *)k_hankel<false, false, true>};
            for (int i = 0; i < 4; i++) {
                hipError_t e1 = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)HK_LDS_BYTES);
                if (e1 != hipSuccess) return check(e1, "fb_hankel(attr)");
            }
            attr_done = true;
        }
        if (dual)
            hipLaunchKernelGGL((k_hankel<false, false, true>), grid, dim3(256), HK_LDS_BYTES, s, J, Sc, Pr,
                               irs, ors, alpha, Nz, Nr);
        else if (paired)
            hipLaunchKernelGGL((k_hankel<true, true>), grid, dim3(256), HK_LDS_BYTES, s, J, Sc, Pr, irs, ors,
                               alpha, Nz, Nr);
        else if (scaled)
            hipLaunchKernelGGL((k_hankel<true, false>), grid, dim3(256), HK_LDS_BYTES, s, J, Sc, Pr, irs, ors,
                               alpha, Nz, Nr);
        else
            hipLaunchKernelGGL((k_hankel<false, false>), grid, dim3(256), HK_LDS_BYTES, s, J, Sc, Pr, irs, ors,
                               alpha, Nz, Nr);
        int r = check(hipGetLastError(), "fb_hankel");
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
