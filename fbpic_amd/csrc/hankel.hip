// Discrete Hankel transform along r as an fp64 MFMA GEMM on gfx950.
//
// out[iz, n] = alpha * sum_k in[iz, k] * mat[k, n]   (complex row x real (Nr,Nr) matrix)
//
// The reference splits the complex (Nz,Nr) array into a real (2Nz,Nr) one, calls cuBLAS
// dgemm and re-interleaves (fbpic/fields/spectral_transform/hankel.py:196-205).  Here the
// split is free: a lane loads one complex element (16 B) and feeds its real part to one
// v_mfma_f64_16x16x4_f64 accumulator chain and its imaginary part to a second chain that
// shares the same B operand; the two accumulators hold re/im of the same output element
// in the same lane/register, so the store is again one 16-B complex write.
//
// Tiling: workgroup = 4 waves = 64 z rows x 64 output columns; wave w owns rows
// [16w,16w+16) and 4 column sub-tiles -> 8 independent accumulator chains (64 VGPRs),
// enough to keep the 64-cycle f64 MFMA pipe full from one wave per SIMD.  B (the Hankel
// matrix, <= 2 MiB) is read through L1/L2: one f64 MFMA consumes 1 KiB of operands per
// 64 cycles, far below the cache bandwidth, so no LDS staging is needed.
// Jobs (field x mode) are batched along gridDim.z so that one launch carries enough
// tiles to fill 256 CUs.
//
// Fragment layouts (cdna_hip_programming.md section 3, f64 row formula):
//   A: lane l -> A[i = l & 15][k = l >> 4]      B: lane l -> B[k = l >> 4][j = l & 15]
//   D: reg r of lane l -> D[i = (l >> 4) + 4 r][j = l & 15]
#include "fb_common.h"

namespace fb {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct HankelJobs {
    const cplx *in[48];
    cplx *out[48];
    const double *mat[48];
};

__global__ __launch_bounds__(256) void k_hankel(HankelJobs J, long irs, long ors, double alpha,
                                                int Nz, int Nr)
{
    const int job = blockIdx.z;
    const cplx *__restrict__ in = J.in[job];
    cplx *__restrict__ out = J.out[job];
    const double *__restrict__ mat = J.mat[job];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int z0 = blockIdx.x * 64 + wave * 16;
    const int n0 = blockIdx.y * 64;
    if (z0 >= Nz) return;   // whole wave out of range (no barriers in this kernel)

    double4_t acc_re[4], acc_im[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        acc_re[t] = (double4_t){0., 0., 0., 0.};
        acc_im[t] = (double4_t){0., 0., 0., 0.};
    }
    const int za = z0 + li;
    const bool za_ok = za < Nz;
    const cplx *arow = in + (long)(za_ok ? za : 0) * irs;
    bool n_ok[4];
    int ncol[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { ncol[t] = n0 + 16 * t + li; n_ok[t] = ncol[t] < Nr; if (!n_ok[t]) ncol[t] = 0; }

    const int ksteps = (Nr + 3) / 4;
#pragma unroll 2
    for (int ks = 0; ks < ksteps; ks++) {
        const int k = ks * 4 + lk;
        const bool k_ok = k < Nr;
        const int kc = k_ok ? k : 0;
        double2 a = *(const double2 *)(arow + kc);
        if (!(k_ok && za_ok)) { a.x = 0.; a.y = 0.; }
        const double *brow = mat + (long)kc * Nr;
        double b[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { b[t] = brow[ncol[t]]; if (!(k_ok && n_ok[t])) b[t] = 0.; }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            acc_re[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b[t], acc_re[t], 0, 0, 0);
            acc_im[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b[t], acc_im[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int n = n0 + 16 * t + li;
        if (n >= Nr) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int zz = z0 + lk + 4 * r;
            if (zz < Nz)
                *(double2 *)(out + (long)zz * ors + n) =
                    make_double2(alpha * acc_re[t][r], alpha * acc_im[t][r]);
        }
    }
}

}  // namespace fb

using namespace fb;

extern "C" int fb_hankel(int njobs, const void *const *in, long in_row_stride, void *const *out,
                         long out_row_stride, const double *const *mat, double alpha, int Nz,
                         int Nr, void *stream)
{
    if (njobs <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    for (int j0 = 0; j0 < njobs; j0 += 48) {
        int nj = njobs - j0 < 48 ? njobs - j0 : 48;
        HankelJobs J;
        for (int j = 0; j < 48; j++) {
            J.in[j] = j < nj ? (const cplx *)in[j0 + j] : nullptr;
            J.out[j] = j < nj ? (cplx *)out[j0 + j] : nullptr;
            J.mat[j] = j < nj ? mat[j0 + j] : nullptr;
        }
        dim3 grid((Nz + 63) / 64, (Nr + 63) / 64, nj);
        hipLaunchKernelGGL(k_hankel, grid, dim3(256), 0, s, J, in_row_stride, out_row_stride,
                           alpha, Nz, Nr);
        int r = check(hipGetLastError(), "fb_hankel");
        if (r) return r;
    }
    return 0;
}
