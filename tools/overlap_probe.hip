// Do fp64 MFMA and fp64 VALU instructions of different waves of one SIMD overlap on gfx950?
// mode 0: every wave runs MFMAs; mode 1: every wave runs FMAs; mode 2: even waves MFMA, odd waves
// FMA (same per-wave instruction counts as modes 0 / 1).  If the pipes overlap, t2 ~ max(t0, t1) / 1
// with half the waves each...  Reported: time per launch and the implied busy cycles.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/overlap_probe tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mode, int iters, double *out)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4) || (mode == 3 && wave < 4);
    const bool do_fma = mode == 1 || (mode == 2 && wave >= 4) || (mode == 4 && wave >= 4);
    const bool do_int = (mode == 5 || mode == 6) && wave >= 4;
    const bool do_mfma2 = mode == 5 && wave < 4;
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    if (do_int) {
        // 64 integer / fp32 VALU instructions per iteration (4 cycles each)
        unsigned y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;
        float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                y0 = y0 * 3u + 7u; y1 = (y1 ^ y0) + 5u; y2 = y2 * 5u + y1; y3 = (y3 + y2) ^ 9u;
                f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, f0); f2 = __builtin_fmaf(f2, 0.999f, f1); f3 = __builtin_fmaf(f3, 0.999f, f2);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = y0 + y1 + y2 + y3 + f0 + f1 + f2 + f3;
    }
    if (do_mfma || do_mfma2) {
        double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (do_fma) {
        double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters; i++) {
            // 64 FMAs per iteration = 256 cycles = 4 MFMA 16x16x4 (64 cycles each)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a);
                x3 = __builtin_fma(x3, b, a); x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a);
                x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
}

int main()
{
    double *out;
    hipMalloc(&out, 256 * 2048 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    // 256 CUs x 2 workgroups of 4 waves = 2 waves per SIMD
    // one workgroup of 8 waves per CU: waves w and w + 4 share SIMD w
    const char *names[7] = {"8 waves MFMA", "8 waves FMA", "waves 0-3 MFMA + 4-7 FMA", "waves 0-3 MFMA, 4-7 idle", "waves 0-3 idle, 4-7 FMA", "waves 0-3 MFMA + 4-7 int/fp32 VALU", "waves 0-3 idle, 4-7 int/fp32 VALU"};
    for (int wg = 1; wg <= 1; wg++)
        for (int mode = 0; mode < 7; mode++) {
            hipLaunchKernelGGL(k, dim3(256 * wg), dim3(512), 0, 0, mode, iters, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * wg), dim3(512), 0, 0, mode, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s): %.3f ms  (per wave: %d x 256 issue cycles = %.3f ms at 2.4 GHz)\n", mode,
                   names[mode], ms, iters, iters * 256 / 2.4e6);
        }
    return 0;
}
