// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand/result lane layout and issue rate.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma4_probe.hip -o tools/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_layout(int *out)   // out[a*64 + b] = bitmask-less: lane of D that is non-zero (or -1), block is implied
{
    const int lane = threadIdx.x;
    for (int a = 0; a < 64; a++)
        for (int b = 0; b < 64; b++) {
            double av = (lane == a) ? 1. : 0., bv = (lane == b) ? 1. : 0.;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0., 0, 0, 0);
            unsigned long long m = __ballot(d != 0.);
            if (lane == 0) out[a * 64 + b] = m ? (__builtin_ctzll(m) | (__builtin_popcountll(m) << 8)) : -1;
        }
}

__global__ void k_rate(double *out, int iters, long long *clk)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

__global__ void k_rate_dep(double *out, int iters, long long *clk)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, c0 = 0;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCHAIN>
__global__ void k_rate16(double *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 c[NCHAIN];
    for (int j = 0; j < NCHAIN; j++) c[j] = (d4){0., 0., 0., 0.};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NCHAIN; j++) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
    }
    double s = 0.;
    for (int j = 0; j < NCHAIN; j++) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCHAIN>
static void rate16(double *o, int waves_per_block, int nblocks, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate16<NCHAIN><<<nblocks, 64 * waves_per_block>>>(o, iters);
    hipEventRecord(e0);
    k_rate16<NCHAIN><<<nblocks, 64 * waves_per_block>>>(o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)nblocks * waves_per_block * iters * NCHAIN * 2048.0;
    printf("16x16x4 f64: %d chains, %d waves/block, %d blocks: %.3f ms  %.1f TFLOP/s\n", NCHAIN,
           waves_per_block, nblocks, ms, flops / (ms * 1e-3) / 1e12);
}

int main()
{
    {
        double *o16; hipMalloc(&o16, 4096 * 512 * 8);
        rate16<1>(o16, 4, 256, 4000);
        rate16<2>(o16, 4, 256, 4000);
        rate16<4>(o16, 4, 256, 2000);
        rate16<8>(o16, 4, 256, 1000);
        rate16<2>(o16, 8, 256, 4000);
        rate16<2>(o16, 4, 512, 4000);
        rate16<4>(o16, 4, 1024, 2000);
        rate16<2>(o16, 8, 512, 4000);
    }
    int *d; hipMalloc(&d, 64 * 64 * sizeof(int));
    k_layout<<<1, 64>>>(d);
    std::vector<int> h(64 * 64);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    // for every A lane, list which B lanes it pairs with and the D lane
    for (int a = 0; a < 64; a++) {
        printf("A%02d:", a);
        for (int b = 0; b < 64; b++) if (h[a * 64 + b] >= 0) printf(" B%02d->D%02d(n%d)", b, h[a * 64 + b] & 255, h[a * 64 + b] >> 8);
        printf("\n");
    }
    double *o; long long *c; hipMalloc(&o, 1024 * 256 * 8); hipMalloc(&c, 8);
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_rate<<<1024, 256>>>(o, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        double ninstr = 1024.0 * 4 * iters * 8;       // wave-instructions
        printf("independent: %.3f ms, %.1f TFLOP/s, s_memtime ticks per instr (one wave) %.2f\n", ms,
               ninstr * 512 / (ms * 1e-3) / 1e12, (double)hc / (iters * 8.0));
        hipEventRecord(e0);
        k_rate_dep<<<1024, 256>>>(o, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("dependent  : %.3f ms, %.1f TFLOP/s, ticks per instr %.2f\n", ms,
               1024.0 * 4 * iters * 4 * 512 / (ms * 1e-3) / 1e12, (double)hc / (iters * 4.0));
    }
    return 0;
}
