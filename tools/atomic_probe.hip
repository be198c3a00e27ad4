// Probe: throughput of global fp64 atomic adds (no return) vs the address pattern of the 48
// active lanes of one instruction, for an HBM-resident (16 GiB) and an L2/MALL-resident
// (16 MiB, like the J grids) target.  Build: hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k_atom(double *buf, int iters, long nelem)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane >= 48) return;
    for (int it = 0; it < iters; it++) {
        // pseudo-random cell of this (wave, iteration)
        unsigned long h = (unsigned long)(wave * 2654435761u + it * 40503u);
        h ^= h >> 13; h *= 0x9E3779B97F4A7C15ul; h ^= h >> 29;
        long idx;
        if (MODE == 0) idx = (long)((h + (unsigned long)lane * 0x51ED27ul * 4099ul) % (unsigned long)nelem);   // 48 unrelated lines
        else if (MODE == 1) idx = (long)((h + (unsigned long)(lane >> 1) * 0x51ED27ul * 4099ul) % (unsigned long)(nelem - 2)) & ~1L | (lane & 1);
        else idx = (long)(h % (unsigned long)(nelem - 64)) / 16 * 16 + lane;   // 48 contiguous doubles (3 lines)
        atomicAdd(buf + idx, 1.0);
    }
}

template <int MODE> static void run(double *d, long nelem, const char *name)
{
    const int iters = 1000, blocks = 2048, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_atom<MODE><<<blocks, threads>>>(d, 10, nelem);
    hipEventRecord(e0);
    k_atom<MODE><<<blocks, threads>>>(d, iters, nelem);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninstr = (double)blocks * threads / 64 * iters;
    printf("%-40s %8.3f ms  %7.2f G atomics/s  %6.3f ns per wave-instruction\n", name, ms,
           ninstr * 48 / (ms * 1e-3) / 1e9, ms * 1e6 / ninstr);
}

int main()
{
    for (int big = 0; big < 2; big++) {
        const long nelem = big ? (1L << 31) : (1L << 21);     // 16 GiB or 16 MiB of doubles
        double *d;
        if (hipMalloc(&d, nelem * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(d, 0, nelem * 8);
        printf("target %s\n", big ? "16 GiB (HBM)" : "16 MiB (cache-resident)");
        run<0>(d, nelem, "48 lanes -> 48 unrelated lines");
        run<1>(d, nelem, "24 (re,im) pairs -> 24 lines");
        run<2>(d, nelem, "48 contiguous doubles -> 3 lines");
        hipFree(d);
    }
    return 0;
}
