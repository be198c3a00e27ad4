// How many workgroups with X bytes of dynamic LDS co-reside on a CU of gfx950?  Every workgroup
// (64 threads) spins for ~200 us; the launch has 256 x N workgroups; elapsed ~ ceil(N / resident) x 200 us.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_probe tools/lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long cycles, double *out)
{
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = lds[1];
}
int main(int argc, char **argv)
{
    double *out; (void)hipMalloc(&out, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long ticks = 20000;     // wall_clock64 runs at 100 MHz: 200 us
    // sizes from the command line (bytes), else the round-2 list
    int sizes[64] = {16384, 20480, 26624, 28672, 30720, 32768, 36864, 38912, 40960, 47104, 51200, 53248, 53880, 54272, 55296, 57344, 65536};
    int nsz = 17;
    if (argc > 1) { nsz = 0; for (int a = 1; a < argc && nsz < 64; a++) sizes[nsz++] = atoi(argv[a]); }
    for (int q = 0; q < nsz; q++) {
        const int sz = sizes[q];
        (void)hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        int resident = 0;
        for (int n = 1; n <= 20; n++) {
            hipLaunchKernelGGL(spin, dim3(256 * n), dim3(64), sz, 0, ticks, out);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(256 * n), dim3(64), sz, 0, ticks, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < 0.3f) resident = n; else break;
        }
        printf("LDS %6d B per workgroup: %2d workgroups per CU run concurrently (%d KB)\n", sz, resident, resident * sz / 1024);
    }
    return 0;
}
