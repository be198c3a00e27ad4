// Shared helpers for the gfx950 kernels of libfbpic_amd.so.
// Whole library is compiled with -ffp-contract=off: cell indices must be bit-identical
// to the reference CPU path (invdr*(r-rmin)-0.5 must not fuse into an FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fbpic_amd.h"

namespace fb {

void set_error(const char *where, const char *what);
int check(hipError_t e, const char *where);

struct cplx { double re, im; };

// by-value pointer tables passed as kernel arguments
struct Ptrs48 { void *p[48]; };
struct CPtrs48 { const void *p[48]; };
struct Ptrs16 { double *p[16]; };
struct CPtrs16 { const double *p[16]; };

// Streaming launch geometry: 256-thread blocks, capped grid, grid-stride loop
// (256 CUs x 8 blocks).
inline int stream_grid(long n, int block = 256, int max_blocks = 256 * 8)
{
    long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

__device__ __forceinline__ double m1pow(int m) { return (m & 1) ? -1. : 1.; }

}  // namespace fb

#define FB_CHECK_LAUNCH(where) return fb::check(hipGetLastError(), where)
