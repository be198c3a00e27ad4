// Cell sort for gfx950: cell index per particle, stable radix sort by cell (rocPRIM),
// inclusive per-cell prefix sum, one-pass permutation of all particle attributes.
// Replaces fbpic/particles/utilities/cuda_sorting.py (Numba-CUDA + Thrust argsort).
#include <cstring>
#include "fb_common.h"
#include <rocprim/device/device_radix_sort.hpp>

namespace fb {

// cuda_sorting.py:21-88.  The expressions r_cell / z_cell are the same as in the
// gather and deposition kernels; with -ffp-contract=off, IEEE sqrt and exact ceil the
// index is bit-identical to the reference CPU arithmetic.
__global__ __launch_bounds__(256) void k_cell_index(long n, const double *__restrict__ x,
        const double *__restrict__ y, const double *__restrict__ z,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        int *__restrict__ cell_idx, int *__restrict__ sorted_idx)
{
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double xj = x[i], yj = y[i], zj = z[i];
        double rj = sqrt(xj * xj + yj * yj);
        double r_cell = invdr * (rj - rmin) - 0.5;
        double z_cell = invdz * (zj - zmin) - 0.5;
        int ir_upper = (int)ceil(r_cell);
        int iz_upper = (int)ceil(z_cell);
        if (ir_upper > Nr) ir_upper = Nr;
        if (iz_upper < 0) iz_upper += Nz;
        else if (iz_upper > Nz - 1) iz_upper -= Nz;
        sorted_idx[i] = (int)i;
        cell_idx[i] = ir_upper + iz_upper * (Nr + 1);
    }
}

// prefill_prefix_sum + incl_prefix_sum (cuda_sorting.py:124-190) in one pass over the
// sorted keys: prefix_sum[c] = number of particles with cell <= c.
__global__ __launch_bounds__(256) void k_prefix(long n, int ncell,
        const int *__restrict__ cell_sorted, int *__restrict__ prefix)
{
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int ci = cell_sorted[i];
        int cn = (i + 1 < n) ? cell_sorted[i + 1] : ncell;
        for (int c = ci; c < cn; c++) prefix[c] = (int)(i + 1);
        if (i == 0)
            for (int c = 0; c < ci; c++) prefix[c] = 0;
    }
}

__global__ __launch_bounds__(256) void k_fill_int(int n, int *p, int v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// write_sorting_buffer (cuda_sorting.py:192-213) for all attributes at once.
__global__ __launch_bounds__(256) void k_permute(long n, const int *__restrict__ sidx, int nattr,
                                                 CPtrs16 src, Ptrs16 dst)
{
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int j = sidx[i];
        for (int k = 0; k < nattr; k++) dst.p[k][i] = src.p[k][j];
    }
}

static inline int key_bits(int ncell)
{
    int b = 1;
    while (((long)1 << b) < (long)ncell) b++;
    return b;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t rocprim_temp_bytes(long n, int ncell)
{
    size_t bytes = 0;
    rocprim::double_buffer<int> k((int *)nullptr, (int *)nullptr);
    rocprim::double_buffer<int> v((int *)nullptr, (int *)nullptr);
    (void)rocprim::radix_sort_pairs((void *)nullptr, bytes, k, v, (size_t)n, 0u,
                                    (unsigned)key_bits(ncell), (hipStream_t)0, false);
    return bytes;
}

}  // namespace fb

using namespace fb;

extern "C" int fb_cell_index(long n, const double *x, const double *y, const double *z,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        int *cell_idx, int *sorted_idx, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_cell_index, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n,
                       x, y, z, invdz, zmin, Nz, invdr, rmin, Nr, cell_idx, sorted_idx);
    FB_CHECK_LAUNCH("fb_cell_index");
}

extern "C" size_t fb_sort_workspace_bytes(long n, int ncell)
{
    if (n <= 0) return 256;
    return align_up(rocprim_temp_bytes(n, ncell), 256) + 256;
}

extern "C" int fb_sort_by_cell(long n, int ncell, int *cell_idx, int *sorted_idx,
        int *cell_idx_alt, int *sorted_idx_alt, int *result_in_alt, int *prefix_sum,
        void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    *result_in_alt = 0;
    if (n <= 0) {
        if (ncell > 0)
            hipLaunchKernelGGL(k_fill_int, dim3((ncell + 255) / 256), dim3(256), 0, s, ncell,
                               prefix_sum, 0);
        FB_CHECK_LAUNCH("fb_sort_by_cell(empty)");
    }
    if (workspace_bytes < fb_sort_workspace_bytes(n, ncell)) {
        set_error("fb_sort_by_cell", "workspace too small");
        return -1;
    }
    // rocPRIM ping-pongs between the two buffer pairs; instead of copying the result back
    // (2 x 4n bytes per sort) the caller is told which pair holds it and swaps its handles.
    rocprim::double_buffer<int> k(cell_idx, cell_idx_alt);
    rocprim::double_buffer<int> v(sorted_idx, sorted_idx_alt);
    size_t temp_bytes = workspace_bytes;
    hipError_t e = rocprim::radix_sort_pairs(workspace, temp_bytes, k, v, (size_t)n, 0u,
                                             (unsigned)key_bits(ncell), s, false);
    if (e != hipSuccess) return check(e, "fb_sort_by_cell(radix)");
    const bool in_alt = (k.current() != cell_idx);
    if (in_alt != (v.current() != sorted_idx)) {
        set_error("fb_sort_by_cell", "key/value buffers out of step");
        return -1;
    }
    *result_in_alt = in_alt ? 1 : 0;
    hipLaunchKernelGGL(k_prefix, dim3(stream_grid(n)), dim3(256), 0, s, n, ncell, k.current(),
                       prefix_sum);
    FB_CHECK_LAUNCH("fb_sort_by_cell");
}

extern "C" int fb_permute(long n, const int *sorted_idx, int nattr, const double *const *src,
                          double *const *dst, void *stream)
{
    if (n <= 0 || nattr <= 0) return 0;
    if (nattr > 16) { set_error("fb_permute", "nattr > 16"); return -1; }
    CPtrs16 a;
    Ptrs16 b;
    for (int k = 0; k < 16; k++) { a.p[k] = k < nattr ? src[k] : nullptr; b.p[k] = k < nattr ? dst[k] : nullptr; }
    hipLaunchKernelGGL(k_permute, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n,
                       sorted_idx, nattr, a, b);
    FB_CHECK_LAUNCH("fb_permute");
}
