// Cell sort for gfx950: cell index per particle, stable radix sort by cell (rocPRIM),
// inclusive per-cell prefix sum, one-pass permutation of all particle attributes.
// Replaces fbpic/particles/utilities/cuda_sorting.py (Numba-CUDA + Thrust argsort).
#include <cstring>
#include "fb_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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


// ---- particle hand-over between z-slabs (boundaries/particle_buffer_handling.py) --------
// The three data movements of a hand-over, every attribute in one launch each:
//   pack   : buf[k][i] = arr[k][idx[i]]                    (the particles that leave)
//   move   : arr[k][dst[i]] = arr[k][src[i]]               (holes filled from the tail)
//   append : arr[k][m + i] = buf[k][i]                     (the arrivals)
// idx / src / dst are int64 (what the selection produces); src and dst are disjoint sets.
__global__ __launch_bounds__(256) void k_handover_pack(long n, const long *__restrict__ idx, int nattr,
                                                       CPtrs16 arr, double *__restrict__ buf, long stride)
{
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const long j = idx[i];
        for (int k = 0; k < nattr; k++) buf[k * stride + i] = arr.p[k][j];
    }
}

__global__ __launch_bounds__(256) void k_handover_move(long n, const long *__restrict__ src,
                                                       const long *__restrict__ dst, int nattr, Ptrs16 arr)
{
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const long a = src[i], b = dst[i];
        for (int k = 0; k < nattr; k++) arr.p[k][b] = arr.p[k][a];
    }
}

__global__ __launch_bounds__(256) void k_handover_append(long n, long m, int nattr, Ptrs16 arr,
                                                         const double *__restrict__ buf, long stride)
{
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
        for (int k = 0; k < nattr; k++) arr.p[k][m + i] = buf[k * stride + i];
}

// ---- counting sort by cell (fast path of Particles.sort_particles) -----------------
// The radix sort above is general; the PIC cycle re-sorts an ALMOST sorted stream every
// step, for which a counting sort needs one pass less over the keys and no permutation
// index:
//   k_bin_rank : cell index (same arithmetic as k_cell_index) + rank of each particle
//                inside its cell.  Lanes of a wave that sit in the same cell form runs
//                (the stream is nearly sorted): one atomicAdd per run on the per-cell
//                counter, the lanes take consecutive ranks -> ~n/ppc atomics, spread over
//                distinct addresses.
//   scan       : inclusive prefix sum of the per-cell counters (rocPRIM) = prefix_sum.
//   k_scatter  : every attribute is written to prefix_sum[c-1] + rank in ONE launch.
// The order of particles inside a cell is the arrival order of the runs (not the original
// order as with the stable radix sort); deposition and gather do not depend on it.
// Optional position push folded into the sort (fb_push_x_bin_sort_particles): the rank pass
// evaluates the pushed position in registers, the scatter pass evaluates it again (same
// expression as k_push_x, particles.hip) and writes it at the sorted slot, so the stand-alone
// push_x sweep (56 B read + 24 B written per particle) disappears.
template <bool PUSH>
__global__ __launch_bounds__(256) void k_bin_rank(long n, const double *__restrict__ x,
        const double *__restrict__ y, const double *__restrict__ z, PushX P,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        int *__restrict__ cell, int *__restrict__ rank, int *__restrict__ count)
{
    const int lane = threadIdx.x & 63;
    const long nchunks = (n + 63) / 64;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long chv = wave0; chv < nchunks; chv += nwaves) {
        const long i = chv * 64 + lane;
        const bool act = i < n;
        int c = -1;
        if (act) {
            double xj = x[i], yj = y[i], zj = z[i];
            if constexpr (PUSH) {
                const double g = P.ig[i];
                xj += P.chdt * g * P.px * P.ux[i];
                yj += P.chdt * g * P.py * P.uy[i];
                zj += P.chdt * g * P.pz * P.uz[i];
            }
            const double rj = sqrt(xj * xj + yj * yj);
            const double r_cell = invdr * (rj - rmin) - 0.5;
            const double z_cell = invdz * (zj - zmin) - 0.5;
            int ir_upper = (int)ceil(r_cell);
            int iz_upper = (int)ceil(z_cell);
            if (ir_upper > Nr) ir_upper = Nr;
            if (iz_upper < 0) iz_upper += Nz;
            else if (iz_upper > Nz - 1) iz_upper -= Nz;
            c = ir_upper + iz_upper * (Nr + 1);
        }
        const int prev = __shfl_up(c, 1);
        const bool is_start = act && (lane == 0 || c != prev);
        const unsigned long long starts = __ballot(is_start);
        const unsigned long long active = __ballot(act);
        const int cnt = __popcll(active);
        // my run: starts at the highest start bit at or below my lane
        const unsigned long long below = starts & ((2ull << lane) - 1ull);
        const int run0 = 63 - __builtin_clzll(below | 1ull);
        int base = 0;
        if (is_start) {
            const unsigned long long rest = (lane + 1 < 64) ? (starts >> (lane + 1)) : 0ull;
            int len = rest ? (__builtin_ctzll(rest) + 1) : (cnt - lane);
            base = atomicAdd(count + c, len);
        }
        base = __shfl(base, run0);
        if (act) {
            cell[i] = c;
            rank[i] = base + (lane - run0);
        }
    }
}

template <bool PUSH>
__global__ __launch_bounds__(256) void k_scatter(long n, const int *__restrict__ cell,
        const int *__restrict__ rank, const int *__restrict__ prefix, int nattr, CPtrs16 src,
        Ptrs16 dst, PushX P, int *__restrict__ cell_sorted, int *__restrict__ sorted_idx,
        int *__restrict__ count, int ncell)
{
    long stride = (long)gridDim.x * blockDim.x;
    // the per-cell counters have been consumed by the scan: leave them zeroed for the next
    // rank pass (saves its memset launch)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += stride) count[i] = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = cell[i];
        const int d = (c > 0 ? prefix[c - 1] : 0) + rank[i];
        if constexpr (PUSH) {
            // attributes 0..7 are x, y, z, ux, uy, uz, w, inv_gamma (checked by the host):
            // every array is read once, the pushed position is written at its sorted slot
            const double ux = src.p[3][i], uy = src.p[4][i], uz = src.p[5][i];
            const double wt = src.p[6][i], g = src.p[7][i];
            dst.p[0][d] = src.p[0][i] + P.chdt * g * P.px * ux;
            dst.p[1][d] = src.p[1][i] + P.chdt * g * P.py * uy;
            dst.p[2][d] = src.p[2][i] + P.chdt * g * P.pz * uz;
            dst.p[3][d] = ux; dst.p[4][d] = uy; dst.p[5][d] = uz;
            dst.p[6][d] = wt; dst.p[7][d] = g;
        }
        for (int k = PUSH ? 8 : 0; k < nattr; k++) dst.p[k][d] = src.p[k][i];
        if (cell_sorted) cell_sorted[d] = c;
        if (sorted_idx) sorted_idx[d] = (int)i;
    }
}

// Inverse of the scatter's destination map: sidx[prefix[c-1] + rank[i]] = i.  With it the
// particles can be walked in DESTINATION (cell-sorted) order by a kernel that reads its
// attributes through sidx and writes them contiguously - the fused sort + push_x + rho
// deposition (deposit.hip, fb_push_x_sort_deposit_rho).  Also leaves the per-cell counters
// zeroed for the next rank pass, like k_scatter.
__global__ __launch_bounds__(256) void k_build_sidx(long n, const int *__restrict__ cell,
        const int *__restrict__ rank, const int *__restrict__ prefix, int *__restrict__ sidx,
        int *__restrict__ count, int ncell)
{
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += stride) count[i] = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = FB_NT_LD(cell + i);
        FB_NT_ST((int)i, sidx + ((c > 0 ? prefix[c - 1] : 0) + FB_NT_LD(rank + i)));
    }
}

static size_t scan_temp_bytes(int ncell)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan((void *)nullptr, bytes, (int *)nullptr, (int *)nullptr,
                                  (size_t)ncell, rocprim::plus<int>(), (hipStream_t)0, false);
    return bytes;
}

static inline int key_bits(int ncell)
{
    int b = 1;
    while (((long)1 << b) < (long)ncell) b++;
    return b;
}


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

extern "C" int fb_handover_pack(long n, const long *idx, int nattr, const double *const *arrays,
                                double *buf, long buf_row_stride, void *stream)
{
    if (n <= 0 || nattr <= 0) return 0;
    if (nattr > 16) { set_error("fb_handover_pack", "nattr > 16"); return -1; }
    CPtrs16 a;
    for (int k = 0; k < 16; k++) a.p[k] = k < nattr ? arrays[k] : nullptr;
    hipLaunchKernelGGL(k_handover_pack, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n, idx,
                       nattr, a, buf, buf_row_stride);
    FB_CHECK_LAUNCH("fb_handover_pack");
}

extern "C" int fb_handover_move(long n, const long *src_idx, const long *dst_idx, int nattr,
                                double *const *arrays, void *stream)
{
    if (n <= 0 || nattr <= 0) return 0;
    if (nattr > 16) { set_error("fb_handover_move", "nattr > 16"); return -1; }
    Ptrs16 a;
    for (int k = 0; k < 16; k++) a.p[k] = k < nattr ? arrays[k] : nullptr;
    hipLaunchKernelGGL(k_handover_move, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n,
                       src_idx, dst_idx, nattr, a);
    FB_CHECK_LAUNCH("fb_handover_move");
}

extern "C" int fb_handover_append(long n, long first, int nattr, double *const *arrays,
                                  const double *buf, long buf_row_stride, void *stream)
{
    if (n <= 0 || nattr <= 0) return 0;
    if (nattr > 16) { set_error("fb_handover_append", "nattr > 16"); return -1; }
    Ptrs16 a;
    for (int k = 0; k < 16; k++) a.p[k] = k < nattr ? arrays[k] : nullptr;
    hipLaunchKernelGGL(k_handover_append, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n,
                       first, nattr, a, buf, buf_row_stride);
    FB_CHECK_LAUNCH("fb_handover_append");
}

extern "C" size_t fb_bin_sort_workspace_bytes(long n, int ncell)
{
    // per-cell counters + per-particle cell and rank + rocPRIM scan scratch
    return align_up((size_t)ncell * sizeof(int), 256) + 2 * align_up((size_t)(n > 0 ? n : 1) * sizeof(int), 256)
           + align_up(scan_temp_bytes(ncell), 256) + 256;
}

namespace fb {

// Rank pass (unless `preranked`) + scan: fills W.cell / W.rank and prefix_sum.  Shared by the
// counting sort below and by the fused sort + deposition of deposit.hip.
int bin_sort_prepare(const char *who, bool push, bool preranked, const PushX &P, long n, int ncell,
        const double *x, const double *y, const double *z, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, int nattr, const double *const *src, int *prefix_sum,
        void *workspace, size_t workspace_bytes, BinSortWs *Wout, hipStream_t s)
{
    if (nattr < 0 || nattr > 16) { set_error(who, "nattr > 16"); return -1; }
    if (ncell != Nz * (Nr + 1)) { set_error(who, "ncell != Nz*(Nr+1)"); return -1; }
    if (workspace_bytes < fb_bin_sort_workspace_bytes(n, ncell)) {
        set_error(who, "workspace too small");
        return -1;
    }
    if (push && n > 0 && (nattr < 8 || src[0] != x || src[1] != y || src[2] != z || src[3] != P.ux ||
                          src[4] != P.uy || src[5] != P.uz || src[7] != P.ig)) {
        set_error(who, "src[0..7] must be x, y, z, ux, uy, uz, w, inv_gamma");
        return -1;
    }
    const BinSortWs W = carve_bin_sort_ws(workspace, workspace_bytes, n, ncell);
    *Wout = W;
    hipError_t e = hipSuccess;
    if (!preranked) {
        // (when `preranked`, fb_deposit_J_rank_next has already filled count, cell and rank)
        e = hipMemsetAsync(W.count, 0, (size_t)ncell * sizeof(int), s);
        if (e != hipSuccess) return check(e, who);
    }
    if (n > 0 && !preranked) {
        const dim3 grid(stream_grid(n, 256, 256 * 16));
        if (push)
            hipLaunchKernelGGL(k_bin_rank<true>, grid, dim3(256), 0, s, n, x, y, z, P, invdz, zmin,
                               Nz, invdr, rmin, Nr, W.cell, W.rank, W.count);
        else
            hipLaunchKernelGGL(k_bin_rank<false>, grid, dim3(256), 0, s, n, x, y, z, P, invdz, zmin,
                               Nz, invdr, rmin, Nr, W.cell, W.rank, W.count);
        int r = check(hipGetLastError(), who);
        if (r) return r;
    }
    size_t temp_bytes = W.temp_bytes;
    e = rocprim::inclusive_scan(W.temp, temp_bytes, W.count, prefix_sum, (size_t)ncell,
                                rocprim::plus<int>(), s, false);
    if (e != hipSuccess) return check(e, who);
    return 0;
}

int bin_sort_build_sidx(const char *who, long n, int ncell, const BinSortWs &W, const int *prefix_sum,
                        int *sidx, hipStream_t s)
{
    hipLaunchKernelGGL(k_build_sidx, dim3(stream_grid(n > ncell ? n : ncell)), dim3(256), 0, s, n,
                       W.cell, W.rank, prefix_sum, sidx, W.count, ncell);
    return check(hipGetLastError(), who);
}

static int bin_sort_impl(const char *who, bool push, bool preranked, const PushX &P, long n, int ncell,
        const double *x, const double *y, const double *z, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, int nattr, const double *const *src,
        double *const *dst, int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
        void *workspace, size_t workspace_bytes, hipStream_t s)
{
    BinSortWs W;
    int r = bin_sort_prepare(who, push, preranked, P, n, ncell, x, y, z, invdz, zmin, Nz, invdr, rmin,
                             Nr, nattr, src, prefix_sum, workspace, workspace_bytes, &W, s);
    if (r) return r;
    int *count = W.count, *cell = W.cell, *rank = W.rank;
    if (n > 0) {
        CPtrs16 a;
        Ptrs16 b;
        for (int k = 0; k < 16; k++) { a.p[k] = k < nattr ? src[k] : nullptr; b.p[k] = k < nattr ? dst[k] : nullptr; }
        if (push)
            hipLaunchKernelGGL(k_scatter<true>, dim3(stream_grid(n)), dim3(256), 0, s, n, cell, rank,
                               prefix_sum, nattr, a, b, P, cell_idx_sorted, sorted_idx, count, ncell);
        else
            hipLaunchKernelGGL(k_scatter<false>, dim3(stream_grid(n)), dim3(256), 0, s, n, cell, rank,
                               prefix_sum, nattr, a, b, P, cell_idx_sorted, sorted_idx, count, ncell);
    }
    FB_CHECK_LAUNCH(who);
}

}  // namespace fb

using namespace fb;

extern "C" int fb_bin_sort_particles(long n, int ncell, const double *x, const double *y,
        const double *z, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        int nattr, const double *const *src, double *const *dst,
        int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
        void *workspace, size_t workspace_bytes, void *stream)
{
    PushX P = {nullptr, nullptr, nullptr, nullptr, 0., 0., 0., 0.};
    return bin_sort_impl("fb_bin_sort_particles", false, false, P, n, ncell, x, y, z, invdz, zmin, Nz,
                         invdr, rmin, Nr, nattr, src, dst, cell_idx_sorted, sorted_idx, prefix_sum,
                         workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fb_push_x_bin_sort_particles(long n, int ncell, const double *x, const double *y,
        const double *z, const double *ux, const double *uy, const double *uz,
        const double *inv_gamma, double c, double dt, double x_push, double y_push,
        double z_push, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        int nattr, const double *const *src, double *const *dst,
        int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
        void *workspace, size_t workspace_bytes, int preranked, void *stream)
{
    // fbpic/particles/push/numba_methods.py:24-30: chdt = c * dt
    PushX P = {ux, uy, uz, inv_gamma, c * dt, x_push, y_push, z_push};
    return bin_sort_impl("fb_push_x_bin_sort_particles", true, preranked != 0, P, n, ncell, x, y, z, invdz, zmin,
                         Nz, invdr, rmin, Nr, nattr, src, dst, cell_idx_sorted, sorted_idx,
                         prefix_sum, workspace, workspace_bytes, (hipStream_t)stream);
}
