// Particle hand-over between neighbouring z-slabs: selection of the leaving particles, packing
// of the messages and compaction of the arrays as library kernels, counts on the device.
//
// Replaces the GPU cut of fbpic/boundaries/particle_buffer_handling.py:178-236 (prefix-sum cut +
// one copy kernel per attribute and side) and add_buffers_to_particles (:289-417), with the
// ownership rule of the reference's CPU path (:58-172: left if z < zbox_min, right if
// z > zbox_max), which is the parity target on every rank.
//
// Message layout (one per neighbour, fixed size so that it can be posted before any count is
// known on the host): FB_HANDOVER_HEADER doubles - [0] = number of particles the sender selected
// (may exceed `cap`: the receiver then knows that a remainder message follows) - then one row of
// `cap` doubles per attribute, the reference's one-row-per-attribute buffer.
#include "fb_common.h"

namespace fb {

// counts[] layout (device, 8 longs): 0 n_left, 1 n_right (selected, may exceed the capacities),
// 2 received-from-left, 3 received-from-right (fb_handover_recv_counts), 7 finished workgroups
__global__ __launch_bounds__(256) void k_handover_select_pack(long n, const double *__restrict__ z,
        const int *__restrict__ prefix_sum, long cut0, long cut1, long cut2, long cut3,
        double zbox_min, double zbox_max, int nattr, CPtrs16 arr, long cap_left, long cap_right,
        long idx_cap, double *__restrict__ send_left, double *__restrict__ send_right,
        int *__restrict__ idx_left, int *__restrict__ idx_right, unsigned long long *counts)
{
    // candidates: [0, o1) can leave to the left (all of [0, o0) do), [o2, n) to the right (all of
    // [o3, n) do).  Without a valid prefix sum every particle is a candidate for both sides.
    long o0 = 0, o1 = n, o2 = 0, o3 = n;
    if (prefix_sum) {
        auto off = [&](long c) -> long { return c < 0 ? 0 : (long)prefix_sum[c]; };
        o0 = off(cut0); o1 = off(cut1); o2 = off(cut2); o3 = off(cut3);
        if (o1 < o0) o1 = o0;
        if (o2 < o1) o2 = o1;
        if (o3 < o2) o3 = o2;
    }
    const bool both = (prefix_sum == nullptr);
    const long T = both ? n : o1 + (n - o2);
    const int lane = threadIdx.x & 63;
    const long step = (long)gridDim.x * blockDim.x;
    const long Tpad = (T + 63) / 64 * 64;       // whole waves take part in the ballots
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < Tpad; t += step) {
        bool left = false, right = false;
        long i = 0;
        if (t < T) {
            i = both ? t : (t < o1 ? t : o2 + (t - o1));
            const double zi = z[i];
            if (both) { left = zi < zbox_min; right = zi > zbox_max; }
            else if (t < o1) left = (i < o0) || (zi < zbox_min);
            else right = (i >= o3) || (zi > zbox_max);
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const bool mine = side ? right : left;
            const unsigned long long mask = __ballot(mine);
            if (!mask) continue;
            long base = 0;
            const int leader = __builtin_ctzll(mask);
            if (lane == leader) base = (long)atomicAdd(counts + side, (unsigned long long)__popcll(mask));
            base = __shfl(base, leader);
            if (mine) {
                const long slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                int *idx = side ? idx_right : idx_left;
                double *buf = side ? send_right : send_left;
                const long cap = side ? cap_right : cap_left;
                if (slot < idx_cap) idx[slot] = (int)i;
                if (buf && slot < cap)
                    for (int k = 0; k < nattr; k++)
                        buf[FB_HANDOVER_HEADER + k * cap + slot] = arr.p[k][i];
            }
        }
    }
    // the last workgroup to finish writes the message headers.  ALL waves of a workgroup must have
    // added their particles before its thread 0 reports the workgroup as finished: without the
    // barrier in front of that increment (rounds 3 and 4) the header could be written while the other
    // three waves of the last workgroup - or of any other one - were still counting, i.e. up to a
    // few waves' worth of particles too small.  The receiver then posted a remainder message 64
    // particles shorter than the sender's (gloo: "op.preamble.length <= op.nbytes", abort of that rank:
    // the rank loss of tests/test_gpu_c4.py, 1 run in ~20) - or, within the capacity, dropped them.
    __threadfence();
    __syncthreads();
    __shared__ bool last;
    if (threadIdx.x == 0) last = (atomicAdd(counts + 7, 1ull) == (unsigned long long)gridDim.x - 1);
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        const unsigned long long nl = atomicAdd(counts + 0, 0ull), nr = atomicAdd(counts + 1, 0ull);
        if (send_left) send_left[0] = (double)nl;
        if (send_right) send_right[0] = (double)nr;
    }
}

__global__ void k_handover_recv_counts(const double *recv_left, const double *recv_right,
                                       unsigned long long *counts)
{
    counts[2] = recv_left ? (unsigned long long)recv_left[0] : 0ull;
    counts[3] = recv_right ? (unsigned long long)recv_right[0] : 0ull;
}

// Compaction in O(number of movers): the n_leave = n_left + n_right leavers vacate their slots;
// the new length is m = n - n_leave.  Leavers at slots >= m simply fall off the end; each hole
// below m is filled with one of the survivors sitting at slots >= m (as many as holes).  Which
// survivor fills which hole is arbitrary (atomic counters): the particles are re-sorted by the
// next deposition anyway.
__global__ __launch_bounds__(256) void k_handover_mark(long n_left, const int *__restrict__ idx_left,
        long n_right, const int *__restrict__ idx_right, long m, int *__restrict__ tail_leaves,
        int *__restrict__ holes, int *__restrict__ counters)
{
    const long step = (long)gridDim.x * blockDim.x, nl = n_left + n_right;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < nl; j += step) {
        const long i = j < n_left ? idx_left[j] : idx_right[j - n_left];
        if (i >= m) tail_leaves[i - m] = 1;
        else holes[atomicAdd(counters, 1)] = (int)i;
    }
}

__global__ __launch_bounds__(256) void k_handover_fill(long n_leave, long m,
        const int *__restrict__ tail_leaves, const int *__restrict__ holes, int *__restrict__ counters,
        int nattr, Ptrs16 arr)
{
    const long step = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n_leave; j += step) {
        if (tail_leaves[j]) continue;
        const long dst = holes[atomicAdd(counters + 1, 1)];
        for (int k = 0; k < nattr; k++) arr.p[k][dst] = arr.p[k][m + j];
    }
}

__global__ __launch_bounds__(256) void k_handover_append_shift(long n, long m, int nattr, Ptrs16 arr,
        const double *__restrict__ buf, long stride, int shift_attr, double shift)
{
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
        for (int k = 0; k < nattr; k++) {
            double v = buf[k * stride + i];
            if (k == shift_attr) v += shift;
            arr.p[k][m + i] = v;
        }
}

}  // namespace fb

using namespace fb;

extern "C" int fb_handover_select_pack(long n, const double *z, const int *prefix_sum,
        long cut0, long cut1, long cut2, long cut3, double zbox_min, double zbox_max,
        int nattr, const double *const *arrays, long cap_left, long cap_right, long idx_cap,
        double *send_left, double *send_right, int *idx_left, int *idx_right, long *counts,
        void *stream)
{
    if (nattr <= 0 || nattr > 16) { set_error("fb_handover_select_pack", "nattr must be 1..16"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts, 0, 8 * sizeof(long), s);
    if (e != hipSuccess) return check(e, "fb_handover_select_pack");
    CPtrs16 a;
    for (int k = 0; k < 16; k++) a.p[k] = k < nattr ? arrays[k] : nullptr;
    // the candidates are a few cell rows when the prefix sum is given, all particles otherwise
    const int grid = prefix_sum ? 128 : stream_grid(n > 0 ? n : 1);
    hipLaunchKernelGGL(k_handover_select_pack, dim3(grid), dim3(256), 0, s, n, z, prefix_sum, cut0, cut1,
                       cut2, cut3, zbox_min, zbox_max, nattr, a, cap_left, cap_right, idx_cap, send_left,
                       send_right, idx_left, idx_right, (unsigned long long *)counts);
    FB_CHECK_LAUNCH("fb_handover_select_pack");
}

extern "C" int fb_handover_recv_counts(const double *recv_left, const double *recv_right, long *counts,
                                       void *stream)
{
    hipLaunchKernelGGL(k_handover_recv_counts, dim3(1), dim3(1), 0, (hipStream_t)stream, recv_left,
                       recv_right, (unsigned long long *)counts);
    FB_CHECK_LAUNCH("fb_handover_recv_counts");
}

extern "C" size_t fb_handover_workspace_bytes(long max_leavers)
{
    const size_t n = (size_t)(max_leavers > 0 ? max_leavers : 1);
    return 2 * align_up(n * sizeof(int), 256) + 256;
}

extern "C" int fb_handover_compact(long n, long n_left, const int *idx_left, long n_right,
        const int *idx_right, int nattr, double *const *arrays, void *workspace,
        size_t workspace_bytes, void *stream)
{
    const long n_leave = n_left + n_right;
    if (n_leave <= 0) return 0;
    if (nattr <= 0 || nattr > 16) { set_error("fb_handover_compact", "nattr must be 1..16"); return -1; }
    if (n_leave > n) { set_error("fb_handover_compact", "more leavers than particles"); return -1; }
    if (workspace_bytes < fb_handover_workspace_bytes(n_leave)) {
        set_error("fb_handover_compact", "workspace too small");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const size_t pb = align_up((size_t)n_leave * sizeof(int), 256);
    int *tail_leaves = (int *)ws, *holes = (int *)(ws + pb), *counters = (int *)(ws + 2 * pb);
    hipError_t e = hipMemsetAsync(tail_leaves, 0, pb, s);
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return check(e, "fb_handover_compact");
    Ptrs16 a;
    for (int k = 0; k < 16; k++) a.p[k] = k < nattr ? arrays[k] : nullptr;
    const long m = n - n_leave;
    hipLaunchKernelGGL(k_handover_mark, dim3(stream_grid(n_leave)), dim3(256), 0, s, n_left, idx_left,
                       n_right, idx_right, m, tail_leaves, holes, counters);
    hipLaunchKernelGGL(k_handover_fill, dim3(stream_grid(n_leave)), dim3(256), 0, s, n_leave, m,
                       tail_leaves, holes, counters, nattr, a);
    FB_CHECK_LAUNCH("fb_handover_compact");
}

extern "C" int fb_handover_append_shift(long n, long first, int nattr, double *const *arrays,
        const double *buf, long buf_row_stride, int shift_attr, double shift, void *stream)
{
    if (n <= 0 || nattr <= 0) return 0;
    if (nattr > 16) { set_error("fb_handover_append_shift", "nattr > 16"); return -1; }
    Ptrs16 a;
    for (int k = 0; k < 16; k++) a.p[k] = k < nattr ? arrays[k] : nullptr;
    hipLaunchKernelGGL(k_handover_append_shift, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream,
                       n, first, nattr, a, buf, buf_row_stride, shift_attr, shift);
    FB_CHECK_LAUNCH("fb_handover_append_shift");
}
