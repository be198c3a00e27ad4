// Shared helpers for the gfx950 kernels of libfbpic_amd.so.
// Whole library is compiled with -ffp-contract=off: cell indices must be bit-identical
// to the reference CPU path (invdr*(r-rmin)-0.5 must not fuse into an FMA).
#pragma once
#include <cstdlib>
#include <cstring>
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

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Waves per workgroup of the kernels whose occupancy is set by a per-wave LDS panel: among 4, 3, 2,
// 1 waves (panel of a workgroup <= 64 KiB) the choice that lets most waves share a CU.  The LDS
// of a workgroup is allocated in blocks of 2 KiB (tools/lds_probe.hip: 3 x 53 248 B and 4 x 40 960 B
// co-reside on a CU, 3 x 53 880 B do not): three workgroups of 53.9 KB do NOT share the 160 KB of a
// CU although 3 x 53.9 < 160 (the fused cubic pass of 2048 x 512 x 64 ppc, Nm = 4: 7.09 ms with
// 3-wave workgroups, 5.7-5.9 ms with four 2-wave ones).
inline int lds_waves_per_workgroup(size_t wave_bytes, int max_waves = 4, int useful_per_cu = 16)
{
    // waves per CU of every choice (capped at what the registers allow anyway: 4 per SIMD for
    // these kernels), then the LARGEST workgroup within 10 % of the best
    size_t per_cu[5] = {0, 0, 0, 0, 0}, best_waves = 0;
    for (int nw = max_waves; nw >= 1; nw--) {
        const size_t wg = (wave_bytes * nw + 2047) / 2048 * 2048;
        if (wg > 64 * 1024 && nw > 1) continue;
        size_t w = (160 * 1024) / wg * nw;
        if (w > (size_t)useful_per_cu) w = useful_per_cu;
        per_cu[nw] = w;
        if (w > best_waves) best_waves = w;
    }
    for (int nw = max_waves; nw >= 1; nw--)
        if (per_cu[nw] * 10 >= best_waves * 9) return nw;
    return 1;
}


// Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b % 8), each with
// its own L2.  For kernels that walk the cell-sorted particle stream, give every XCD one
// contiguous eighth of the stream: neighbouring cells (which share grid nodes) then meet in
// ONE L2 - field nodes are fetched from HBM once instead of once per XCD, and the lines that
// deposition atomics update stop bouncing between L2s.  Launch with xcd_grid(nb) workgroups;
// logical ids >= nb are the idle tail.
constexpr int FB_NXCD = 8;
inline long xcd_grid(long nblocks) { return (nblocks + FB_NXCD - 1) / FB_NXCD * FB_NXCD; }
#ifdef __HIPCC__
__device__ __forceinline__ long xcd_block_id()
{
    const long per = gridDim.x / FB_NXCD;
    return (long)(blockIdx.x % FB_NXCD) * per + blockIdx.x / FB_NXCD;
}
#endif

// GRADED ranges of 64-particle chunks for the particle kernels that walk ranges (one-wave workgroups).  The waves of a
// launch end one wave duration apart, and once the dispatcher has handed out its last workgroup the chip drains for that
// long (C3: a wave of 10 chunks lasts a fifth of the launch).  Workgroups start in the order of their index within an
// XCD, so the waves with the HIGHEST indices - the last 1 / den of an XCD's chunks - walk `short_cpw` chunks each and
// the others `cpw` (short ranges everywhere would pay the per-wave prologue of these kernels for every chunk or two).
// A static cut: no atomics, every chunk belongs to exactly one wave.  long_waves < 0: the plain cut (every wave `cpw`
// chunks, workgroups of any number of waves).  profiles/r06_graded_ranges.txt.
struct WaveRanges {
    int cpw, long_waves, short_cpw;
    long chunks_per_xcd, long_chunks;      // chunks of an XCD; those of its long waves
};
// (FBPIC_AMD_CYCLE_TAIL = "<den>,<short>": developer override for scans; "0" = plain cut)
inline WaveRanges graded_ranges(long nchunks, int cpw, int nwaves, long *nblocks, int den = 8, int shrt = 2)
{
    static const char *env = getenv("FBPIC_AMD_CYCLE_TAIL");
    if (env && *env) { den = atoi(env); const char *c = strchr(env, ','); shrt = c ? atoi(c + 1) : 1; }
    WaveRanges R = {cpw, -1, cpw, 0, 0};
    const long total_waves = (nchunks + cpw - 1) / cpw;
    *nblocks = xcd_grid((total_waves + nwaves - 1) / nwaves);
    if (nwaves == 1 && den > 0 && shrt > 0 && shrt < cpw) {
        const long cx = (nchunks + FB_NXCD - 1) / FB_NXCD;
        long tail = cx / den;
        tail = (tail + shrt - 1) / shrt * shrt;                           // whole short ranges
        if (tail > 0 && tail < cx) {
            const long lc = cx - tail, lw = (lc + cpw - 1) / cpw, sw = tail / shrt;
            R.long_waves = (int)lw; R.short_cpw = shrt; R.chunks_per_xcd = cx; R.long_chunks = lc;
            *nblocks = FB_NXCD * (lw + sw);
        }
    }
    return R;
}
#ifdef __HIPCC__
// first chunk and number of chunks of this wave; false: nothing to do (a wave beyond the end of its XCD's chunks)
__device__ __forceinline__ bool wave_range(const WaveRanges &R, int nwaves, int wave, long &chunk0, int &nch)
{
    chunk0 = (xcd_block_id() * nwaves + wave) * R.cpw;
    nch = R.cpw;
    if (R.long_waves >= 0) {
        const long xcd = blockIdx.x % FB_NXCD, j = blockIdx.x / FB_NXCD;
        const bool lng = j < R.long_waves;
        const long o = lng ? j * R.cpw : R.long_chunks + (j - R.long_waves) * R.short_cpw;
        const long room = (lng ? R.long_chunks : R.chunks_per_xcd) - o;
        chunk0 = xcd * R.chunks_per_xcd + o;
        nch = (int)max(0L, min((long)(lng ? R.cpw : R.short_cpw), room));
        nch = __builtin_amdgcn_readfirstlane(nch);
    }
    return nch > 0;
}
#endif


// Layout of the counting-sort workspace (fb_bin_sort_workspace_bytes): per-cell counters,
// per-particle cell and rank, then the rocPRIM scan scratch.  Shared by sort.hip and by the
// deposition kernel that pre-computes cell + rank for the sort that follows it.
struct BinSortWs {
    int *count, *cell, *rank;
    void *temp;
    size_t temp_bytes;
};
inline BinSortWs carve_bin_sort_ws(void *workspace, size_t workspace_bytes, long n, int ncell)
{
    BinSortWs w;
    char *ws = (char *)workspace;
    w.count = (int *)ws;
    ws += align_up((size_t)ncell * sizeof(int), 256);
    const size_t pb = align_up((size_t)(n > 0 ? n : 1) * sizeof(int), 256);
    w.cell = (int *)ws; ws += pb;
    w.rank = (int *)ws; ws += pb;
    w.temp = ws;
    w.temp_bytes = workspace_bytes - (size_t)(ws - (char *)workspace);
    return w;
}

// Position push evaluated inside another kernel (same expression as k_push_x, particles.hip)
struct PushX {
    const double *ux, *uy, *uz, *ig;
    double chdt, px, py, pz;
};

// Particle streams (positions, momenta, weights, home cells, permutation indices: read once and written once per
// pass, 500 MB per step at the headline size) are NON-TEMPORAL: global_load / global_store ... nt.  What the grid
// kernels of the step re-read - interpolation grids, deposition records, the 46 MB spectral slab, the solver's tables
// (~105 MB together) - then survives in L2 / the 256 MB Infinity Cache between the particle passes.  Round 6, C2: fused
// spectral launch 49.6 -> 44.5 us, step 0.384 -> 0.373 ms (profiles/r06_nontemporal_streams.txt).  -DFB_NO_NT: plain
// accesses, for A/B builds.
// (FB_NT_LDG: the permuted gathers of the sorting pass - a line is shared by the lanes of neighbouring particles)
#if !defined(FB_NO_NT) && !defined(FB_NO_NT_GATHER)
#define FB_NT_LDG(p) __builtin_nontemporal_load(p)
#else
#define FB_NT_LDG(p) (*(p))
#endif
#ifndef FB_NO_NT
#define FB_NT_LD(p) __builtin_nontemporal_load(p)
#define FB_NT_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define FB_NT_LD(p) (*(p))
#define FB_NT_ST(v, p) (*(p) = (v))
#endif

// s_waitcnt vmcnt(0) as an INSTRUCTION the compiler sees (an asm statement would leave its wait-count
// bookkeeping believing the loads - also those into LDS - are still pending: it then waits again,
// with vmcnt(0), in front of the first use, i.e. for whatever stores and atomics were issued since)
#ifdef __HIPCC__
__device__ __forceinline__ void fb_wait_vm()
{
    // (compiler barriers on both sides: without the first one a load requested just in front of the wait may
    // be scheduled behind it - seen in the prologue of k_perm_deposit_J_rho, whose loop then waited at its top)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), expcnt and lgkmcnt untouched (gfx9 encoding)
    asm volatile("" ::: "memory");
}
// "this value is needed HERE": the compiler waits for the load that produces it in front of this point (loads
// through const __restrict__ pointers are invariant to it and move freely across fb_wait_vm's barriers - a
// prologue that ends with fb_wait_vm may still leave its loads in flight, and the loop then waits at its top)
template <class T> __device__ __forceinline__ void fb_consume(T &v) { asm volatile("" : "+v"(v)); }
#endif

// Optional by-product of a kernel that holds x, y, z, u, inv_gamma of every particle in
// registers (fb_deposit_J_rank_next, fb_gather_push_rank_next): Simulation.step pushes the
// positions by another half step and then re-sorts them (main.py:519-528).  The kernel also
// evaluates that pushed position (same expression as k_push_x), its cell and the rank of the
// particle inside that cell (one atomic per run of equal cells, as k_bin_rank in sort.hip): the
// counting sort then needs neither its own pass over the particles nor the 56 B / particle
// that pass reads.
struct RankNext {
    double chdt, px, py, pz;
    int *cell, *rank, *count;
};

// sort.hip: rank pass (unless preranked) + scan of the counting sort, and the inverse
// permutation sidx[destination] = source; used by the fused sort + push_x + rho deposition
int bin_sort_prepare(const char *who, bool push, bool preranked, const PushX &P, long n, int ncell,
        const double *x, const double *y, const double *z, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, int nattr, const double *const *src, int *prefix_sum,
        void *workspace, size_t workspace_bytes, BinSortWs *Wout, hipStream_t s);
int bin_sort_build_sidx(const char *who, long n, int ncell, const BinSortWs &W, const int *prefix_sum,
                        int *sidx, hipStream_t s);

// Longest range of 64-particle chunks one wave of a particle kernel walks (launches aim at 16384 waves and lengthen the
// ranges beyond that).  Round 6, C5 (1 M chunks): ranges of 64 chunks make a wave last ~1 ms of a 5 ms launch, and the last
// generation of waves leaves the chip half empty - gather + push + rank 5.07 -> 4.77 ms with ranges of 8, the sorting
// deposition pass 5.64 -> 5.46 with 16 (32: 5.44, 8: 5.62), profiles/r06_range_per_wave.txt.  FBPIC_AMD_CPW_CAP overrides
// every kernel's cap (developer knob for such scans).
inline int fb_cpw_cap(int dflt)
{
    static const int e = getenv("FBPIC_AMD_CPW_CAP") ? atoi(getenv("FBPIC_AMD_CPW_CAP")) : 0;
    return e > 0 ? e : dflt;
}

}  // namespace fb

#define FB_CHECK_LAUNCH(where) return fb::check(hipGetLastError(), where)
