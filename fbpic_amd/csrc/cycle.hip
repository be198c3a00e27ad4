// The particle work of one PIC step as ONE pass over the particles (gfx950):
//   gather E,B at x(n) -> Vay push_p -> push_x(dt/2) -> deposit J from x(n+1/2) -> push_x(dt/2)
//   -> deposit rho from x(n+1)
// i.e. main.py:469-528 of the reference (gather, push_p, push_x, deposit('J'), push_x,
// deposit('rho_next')) with every particle attribute read once and written once: 64 B read +
// 56 B written per particle, where the two-pass sequence (fb_gather_push_rank_next +
// fb_push_x_sort_deposit_J_rho) moves 252 B and re-sorts the arrays every step.
//
// What makes the single pass possible is NOT sorting every step.  The run-based deposition and
// the segment-based gather need particles of one cell to be contiguous; a plasma changes that
// order slowly (a 0.01 c thermal plasma: ~1.6 % of the particles change cell per step), so the
// arrays are counting-sorted every few steps only (fb_bin_sort_particles, which also records the
// cell of every particle at that moment: `home`), and in between
//   * the runs / segments of a wave are those of the HOME cells - contiguous by construction, 2-3
//     per 64 particles at 32 ppc, found with one ballot on `home`;
//   * a particle that still has the stencil of its home cell takes part in the run (matrix-core
//     reduction, one flush per cell) or reads the segment's staged node values;
//   * a particle that has left its home cell (a "stray") does not break the run of its
//     neighbours: the deposition writes its (Sz Sr) x amplitude products directly (lane = node x
//     amplitude, one atomic instruction per engine, DepEngine::scatter_one), the gather stages
//     its own stencil as one more segment of the same L2 round trip.
// Correctness never depends on the order (every particle is deposited / gathered exactly once
// either way); only the share of strays - reported in `stats` for the host's sort policy - does.
//
// Per-particle arithmetic is that of the separate entry points (k_gather / vay / k_push_x /
// DepEngine::stage): momenta and positions are bit-identical to the four-call sequence, the
// deposited sums differ by summation order only.
#include <cstdlib>
#include <cstring>
#include "fb_common.h"
#include "push_common.h"
#include "dep_engine.h"
#include "cycle_dep.h"

namespace fb {

struct CycleArgs {
    long n;
    // (order: the kernel fetches these pointers from the kernel-argument segment in groups - x, y, z,
    // home | ux, uy, uz, ig - with one scalar load per group, see karg_ptrs)
    double *x, *y, *z;
    const int *home;                           // cell ir_upper + iz_upper (Nr+1) at the last sort
    double *ux, *uy, *uz, *ig;
    const double *w;
    int home_shift;                            // ... minus this: (cells the grid has moved since) x (Nr+1)
    int regroup_at;                            // chunks with more J-strays than this are regrouped in the wave
    int regroup_pairs;                         // ... by (J cell, rho cell) pairs (else by the J cell, rho strays scattered)
    double *Ex, *Ey, *Ez, *Bx, *By, *Bz;       // optional: gathered fields stored
    double invdz, zmin;
    int Nz;
    double invdr, rmin;
    int Nr;
    double inv_ncol;                           // 1 / (Nr + 1)
    double rmax_gather;
    GatherGrids G;
    long rsG;
    const cplx *baseG;                         // lowest of the grids (all within 4 GiB of it)
    double econst, bconst, chdt;               // q dt/(m c), q dt/(2 m), c dt_x
    double wzmin, wzmax;                       // periodic wrap of z before the gather (off: max <= min)
    double q, c_light;
    DepGrids GJ;
    long rsJ;
    DepGrids GR;
    long rsR;
    cplx *baseJ, *baseR;                       // dep_grids_base() of the two targets
    const double *beta0, *betah;               // Ruyten coefficients, mode 0 / modes >= 1
    WaveRanges rg;                             // range of chunks of every wave (graded_ranges, fb_common.h)
    unsigned long long *stats;                 // optional: [0, 512) strays of the J deposition, [512, 1024) chunks
                                               // with more than FB_CYCLE_BAD_CHUNK of them
    // RANK mode (fb_gather_push_rank_next_home): no deposition; x is left at x(n+1/2) and the cell
    // of x(n+1) and the particle's rank in it go to the counting-sort workspace
    int *rk_cell, *rk_rank, *rk_count;
};

// The node values of chunk c+1 are requested right after the stencil sums of chunk c and travel
// during its push and depositions: the gather panel has LDS of its own next to the panel of the
// deposition engines (12.5 KB per wave at Nm = 2, 3 waves per SIMD).  (Requesting them at the end
// of chunk c into a shared panel - 7.8 KB, 4 waves per SIMD - and forcing 128 registers were
// measured too: 266-273 us against 266-274 at C2, profiles/README.md.)
// WIDE: grids / deposition targets that are not within 4 GiB of each other (separately allocated
// arrays): 64-bit pointers per lane instead of a scalar base + 32-bit lane offsets.
template <int SHAPE, int NM, bool WIDE> struct CyclePlan {
    static constexpr int S = ShapeTraits<SHAPE>::S, H = ShapeTraits<SHAPE>::H;
    static constexpr int NV = S * S * 6 * NM;              // complex node values of a segment
    static constexpr int NVL = (NV + 63) / 64;             // ... per lane
    // segments staged per round (cubic: 3 KB per segment at Nm = 2, 6 KB at Nm = 4)
    static constexpr int NSEG = (NVL == 1) ? 6 : (S == 2 ? 3 : 2);
    // panel stride in doubles: load j of a segment fills the 16-B slots 64 j ... of its lanes
    // (only NV of them in all), + a 16-B pad (segments start on different banks)
    static constexpr int PSTR = 2 * NV + 2;
    // WIDE: the two DepEngines one after the other on one panel (64-bit pointers per lane); else the
    // merged engine of cycle_dep.h (J and rho staged together, one traversal of the runs).
    // (-DFB_CYCLE_TWO_ENGINES: round 4's form of the 32-bit path as well, for A/B builds)
    // Cubic shape (round 6): the two DepEngines (node-row MFMA blocks, dep_engine.h), 32-bit offsets.
#ifdef FB_CYCLE_TWO_ENGINES
    static constexpr bool MERGED = false;
#else
    static constexpr bool MERGED = !WIDE && SHAPE == FB_SHAPE_LINEAR;
#endif
    using EJ = DepEngine<SHAPE, 3, NM, true, !WIDE && !MERGED>;
    using ER = DepEngine<SHAPE, 1, NM, true, !WIDE && !MERGED>;
    using ED = CycleDep<NM>;
    static constexpr int DEP2_DOUBLES = EJ::L::WAVE_DOUBLES > ER::L::WAVE_DOUBLES ? EJ::L::WAVE_DOUBLES
                                                                                  : ER::L::WAVE_DOUBLES;
    static constexpr int DEP_DOUBLES = MERGED ? ED::L::WAVE_DOUBLES : DEP2_DOUBLES;
    static constexpr int GATHER_DOUBLES = NSEG * PSTR;
    // Linear shape: the two panels do not share LDS - the node values of chunk c+1 arrive while chunk c
    // deposits.  Cubic shape, Nm >= 3: they DO (the gather panel lies on top of the deposition panel:
    // 18 KB per wave at Nm = 4 instead of 30, i.e. 8 waves per CU instead of 5), and the node values of
    // chunk c+1 are requested at the END of chunk c.
    static constexpr bool OVERLAY = (SHAPE == FB_SHAPE_CUBIC) && NM >= 3;
    static constexpr int WAVE_DOUBLES = OVERLAY ? (GATHER_DOUBLES > DEP_DOUBLES ? GATHER_DOUBLES : DEP_DOUBLES)
                                                : GATHER_DOUBLES + DEP_DOUBLES;
};

// Array pointers are fetched from the kernel-argument segment where they are used (one scalar
// load each) instead of living in scalar registers across the whole chunk loop: with ~20 array
// pointers on top of the geometry, the two engines' state and the lane masks the kernel needs
// more than the 102 SGPRs of a wave, and every spilled one costs v_readlane / v_writelane
// instructions on the VALU - the unit this kernel is bound by.
template <class T>
__device__ __forceinline__ T *karg_ptr(int byte_offset)
{
    T *p;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(p) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(byte_offset));
    return p;
}
// ... and several neighbouring pointers with ONE scalar load and one wait (every karg_ptr is a
// round trip to the scalar cache that the wave sits out: 20 per chunk before, 5 now)
template <int N> struct KPtrs { char *p[N]; };
__device__ __forceinline__ KPtrs<2> karg_ptrs2(int byte_offset)
{
    typedef unsigned long v2 __attribute__((ext_vector_type(2)));
    v2 v;
    asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(byte_offset));
    return {{(char *)v[0], (char *)v[1]}};
}
__device__ __forceinline__ KPtrs<4> karg_ptrs4(int byte_offset)
{
    typedef unsigned long v4 __attribute__((ext_vector_type(4)));
    v4 v;
    asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(byte_offset));
    return {{(char *)v[0], (char *)v[1], (char *)v[2], (char *)v[3]}};
}
__device__ __forceinline__ KPtrs<8> karg_ptrs8(int byte_offset)
{
    typedef unsigned long v8 __attribute__((ext_vector_type(8)));
    v8 v;
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(byte_offset));
    return {{(char *)v[0], (char *)v[1], (char *)v[2], (char *)v[3], (char *)v[4], (char *)v[5], (char *)v[6],
             (char *)v[7]}};
}
#define KOFF(field) ((int)__builtin_offsetof(CycleArgs, field))
static_assert(__builtin_offsetof(CycleArgs, y) == __builtin_offsetof(CycleArgs, x) + 8 &&
              __builtin_offsetof(CycleArgs, z) == __builtin_offsetof(CycleArgs, x) + 16 &&
              __builtin_offsetof(CycleArgs, home) == __builtin_offsetof(CycleArgs, x) + 24 &&
              __builtin_offsetof(CycleArgs, ux) == __builtin_offsetof(CycleArgs, x) + 32 &&
              __builtin_offsetof(CycleArgs, ig) == __builtin_offsetof(CycleArgs, x) + 56 &&
              __builtin_offsetof(CycleArgs, Bz) == __builtin_offsetof(CycleArgs, Ex) + 40 &&
              __builtin_offsetof(CycleArgs, betah) == __builtin_offsetof(CycleArgs, beta0) + 8,
              "pointer groups of karg_ptrs");
#ifdef FB_ISA_MARKS
#define FB_MARK(x) asm volatile("; MARK " x)
#elif defined(FB_CYCLE_TRACE)
// timing experiment (tools/cycle_trace.py): shader-clock time a wave spends between the marks of the chunk
// loop, summed per mark over all waves of the launch
__device__ unsigned long long cycle_trace_buf[16];
#define FB_MARK(x) do { const unsigned long long t_ = __builtin_readcyclecounter(); \
                        tr_acc[tr_k] += t_ - tr_t; tr_t = t_; tr_k = (tr_k + 1) % 9; } while (0)
#else
#define FB_MARK(x)
#endif
#define KP(T, field) karg_ptr<T>((int)__builtin_offsetof(CycleArgs, field))
// a chunk of 64 particles of which more than this many have left their home cell costs several times
// the normal chunk (every stray is a gather segment and a scatter of its own): counted for the host's
// sort policy (a laser wake turns whole regions into such chunks, and their waves are the kernel's tail)
#define FB_CYCLE_BAD_CHUNK 16
// strays of the J deposition per 64 particles from which a chunk is regrouped inside the wave (see the
// deposition part of the chunk loop)
#ifndef FB_CYCLE_REGROUP_AT
#define FB_CYCLE_REGROUP_AT 12
#endif

// Front half of a chunk: what can be done as soon as the positions are there - the keys of the
// home runs, the stencil origin of every particle, which particles are strays, the segment of
// every lane - and the node values of the first round of segments REQUESTED (not waited for):
// `global_load_lds` puts them straight into the wave's gather panel, no registers in between.
// cycle_linear_body runs it for chunk c+1 right after the stencil sums of chunk c have left
// the panel: the L2 round trip overlaps the push and both depositions of chunk c.
struct CycleFront {
    double x, y, z, rj;               // position (z wrapped into the periodic box), radius
    int hkz, hkr, hnb;                // stencil key of the lane's home run
    int kz, kr, myseg;                // own stencil origin; segment holding this lane's node values
    unsigned long long runstarts, insidem;
    int nseg, cnt;
    unsigned long long rem_r, rem_s;  // segments not requested yet (runs / strays)
};

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

template <int SHAPE, int NM, bool WIDE, bool RANK>
__device__ __forceinline__ void cycle_linear_body(const CycleArgs &A)
{
    using P = CyclePlan<SHAPE, NM, WIDE>;
    using EJ = typename P::EJ;
    using ER = typename P::ER;
    using ED = typename P::ED;
    constexpr int S = P::S, H = P::H, NV = P::NV, NVL = P::NVL, NSEG = P::NSEG, PSTR = P::PSTR;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per wave: the gather panel (node values of the segments, written by the loads themselves)
    // and, behind it, the panel of the deposition engines
    // (RANK mode has no deposition panel)
    double *gpanel = lds + (size_t)wave * (RANK ? P::GATHER_DOUBLES : P::WAVE_DOUBLES);
    double *dpanel = P::OVERLAY ? gpanel : gpanel + P::GATHER_DOUBLES;
    const long n = A.n;
    const int Nz = A.Nz, Nr = A.Nr, ncol = Nr + 1;
    const long rs = A.rsG;
    EJ ej;
    ER er;
    ED ed;
    if constexpr (!RANK) {
        if constexpr (!P::MERGED) {
            ej.init(dpanel, lane, A.GJ, A.rsJ, 0, Nz, Nr, A.baseJ);
            er.init(dpanel, lane, A.GR, A.rsR, 0, Nz, Nr, A.baseR);
        } else {
            ed.init(dpanel, lane, A.GJ, A.rsJ, A.GR, A.rsR, Nz, Nr, A.baseJ);
        }
    }
    // RANK: (cell, rank) of the previous chunk, written once the rank atomic has returned
    long pd_i = -1;
    int pd_cell = 0, pd_base = 0, pd_run0 = 0;
    const DepGeom geom = {A.invdz, A.zmin, Nz, A.invdr, A.rmin, Nr};
    const bool store_eb = A.Ex != nullptr;
    const unsigned long long le = (2ull << lane) - 1ull, lt = (1ull << lane) - 1ull;

    // staging role of this lane in the gather: node value o = lane + 64 j of every segment
    // (node (jz, jr) = the low bits of the lane - 64 is a multiple of S S -, field o / (S S))
    const int st_jr = lane & (S - 1), st_jz = (lane / S) & (S - 1);
    // its field as a 32-bit byte offset from the lowest of the grids (scalar base + lane offset
    // addressing; the host has checked that all of them lie within 4 GiB)
    unsigned st_rel[NVL];
    const char *st_ptr[NVL];          // WIDE: the field itself
    unsigned st_on = 0u;
    const char *gbase = (const char *)A.baseG;
    const int rsB = (int)(16 * rs);
#pragma unroll
    for (int j = 0; j < NVL; j++) {
        const int o = lane + 64 * j;
        const bool on = o < NV;
        st_ptr[j] = (const char *)A.G.g[on ? o / (S * S) : 0];
        st_rel[j] = WIDE ? 0u : (unsigned)(st_ptr[j] - gbase);
        st_on |= on ? (1u << j) : 0u;
    }
    // request the node values of the next (up to NSEG) segments of `f`: runs first, then strays
    auto request = [&](CycleFront &f, int ns) {
#pragma unroll
        for (int u = 0; u < NSEG; u++) {
            if (u < ns) {
                int skz, skr;
                if (f.rem_r) {
                    const int l = __builtin_ctzll(f.rem_r);
                    f.rem_r &= f.rem_r - 1ull;
                    skz = __builtin_amdgcn_readlane(f.hkz, l);
                    skr = __builtin_amdgcn_readlane(f.hkr, l);
                } else {
                    const int l = __builtin_ctzll(f.rem_s);
                    f.rem_s &= f.rem_s - 1ull;
                    skz = __builtin_amdgcn_readlane(f.kz, l);
                    skr = __builtin_amdgcn_readlane(f.kr, l);
                }
                // z wrap, mirror below the axis (its sign is applied to the weights, see the
                // stencil sums), clamp at the outer edge (gathering/inline_functions.py:20-90)
                int row = skz + st_jz, col = skr + st_jr;
                if (row < 0) row += Nz; else if (row > Nz - 1) row -= Nz;
                if (col < 0) col = -col - 1; else if (col > Nr - 1) col = Nr - 1;
                const unsigned off = (unsigned)(row * rsB + col * 16);
#pragma unroll
                for (int j = 0; j < NVL; j++) {
                    // lane l of the wave -> 16 bytes at (LDS base) + 16 l
                    if ((st_on >> j) & 1u) {
                        const char *src = WIDE ? st_ptr[j] + ((long)row * (16 * rs) + (long)col * 16)
                                               : gbase + (st_rel[j] + off);
                        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(gpanel + u * PSTR + 128 * j),
                                                         16, 0, 0);
                    }
                }
            }
        }
    };
    auto front = [&](CycleFront &f, long base, double xj, double yj, double zj, int hn, bool ask) {
        // (Lanes beyond the last particle hold a copy of it - load_pos - and compute along: only
        // the masks, the stores and the deposited weight know that they are not particles.  The
        // chunk loop then has no divergent region around its arithmetic.)
        const bool act = base + lane < n;
        f.cnt = (int)min((long)64, n - base);
        if (A.wzmax > A.wzmin) {
            // boundaries/particle_buffer_handling.py:536-556 (k_shift_periodic): one step of each
            // of its two loops, the loops themselves only if a particle is further out than a box
            const double l_box = A.wzmax - A.wzmin;
            zj = (zj >= A.wzmax) ? zj - l_box : zj;
            zj = (zj < A.wzmin) ? zj + l_box : zj;
            if (__ballot(zj >= A.wzmax || zj < A.wzmin)) {
                while (zj >= A.wzmax) zj -= l_box;
                while (zj < A.wzmin) zj += l_box;
            }
        }
        f.x = xj; f.y = yj; f.z = zj;
        // home cell -> stencil key of the lane's run (linear shape: lowest node of the 2 x 2
        // stencil = (iz_upper - 1, ir_upper - 1), the same for gather and deposition).
        // hc / ncol: (hc + 0.5) / ncol is at least 0.5 / ncol away from an integer, so the floor
        // of the rounded product is the quotient for any int hc.
        const int hc = hn - A.home_shift;
        const int hzu = (int)floor(((double)hc + 0.5) * A.inv_ncol);
        const int hru = hc - hzu * ncol;
        f.hkz = hzu - H; f.hkr = hru - H; f.hnb = H - hru;
        const int hprev = __shfl_up(hc, 1);
        f.runstarts = __ballot(act && (lane == 0 || hc != hprev));
        // own stencil origin (threading_methods.py:108-117)
        const double rj = sqrt(xj * xj + yj * yj);
        f.rj = rj;
        const double r_cell = A.invdr * (rj - A.rmin) - 0.5;
        const double z_cell = A.invdz * (zj - A.zmin) - 0.5;
        const bool inside = act && rj < A.rmax_gather;
        // (cubic: threading_methods.py:312-321 - the stencil starts one node below the lower node)
        const int kr = (int)floor(r_cell) - (H - 1), kz = (int)floor(z_cell) - (H - 1);
        f.kz = kz; f.kr = kr;
        const bool g_home = inside && kz == f.hkz && kr == f.hkr;
        const unsigned long long g_homem = __ballot(g_home);
        f.insidem = __ballot(inside);
        const unsigned long long straym = __ballot(inside && !g_home);
        // Segments of the runs: only runs that still hold a particle of their own cell are staged
        // (an emptied run would be staged for nothing - and from whatever `home` holds: the
        // result must not depend on it).  One bit per such run, at its first home lane.
        const int mystart = 63 - __builtin_clzll((f.runstarts & le) | 1ull);
        const bool g_first = g_home && ((g_homem & lt) >> mystart) == 0ull;
        const unsigned long long g_runs = __ballot(g_first);
        const int ngruns = __popcll(g_runs);
        f.nseg = ngruns + __popcll(straym);
        // segment of this lane: its run, or - a stray - one of its own behind the runs
        f.myseg = g_home ? __popcll(g_runs & le) - 1 : ngruns + __popcll(straym & lt);
        f.rem_r = g_runs; f.rem_s = straym;
#ifndef FB_CYCLE_NO_REGROUP
        if (__popcll(straym) > A.regroup_at) {
            // many strays: one segment per DISTINCT stencil among them (in a wake most of them have moved
            // on to the same two or three neighbouring cells), not one per particle
            unsigned long long rem = straym, leaders = 0ull;
            int ns = 0, mine = 0;
            while (rem) {
                const int l = __builtin_ctzll(rem);
                const int z_ = __builtin_amdgcn_readlane(kz, l), r_ = __builtin_amdgcn_readlane(kr, l);
                const unsigned long long m_ = __ballot(inside && !g_home && kz == z_ && kr == r_);
                if ((m_ >> lane) & 1ull) mine = ns;
                leaders |= 1ull << l;
                ns++;
                rem &= ~m_;
            }
            f.nseg = ngruns + ns;
            if (!g_home) f.myseg = ngruns + mine;
            f.rem_s = leaders;
        }
#endif
#ifndef FB_KNOCK_NODES            // (timing experiment: no node loads for the next chunk)
        if (ask) request(f, min(NSEG, f.nseg));
#endif
    };

    // range of chunks of this wave: [chunk0, chunk0 + cpw_w)
    long chunk0;
    int cpw_w;
    if (!wave_range(A.rg, nwaves, wave, chunk0, cpw_w)) return;
    long base = chunk0 * 64;
    if (base >= n) return;
    // Software pipeline.  Every wait of this kernel for vector memory is ONE s_waitcnt vmcnt(0) per
    // chunk, placed where nothing recent is outstanding: behind the staging of J of chunk c.  By then
    //   * the node values of chunk c+1 (requested after the stencil sums of chunk c),
    //   * the positions / home cells of chunk c+2 and the momenta of chunk c+1 (requested at the top
    //     of chunk c)
    // have been travelling for the gather sums, the front half, the push and the staging, and the
    // stores and atomics of chunk c are issued only AFTER it (they are waited for a chunk later).
    // (The counter is shared by loads, stores and atomics and counts in order: a vmcnt(0) in front
    // of the stencil sums - where it stood first - also waited for the rho atomics issued just
    // before it and for the momenta requested just before it.)
    double xn, yn, zn;                 // positions / home cell of the next chunk
    int hn;
    double xq = 0., yq = 0., zq = 0.;  // ... of the one after it
    int hq = 0;
    double mux, muy, muz, mig;         // momenta, 1/gamma of the next chunk
    double mw = 0.;                    // ... and its weights (read by the first half of the staging, in front of the wait)
    auto load_pos = [&](long b, double &x_, double &y_, double &z_, int &h_) {
        const long i = min(b + lane, n - 1);
        const KPtrs<4> q = karg_ptrs4(KOFF(x));            // x, y, z, home
        x_ = FB_NT_LD(((const double *)q.p[0]) + i); y_ = FB_NT_LD(((const double *)q.p[1]) + i); z_ = FB_NT_LD(((const double *)q.p[2]) + i);
        h_ = FB_NT_LD(((const int *)q.p[3]) + i);
    };
    auto load_mom = [&](long b) {
        const long i = min(b + lane, n - 1);
        const KPtrs<4> q = karg_ptrs4(KOFF(ux));           // ux, uy, uz, ig
        mux = FB_NT_LD(((const double *)q.p[0]) + i); muy = FB_NT_LD(((const double *)q.p[1]) + i); muz = FB_NT_LD(((const double *)q.p[2]) + i);
        mig = FB_NT_LD(((const double *)q.p[3]) + i);
        if constexpr (!RANK) mw = FB_NT_LD(KP(const double, w) + i);
    };
    load_pos(base, xn, yn, zn, hn);
    load_mom(base);
    CycleFront fr;
    front(fr, base, xn, yn, zn, hn, true);
    if (cpw_w > 1) load_pos(base + 64, xn, yn, zn, hn);
    fb_wait_vm();
    unsigned int nstray_J = 0, nbad = 0;
#ifdef FB_CYCLE_TRACE
    unsigned long long tr_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_t = __builtin_readcyclecounter();
    int tr_k = 0;          // slot k: time from mark k - 1 (or the loop end) to mark k
#endif
    for (int ch = 0; ch < cpw_w; ch++) {
        const long i = min(base + lane, n - 1);
        const bool act = base + lane < n;
        const int cnt = fr.cnt;
        const double xj = fr.x, yj = fr.y, zj = fr.z;
        const int hkz = fr.hkz, hkr = fr.hkr, hnb = fr.hnb, myseg = fr.myseg, nseg = fr.nseg;
        const unsigned long long runstarts = fr.runstarts;
        const bool inside = (fr.insidem >> lane) & 1ull;
        FB_MARK("M_TOP");
        const long nbase = base + 64;
        const bool more = (ch + 1 < cpw_w) && nbase < n;
        // momenta and weights of this chunk (requested a chunk ago); the momenta and weights of the next chunk
        // and the positions of the one after it leave now
        double pux = mux, puy = muy, puz = muz, pig = mig;
        const double pw = mw;
        if (more) {
            load_mom(nbase);
            if (ch + 2 < cpw_w) load_pos(nbase + 64, xq, yq, zq, hq);
        }
        // ---- gather: shape factors (threading_methods.py:108-117, 312-321), cos, sin
        double cs, sn, Sz[S], Sr[S];
        int nbel;                     // stencil columns below the axis (0 .. H)
        {
            const double rj = fr.rj;
            const double r_cell = A.invdr * (rj - A.rmin) - 0.5;
            const double z_cell = A.invdz * (zj - A.zmin) - 0.5;
            const int kr = fr.kr, kz = fr.kz;
            if constexpr (SHAPE == FB_SHAPE_LINEAR) {
                Sr[0] = (kr + 1) - r_cell; Sr[1] = r_cell - kr;
                Sz[0] = (kz + 1) - z_cell; Sz[1] = z_cell - kz;
            } else {
                double l = r_cell - kr;
                double a = l - 2., b = l - 1., cc = 2. - l, d = 1. - l;
                Sr[0] = -1. / 6. * (a * (a * a));
                Sr[1] = 1. / 6. * (3. * (b * (b * b)) - 6. * (b * b) + 4.);
                Sr[2] = 1. / 6. * (3. * (cc * (cc * cc)) - 6. * (cc * cc) + 4.);
                Sr[3] = -1. / 6. * (d * (d * d));
                l = z_cell - kz;
                a = l - 2.; b = l - 1.; cc = 2. - l; d = 1. - l;
                Sz[0] = -1. / 6. * (a * (a * a));
                Sz[1] = 1. / 6. * (3. * (b * (b * b)) - 6. * (b * b) + 4.);
                Sz[2] = 1. / 6. * (3. * (cc * (cc * cc)) - 6. * (cc * cc) + 4.);
                Sz[3] = -1. / 6. * (d * (d * d));
            }
            nbel = -kr;
            const double invr = 1. / rj;
            cs = (rj != 0.) ? xj * invr : 1.;
            sn = (rj != 0.) ? yj * invr : 0.;
        }
        // Weights of the S x S nodes.  The stencil columns of a particle near the axis that lie
        // below it (jr < nbel: one for the linear shape, up to two for the cubic one) take the values
        // of their mirror nodes times -(-1)^m (r, t components) or +(-1)^m (z)
        // (gathering/inline_functions.py:70-79, 151-158) - the sign goes to the weight here ((-w) v =
        // -(w v) exactly: the same sums as k_gather's signed panel values)
        double wp[S][S], wm[S][H];
#pragma unroll
        for (int jz = 0; jz < S; jz++) {
#pragma unroll
            for (int jr = 0; jr < S; jr++) wp[jz][jr] = Sz[jz] * Sr[jr];
#pragma unroll
            for (int jr = 0; jr < H; jr++) wm[jz][jr] = (jr < nbel) ? -wp[jz][jr] : wp[jz][jr];
        }
        FB_MARK("M_EVAL");
        double F[6] = {0., 0., 0., 0., 0., 0.};
        for (int s0 = 0; s0 < nseg; s0 += NSEG) {
            const int ns = min(NSEG, nseg - s0);
            // more segments than one round holds (a badly out-of-date order): requested here
            // (the first round was requested a chunk ago and has landed: see the pipeline above)
            if (s0 > 0) {
                request(fr, ns);
                fb_wait_vm();
            }
            wave_lds_release();
            if (inside && myseg >= s0 && myseg < s0 + ns) {
                const double *Pp = (const double *)__builtin_assume_aligned(gpanel + (size_t)(myseg - s0) * PSTR, 16);
                double er_ = 1., ei_ = 0.;            // exptheta_m = (cos - i sin)^m
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    const double factor = (m == 0) ? 1. : 2.;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const double *Pf = Pp + 2 * (m * 6 + k) * S * S;
                        // mirror sign of this field: (k % 3 == 2) ? (-1)^m : -(-1)^m
                        const bool neg = ((k % 3 == 2) ? (m & 1) : !(m & 1));
                        double fr_ = 0., fi_ = 0.;
#pragma unroll
                        for (int jz = 0; jz < S; jz++)
#pragma unroll
                            for (int jr = 0; jr < S; jr++) {
                                const double2 v = *(const double2 *)(Pf + 2 * (jz * S + jr));
                                const double w_ = (jr < H && neg) ? wm[jz][jr] : wp[jz][jr];
                                fr_ = __builtin_fma(w_, v.x, fr_); fi_ = __builtin_fma(w_, v.y, fi_);
                            }
                        // m = 0: exptheta = 1 + 0i and factor = 1 (as in k_gather)
                        if (m == 0) F[k] += fr_;
                        else F[k] += factor * (fr_ * er_ - fi_ * ei_);
                    }
                    const double nr_ = er_ * cs - ei_ * (-sn);
                    const double ni_ = er_ * (-sn) + ei_ * cs;
                    er_ = nr_; ei_ = ni_;
                }
            }
            // every read of the panel has returned before it is written again
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wave_lds_acquire();
        }
        FB_MARK("M_FRONT");
        // ---- front half of the next chunk: its node loads travel during the rest of this one
        if (more) {
            front(fr, nbase, xn, yn, zn, hn, !P::OVERLAY);
        }

        FB_MARK("M_VAY");
        // ---- (r,t) -> (x,y), Vay push, first half push of the position (gather_finish)
        const double ex = cs * F[0] - sn * F[1], ey = sn * F[0] + cs * F[1], ez = F[2];
        const double bx = cs * F[3] - sn * F[4], by = sn * F[3] + cs * F[4], bz = F[5];
        vay(pux, puy, puz, pig, ex, ey, ez, bx, by, bz, A.econst, A.bconst);
        // push/numba_methods.py:28-30 with push_x = push_y = push_z = 1
        const double xh = xj + A.chdt * pig * 1. * pux;
        const double yh = yj + A.chdt * pig * 1. * puy;
        const double zh = zj + A.chdt * pig * 1. * puz;
        // second half push: the positions are written once
        const double x1 = xh + A.chdt * pig * 1. * pux;
        const double y1 = yh + A.chdt * pig * 1. * puy;
        const double z1 = zh + A.chdt * pig * 1. * puz;
        // the chunk's one wait for vector memory (see the pipeline above), then its stores
        auto wait_and_store = [&]() {
            fb_wait_vm();
            xn = xq; yn = yq; zn = zq; hn = hq;
#ifdef FB_KNOCK_STORES
            asm volatile("" :: "v"(pux), "v"(puy), "v"(puz), "v"(pig), "v"(x1), "v"(y1), "v"(z1));
            if (false) {
#else
            if (act) {
#endif
                if (store_eb) {
                    const KPtrs<4> e4 = karg_ptrs4(KOFF(Ex));      // Ex, Ey, Ez, Bx
                    const KPtrs<2> b2 = karg_ptrs2(KOFF(By));      // By, Bz
                    ((double *)e4.p[0])[i] = ex; ((double *)e4.p[1])[i] = ey; ((double *)e4.p[2])[i] = ez;
                    ((double *)e4.p[3])[i] = bx; ((double *)b2.p[0])[i] = by; ((double *)b2.p[1])[i] = bz;
                }
                const KPtrs<8> q = karg_ptrs8(KOFF(x));            // x, y, z, home, ux, uy, uz, ig
                FB_NT_ST(pux, ((double *)q.p[4]) + i); FB_NT_ST(puy, ((double *)q.p[5]) + i); FB_NT_ST(puz, ((double *)q.p[6]) + i);
                FB_NT_ST(pig, ((double *)q.p[7]) + i);
                if constexpr (RANK) {
                    // the second half push belongs to the sort pass that follows (which deposits J
                    // from x(n+1/2) first): the position is left at x(n+1/2)
                    FB_NT_ST(xh, ((double *)q.p[0]) + i); FB_NT_ST(yh, ((double *)q.p[1]) + i); FB_NT_ST(zh, ((double *)q.p[2]) + i);
                } else {
                    FB_NT_ST(x1, ((double *)q.p[0]) + i); FB_NT_ST(y1, ((double *)q.p[1]) + i); FB_NT_ST(z1, ((double *)q.p[2]) + i);
                }
            }
        };
        if constexpr (RANK) {
            wait_and_store();
            // cell of x(n+1) as in k_cell_index / k_bin_rank and the rank inside it: one atomic per
            // run of equal destination cells (gather_finish of particles.hip); the pair is written
            // one chunk later, when the atomic's value has long arrived
            {
                const int b_ = __shfl(pd_base, pd_run0);
                if (pd_i >= 0) { FB_NT_ST(pd_cell, A.rk_cell + pd_i); FB_NT_ST(b_ + (lane - pd_run0), A.rk_rank + pd_i); }
            }
            int rk_c = -1;
            if (act) {
                const double rq = sqrt(x1 * x1 + y1 * y1);
                int ir_upper = (int)ceil(A.invdr * (rq - A.rmin) - 0.5);
                int iz_upper = (int)ceil(A.invdz * (z1 - A.zmin) - 0.5);
                if (ir_upper > Nr) ir_upper = Nr;
                if (iz_upper < 0) iz_upper += Nz;
                else if (iz_upper > Nz - 1) iz_upper -= Nz;
                rk_c = ir_upper + iz_upper * (Nr + 1);
            }
            const int prev = __shfl_up(rk_c, 1);
            const bool rk_start = act && (lane == 0 || rk_c != prev);
            const unsigned long long rstarts = __ballot(rk_start);
            const unsigned long long below = rstarts & le;
            const int rk_run0 = 63 - __builtin_clzll(below | 1ull);
            int rk_base = 0;
            if (rk_start) {
                const unsigned long long rest = (lane + 1 < 64) ? (rstarts >> (lane + 1)) : 0ull;
                const int len = rest ? (__builtin_ctzll(rest) + 1) : (cnt - lane);   // (active lanes: a prefix)
                rk_base = atomicAdd(A.rk_count + rk_c, len);
            }
            pd_i = act ? base + lane : -1;
            pd_cell = rk_c; pd_base = rk_base; pd_run0 = rk_run0;
            if (!more) break;
            base = nbase;
            continue;
        }
        const double wj = act ? A.q * pw : 0.;       // a lane without a particle deposits nothing

        FB_MARK("M_JSTAGE");
        // ---- J from x(n+1/2), rho from x(n+1)
        int dkz, dkr, dnb;
        // (Ruyten coefficients of both depositions: requested here, so that no staging waits for a
        // load - i.e. for the atomics issued before it)
        auto ruyten_index = [&](double xa, double ya) {       // DepEngine::ruyten_index
            const double ra = sqrt(xa * xa + ya * ya);
            return min((int)ceil(A.invdr * (ra - A.rmin) - 0.5), Nr);
        };
        const int irJ = ruyten_index(xh, yh), irR = ruyten_index(x1, y1);
#ifdef FB_KNOCK_BETA              // (timing experiment: no Ruyten loads)
        const double bJ0 = 0.01 * irJ, bJh = 0.02 * irJ, bR0 = 0.01 * irR, bRh = 0.02 * irR;
#else
        const KPtrs<2> bq = karg_ptrs2(KOFF(beta0));           // beta0, betah
        const double bJ0 = ((const double *)bq.p[0])[irJ], bJh = ((const double *)bq.p[1])[irJ];
        const double bR0 = ((const double *)bq.p[0])[irR], bRh = ((const double *)bq.p[1])[irR];
#endif
        if constexpr (!P::MERGED) wait_and_store();
        if constexpr (!P::MERGED) {
            ej.stage_with(true, xh, yh, zh, wj, pux, puy, puz, pig, A.c_light, geom, bJ0, bJh, dkz, dkr, dnb);
            {
                const bool home = act && dkz == hkz && dkr == hkr && dnb == hnb;
                const unsigned long long hm = __ballot(home), sm = __ballot(act && !home);
                nstray_J += __popcll(sm);
                nbad += (__popcll(sm) > FB_CYCLE_BAD_CHUNK) ? 1u : 0u;
                wave_lds_release();
                ej.reduce_home(cnt, runstarts, hm, hkz, hkr, hnb);
                ej.scatter_strays(sm, dkz, dkr, dnb);
                wave_lds_acquire();
            }
            er.stage_with(true, x1, y1, z1, wj, 0., 0., 0., 0., 0., geom, bR0, bRh, dkz, dkr, dnb);
            {
                const bool home = act && dkz == hkz && dkr == hkr && dnb == hnb;
                const unsigned long long hm = __ballot(home), sm = __ballot(act && !home);
                wave_lds_release();
                er.reduce_home(cnt, runstarts, hm, hkz, hkr, hnb);
                er.scatter_strays(sm, dkz, dkr, dnb);
                wave_lds_acquire();
            }
        } else {
            int rkz, rkr, rnb;
            // (the chunk's one wait for vector memory sits INSIDE the staging, in front of the radial shape
            // factors: the Ruyten coefficients requested above travel during the rest of it)
#ifndef FB_CYCLE_WAIT_FIRST
            double rcJ, rcR;
            ed.template stage_pre<0>(xh, yh, zh, wj, pux, puy, puz, pig, A.c_light, geom, dkz, dkr, dnb, rcJ);
        FB_MARK("M_RSTAGE");
            ed.template stage_pre<1>(x1, y1, z1, wj, 0., 0., 0., 0., 0., geom, rkz, rkr, rnb, rcR);
            wait_and_store();
            ed.template stage_post<0>(rcJ, bJ0, bJh);
            ed.template stage_post<1>(rcR, bR0, bRh);
#else
            wait_and_store();
            ed.template stage<0>(xh, yh, zh, wj, pux, puy, puz, pig, A.c_light, geom, bJ0, bJh, dkz, dkr, dnb);
        FB_MARK("M_RSTAGE");
            ed.template stage<1>(x1, y1, z1, wj, 0., 0., 0., 0., 0., geom, bR0, bRh, rkz, rkr, rnb);
#endif
            const bool homeJ = act && dkz == hkz && dkr == hkr && dnb == hnb;
            const bool homeR = act && rkz == hkz && rkr == hkr && rnb == hnb;
            const unsigned long long hmJ = __ballot(homeJ), smJ = __ballot(act && !homeJ);
            const unsigned long long hmR = __ballot(homeR), smR = __ballot(act && !homeR);
            nstray_J += __popcll(smJ);
            nbad += (__popcll(smJ) > FB_CYCLE_BAD_CHUNK) ? 1u : 0u;
            wave_lds_release();
        FB_MARK("M_SCATTER");
#ifndef FB_CYCLE_NO_REGROUP
            // A chunk full of particles that have left their home cells (a laser wake moves a quarter of the
            // electrons to the next cell within a step): one scatter per stray costs several times the run
            // reduction.  Such a chunk is REGROUPED inside the wave: its staged columns are sorted by the pair
            // (cell of the J deposition, cell of the rho deposition) - ranks from one ballot per distinct pair -,
            // and the runs of the reduction are those of the sorted order: every particle takes part in a run of
            // both engines, nothing is scattered (CycleDep::reduce_pairs).  regroup_pairs = 0 (developer
            // override FBPIC_AMD_CYCLE_PAIRS): sorted by the J cell alone, the particles whose rho cell differs
            // remain strays of the rho engine - 0.867 against 0.844 ms per pass at C3.  The particle arrays
            // themselves keep their order.
            if (__popcll(smJ) > A.regroup_at) {
                const unsigned long long actm = __ballot(act);
                int pos = cnt + __popcll(~actm & lt);                 // lanes without a particle: behind
                const bool pairs = A.regroup_pairs != 0;              // wave-uniform
                {
                    unsigned long long rem = actm;
                    int nbefore = 0;
                    while (rem) {
                        const int l = __builtin_ctzll(rem);
                        const int z_ = __builtin_amdgcn_readlane(dkz, l), r_ = __builtin_amdgcn_readlane(dkr, l);
                        const int n_ = __builtin_amdgcn_readlane(dnb, l);
                        bool same = act && dkz == z_ && dkr == r_ && dnb == n_;
                        if (pairs) {
                            const int z2 = __builtin_amdgcn_readlane(rkz, l), r2 = __builtin_amdgcn_readlane(rkr, l);
                            const int n2 = __builtin_amdgcn_readlane(rnb, l);
                            same = same && rkz == z2 && rkr == r2 && rnb == n2;
                        }
                        const unsigned long long m_ = __ballot(same);
                        if ((m_ >> lane) & 1ull) pos = nbefore + __popcll(m_ & lt);
                        nbefore += __popcll(m_);
                        rem &= ~m_;
                    }
                }
                ed.permute_columns(pos);
                // the keys at the sorted positions (lane i sends to lane pos(i))
                const int jz = __builtin_amdgcn_ds_permute(4 * pos, dkz), jr = __builtin_amdgcn_ds_permute(4 * pos, dkr);
                const int jn = __builtin_amdgcn_ds_permute(4 * pos, dnb);
                const int rz = __builtin_amdgcn_ds_permute(4 * pos, rkz), rr = __builtin_amdgcn_ds_permute(4 * pos, rkr);
                const int rn = __builtin_amdgcn_ds_permute(4 * pos, rnb);
                const bool in = lane < cnt;
                const int pz = __shfl_up(jz, 1), pr = __shfl_up(jr, 1), pn = __shfl_up(jn, 1);
                if (pairs) {
                    const int qz = __shfl_up(rz, 1), qr = __shfl_up(rr, 1), qn = __shfl_up(rn, 1);
                    const unsigned long long starts2 = __ballot(in && (lane == 0 || jz != pz || jr != pr || jn != pn ||
                                                                       rz != qz || rr != qr || rn != qn));
                    wave_lds_release();
                    ed.reduce_pairs(cnt, starts2, jz, jr, jn, rz, rr, rn);
                    wave_lds_acquire();
                } else {
                const unsigned long long starts = __ballot(in && (lane == 0 || jz != pz || jr != pr || jn != pn));
                const bool sameR = in && rz == jz && rr == jr && rn == jn;
                const unsigned long long strayR = __ballot(in && !sameR);
                wave_lds_release();
                if (strayR) {
                    ed.scatter_strays(0ull, strayR, jz, jr, jn, rz, rr, rn);
                    ed.zero_amplitudes(false, !sameR);
                    wave_lds_release();
                }
                ed.reduce(cnt, starts, cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull), jz, jr, jn);
                wave_lds_acquire();
                }
            } else
#endif
            if (smJ | smR) {
                // strays first (they read their staged amplitudes), then their amplitudes are zeroed:
                // the products below need no per-particle mask
                ed.scatter_strays(smJ, smR, dkz, dkr, dnb, rkz, rkr, rnb);
                ed.zero_amplitudes(!homeJ, !homeR);
                wave_lds_release();
            }
        FB_MARK("M_REDUCE");
#ifndef FB_CYCLE_NO_REGROUP
            if (!(__popcll(smJ) > A.regroup_at))
#endif
            {
                ed.reduce(cnt, runstarts, hmJ | hmR, hkz, hkr, hnb);
                wave_lds_acquire();
            }
        }
        FB_MARK("M_END");
        if (!more) break;
        if constexpr (P::OVERLAY) {
            // the panel is free again (every LDS read of the depositions has returned): the node values
            // of the next chunk leave now
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            request(fr, min(NSEG, fr.nseg));
        }
        base = nbase;
    }
    if constexpr (RANK) {
        const int b_ = __shfl(pd_base, pd_run0);
        if (pd_i >= 0) { FB_NT_ST(pd_cell, A.rk_cell + pd_i); FB_NT_ST(b_ + (lane - pd_run0), A.rk_rank + pd_i); }
        return;
    }
    if constexpr (!P::MERGED) {
        ej.flush(false);
        er.flush(false);
    } else {
        ed.flush(false);
    }
#ifdef FB_CYCLE_TRACE
    if constexpr (!RANK) {
        if (lane == 0)
            for (int k = 0; k < 9; k++) atomicAdd(&cycle_trace_buf[k], tr_acc[k]);
    }
#endif
    if (A.stats && lane == 0) {
        const int slot = (blockIdx.x * nwaves + wave) & 511;
        atomicAdd(A.stats + slot, (unsigned long long)nstray_J);
        if (nbad) atomicAdd(A.stats + 512 + slot, (unsigned long long)nbad);
    }
}

template <int NM, bool WIDE, bool RANK>
#ifndef FB_CYCLE_ATTR
#define FB_CYCLE_ATTR
#endif
__global__ __launch_bounds__(256) FB_CYCLE_ATTR void k_cycle_linear(CycleArgs A) { cycle_linear_body<FB_SHAPE_LINEAR, NM, WIDE, RANK>(A); }
// cubic shape (round 6): the same pass with the 4 x 4 stencil - lane-by-lane stencil sums from the staged
// node values, the two cubic DepEngines on the home runs
template <int NM, bool WIDE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cycle_cubic(CycleArgs A) { cycle_linear_body<FB_SHAPE_CUBIC, NM, WIDE, false>(A); }
// (The ranking form has no deposition panel - 4.7 KB of LDS per wave - and needs 129-136 VGPRs; asked to
// fit 128 the compiler finds them without a spill at Nm = 2, and a fourth wave per SIMD fits: measured
// SLOWER, 0.165 against 0.156 ms per launch at C2 - more waves in flight span more of the grid than the
// XCD's L2 keeps, as with longer per-wave ranges.  Not used.)

template <int SHAPE, int NM, bool WIDE, bool RANK = false>
static int launch_cycle_linear(const CycleArgs &A0, hipStream_t s)
{
    using P = CyclePlan<SHAPE, NM, WIDE>;
    CycleArgs A = A0;
    const size_t wave_bytes = 8 * (size_t)(RANK ? P::GATHER_DOUBLES : P::WAVE_DOUBLES);
    // one wave per workgroup (nothing is shared between the waves of this kernel): 0.260 against
    // 0.269 ms per launch for the 4-wave workgroups lds_waves_per_workgroup() picks (2 waves: 0.263,
    // 3: 0.266) - finer dispatch granularity at the same 11-12 waves per CU
    const int nwaves = 1;
    const long nchunks = (A.n + 63) / 64;
    // (chunks per wave: 2 .. 6 measure the same, longer per-wave ranges LOSE - 8: +4 %, 11: +22 %, 22:
    // x 2.3 - because the waves in flight then span the whole grid instead of a moving front of it
    // and its node lines leave the XCD's L2)
    const long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    {
        // (FBPIC_AMD_CYCLE_CPW: developer override of the chunks per wave, for the scans of tools/)
        static const int env_cpw = getenv("FBPIC_AMD_CYCLE_CPW") ? atoi(getenv("FBPIC_AMD_CYCLE_CPW")) : 0;
        if (env_cpw > 0) cpw = env_cpw;
    }
    if (cpw < 1) cpw = 1;
    if (cpw > 64) cpw = 64;
    // graded ranges for the linear kernels (both forms): the launch 0.234 -> 0.229 ms at C2 (ranges of 4 chunks), 0.83 ->
    // 0.745 ms at C3 (ranges of 10), profiles/r06_graded_ranges.txt; the cubic kernel keeps the plain cut
    long nblocks;
    A.rg = graded_ranges(nchunks, cpw, nwaves, &nblocks, SHAPE == FB_SHAPE_LINEAR ? 8 : 0);
    if constexpr (SHAPE == FB_SHAPE_CUBIC)
        hipLaunchKernelGGL((k_cycle_cubic<NM, WIDE>), dim3((unsigned)nblocks), dim3(64 * nwaves),
                           wave_bytes * nwaves, s, A);
    else
        hipLaunchKernelGGL((k_cycle_linear<NM, WIDE, RANK>), dim3((unsigned)nblocks), dim3(64 * nwaves),
                           wave_bytes * nwaves, s, A);
    return check(hipGetLastError(), RANK ? "fb_gather_push_rank_next_home" : "fb_gather_push_deposit_J_rho");
}

template <int NM>
static int launch_cycle(const CycleArgs &A, int shape, bool wide, bool rank, hipStream_t s)
{
    constexpr int L = FB_SHAPE_LINEAR, C = FB_SHAPE_CUBIC;
    if (shape == C) return wide ? launch_cycle_linear<C, NM, true>(A, s) : launch_cycle_linear<C, NM, false>(A, s);
    if (rank) return wide ? launch_cycle_linear<L, NM, true, true>(A, s) : launch_cycle_linear<L, NM, false, true>(A, s);
    return wide ? launch_cycle_linear<L, NM, true>(A, s) : launch_cycle_linear<L, NM, false>(A, s);
}

}  // namespace fb

using namespace fb;

#ifdef FB_CYCLE_TRACE
extern "C" int fb_debug_cycle_trace(unsigned long long *host_out, int reset)
{
    int r = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cycle_trace_buf), 16 * 8);
    if (reset) { unsigned long long z[16] = {0}; r |= (int)hipMemcpyToSymbol(HIP_SYMBOL(cycle_trace_buf), z, 16 * 8); }
    return r;
}
#endif

extern "C" int fb_gather_push_deposit_supported(int shape, int Nm)
{
    return (shape == FB_SHAPE_LINEAR || shape == FB_SHAPE_CUBIC) && Nm >= 1 && Nm <= 4;
}

static int cycle_entry(const char *who, bool rank, int shape, int Nm, long n,
        double *x, double *y, double *z, double *ux, double *uy, double *uz, double *inv_gamma,
        const double *w, const int *home_cell,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        double q, double m, double c, double dt, double dt_x, double wrap_zmin, double wrap_zmax,
        void *const *J, long J_row_stride, long J_col_stride,
        void *const *rho, long rho_row_stride, long rho_col_stride,
        const double *ruyten_m0, const double *ruyten_mh, unsigned long long *stats,
        int *rk_cell, int *rk_rank, int *rk_count, int home_cell_shift, void *stream)
{
    if (n <= 0) return 0;
    if (!fb_gather_push_deposit_supported(shape, Nm) || (rank && shape != FB_SHAPE_LINEAR)) {
        set_error(who, rank ? "linear shape, Nm = 1..4 (use the separate entry points otherwise)"
                            : "linear or cubic shape, Nm = 1..4 (use the separate entry points otherwise)");
        return -1;
    }
    if (!home_cell) { set_error(who, "home_cell (cell of every particle at the last sort) is required"); return -1; }
    CycleArgs A;
    bool wide = false;
    A.n = n;
    A.x = x; A.y = y; A.z = z; A.ux = ux; A.uy = uy; A.uz = uz; A.ig = inv_gamma; A.w = w;
    A.home = home_cell;
    A.home_shift = home_cell_shift;
    {
        // (FBPIC_AMD_CYCLE_REGROUP: developer override of the threshold; 64 = never)
        static const int env_at = getenv("FBPIC_AMD_CYCLE_REGROUP") ? atoi(getenv("FBPIC_AMD_CYCLE_REGROUP")) : -1;
        A.regroup_at = env_at >= 0 ? env_at : FB_CYCLE_REGROUP_AT;
        static const int env_pairs = getenv("FBPIC_AMD_CYCLE_PAIRS") ? atoi(getenv("FBPIC_AMD_CYCLE_PAIRS")) : -1;
        A.regroup_pairs = env_pairs >= 0 ? env_pairs : 1;
    }
    A.Ex = Ex; A.Ey = Ey; A.Ez = Ez; A.Bx = Bx; A.By = By; A.Bz = Bz;
    A.invdz = invdz; A.zmin = zmin; A.Nz = Nz; A.invdr = invdr; A.rmin = rmin; A.Nr = Nr;
    A.inv_ncol = 1. / (double)(Nr + 1);
    A.rmax_gather = rmax_gather;
    for (int i = 0; i < 6 * FB_MAX_MODES; i++) A.G.g[i] = i < 6 * Nm ? (const cplx *)grids[i] : nullptr;
    A.rsG = row_stride;
    {
        uintptr_t lo = ~(uintptr_t)0, hi = 0;
        for (int i = 0; i < 6 * Nm; i++) {
            const uintptr_t a = (uintptr_t)grids[i];
            if (a < lo) lo = a;
            if (a > hi) hi = a;
        }
        // (FBPIC_AMD_CYCLE_WIDE=1: the 64-bit addressing whatever the layout - for the tests)
        const char *ew = getenv("FBPIC_AMD_CYCLE_WIDE");
        wide = (ew && atoi(ew) != 0) ||
               (double)(hi - lo) + 16. * (double)row_stride * (double)(Nz + 1) >= 4294967296.;
        A.baseG = (const cplx *)lo;
    }
    // fbpic/particles/push/numba_methods.py:41-42, 24-30
    A.econst = q * dt / (m * c);
    A.bconst = 0.5 * q * dt / m;
    A.chdt = c * dt_x;
    A.wzmin = wrap_zmin; A.wzmax = wrap_zmax;
    A.q = q; A.c_light = c;
    A.GJ.cs = J_col_stride > 0 ? J_col_stride : 1;
    A.GR.cs = rho_col_stride > 0 ? rho_col_stride : 1;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) {
        A.GJ.g[i] = (!rank && i < 3 * Nm) ? (cplx *)J[i] : nullptr;
        A.GR.g[i] = (!rank && i < Nm) ? (cplx *)rho[i] : nullptr;
    }
    A.rsJ = J_row_stride; A.rsR = rho_row_stride;
    A.baseJ = A.baseR = nullptr;
    if (!rank) {
        // the merged engine (cycle_dep.h) addresses J and rho from ONE base with the same strides
        // (the views of one record array, or arrays of one slab); anything else: 64-bit pointers
        DepGrids U;
        U.cs = A.GJ.cs;
        for (int i = 0; i < 3 * Nm; i++) U.g[i] = A.GJ.g[i];
        DepGrids V = A.GR;
        cplx *bj = dep_grids_base(U, 3 * Nm, J_row_stride, Nz), *br = dep_grids_base(V, Nm, rho_row_stride, Nz);
        if (bj && br && J_row_stride == rho_row_stride && A.GJ.cs == A.GR.cs) {
            const uintptr_t lo = (uintptr_t)bj < (uintptr_t)br ? (uintptr_t)bj : (uintptr_t)br;
            uintptr_t hi = 0;
            for (int i = 0; i < 3 * Nm; i++) if ((uintptr_t)A.GJ.g[i] > hi) hi = (uintptr_t)A.GJ.g[i];
            for (int i = 0; i < Nm; i++) if ((uintptr_t)A.GR.g[i] > hi) hi = (uintptr_t)A.GR.g[i];
            if ((double)(hi - lo) + 16. * (double)J_row_stride * (double)(Nz + 1) < 4294967296.)
                A.baseJ = A.baseR = (cplx *)lo;
        }
        if (!A.baseJ) wide = true;
    }
    A.beta0 = ruyten_m0; A.betah = ruyten_mh;
    A.rg = WaveRanges{1, -1, 1, 0, 0};
    A.stats = stats;
    A.rk_cell = rk_cell; A.rk_rank = rk_rank; A.rk_count = rk_count;
    hipStream_t s = (hipStream_t)stream;
    switch (Nm) {
    case 1: return launch_cycle<1>(A, shape, wide, rank, s);
    case 2: return launch_cycle<2>(A, shape, wide, rank, s);
    case 3: return launch_cycle<3>(A, shape, wide, rank, s);
    default: return launch_cycle<4>(A, shape, wide, rank, s);
    }
}

extern "C" int fb_gather_push_deposit_J_rho(int shape, int Nm, long n,
        double *x, double *y, double *z, double *ux, double *uy, double *uz, double *inv_gamma,
        const double *w, const int *home_cell,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        double q, double m, double c, double dt, double dt_x, double wrap_zmin, double wrap_zmax,
        void *const *J, long J_row_stride, long J_col_stride,
        void *const *rho, long rho_row_stride, long rho_col_stride,
        const double *ruyten_m0, const double *ruyten_mh, unsigned long long *stats,
        int home_cell_shift, void *stream)
{
    return cycle_entry("fb_gather_push_deposit_J_rho", false, shape, Nm, n, x, y, z, ux, uy, uz, inv_gamma, w,
                       home_cell, rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, grids, row_stride, Ex, Ey, Ez,
                       Bx, By, Bz, q, m, c, dt, dt_x, wrap_zmin, wrap_zmax, J, J_row_stride, J_col_stride, rho,
                       rho_row_stride, rho_col_stride, ruyten_m0, ruyten_mh, stats, nullptr, nullptr, nullptr,
                       home_cell_shift, stream);
}

extern "C" int fb_gather_push_rank_next_home(int shape, int Nm, long n,
        double *x, double *y, double *z, double *ux, double *uy, double *uz, double *inv_gamma,
        const int *home_cell,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        double q, double m, double c, double dt, double dt_x, double wrap_zmin, double wrap_zmax,
        int ncell, void *sort_workspace, size_t workspace_bytes, int counts_are_zero,
        int home_cell_shift, void *stream)
{
    const char *who = "fb_gather_push_rank_next_home";
    hipStream_t s = (hipStream_t)stream;
    if (ncell != Nz * (Nr + 1)) { set_error(who, "ncell != Nz*(Nr+1)"); return -1; }
    if (workspace_bytes < fb_bin_sort_workspace_bytes(n, ncell)) { set_error(who, "workspace too small"); return -1; }
    const BinSortWs W = carve_bin_sort_ws(sort_workspace, workspace_bytes, n, ncell);
    if (!counts_are_zero) {
        hipError_t e = hipMemsetAsync(W.count, 0, (size_t)ncell * sizeof(int), s);
        if (e != hipSuccess) return check(e, who);
    }
    return cycle_entry(who, true, shape, Nm, n, x, y, z, ux, uy, uz, inv_gamma, x /* w: not read */, home_cell,
                       rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, grids, row_stride, Ex, Ey, Ez, Bx, By, Bz,
                       q, m, c, dt, dt_x, wrap_zmin, wrap_zmax, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr,
                       nullptr, W.cell, W.rank, W.count, home_cell_shift, stream);
}
