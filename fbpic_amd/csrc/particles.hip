// Particle kernels for gfx950: position/momentum push, periodic shift, field gather.
// One lane = one (or two, 16-byte vectorised) macroparticle(s); structure-of-arrays
// float64 streams, coalesced; grid-stride over a capped grid (256 CUs x 8 blocks).
#include <cstdlib>
#include "fb_common.h"
#include "push_common.h"

namespace fb {

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ------------------------------------------------------------------ push_x
// Reference arithmetic: x += (c*dt) * inv_gamma * push * ux, left to right
// (fbpic/particles/push/numba_methods.py:25-30).  80 B / particle.
template <int V>
__global__ __launch_bounds__(256) void k_push_x(long nvec, double *__restrict__ x,
        double *__restrict__ y, double *__restrict__ z, const double *__restrict__ ux,
        const double *__restrict__ uy, const double *__restrict__ uz,
        const double *__restrict__ ig, double chdt, double px, double py, double pz)
{
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        if constexpr (V == 2) {
            double2 g = ((const double2 *)ig)[i];
            double2 a = ((double2 *)x)[i], u = ((const double2 *)ux)[i];
            a.x += chdt * g.x * px * u.x; a.y += chdt * g.y * px * u.y;
            ((double2 *)x)[i] = a;
            a = ((double2 *)y)[i]; u = ((const double2 *)uy)[i];
            a.x += chdt * g.x * py * u.x; a.y += chdt * g.y * py * u.y;
            ((double2 *)y)[i] = a;
            a = ((double2 *)z)[i]; u = ((const double2 *)uz)[i];
            a.x += chdt * g.x * pz * u.x; a.y += chdt * g.y * pz * u.y;
            ((double2 *)z)[i] = a;
        } else {
            double g = ig[i];
            x[i] += chdt * g * px * ux[i];
            y[i] += chdt * g * py * uy[i];
            z[i] += chdt * g * pz * uz[i];
        }
    }
}

// ------------------------------------------------------------------ push_p (vay(): push_common.h)
template <int V>
__global__ __launch_bounds__(256) void k_push_p(long nvec, double *__restrict__ ux,
        double *__restrict__ uy, double *__restrict__ uz, double *__restrict__ ig,
        const double *__restrict__ Ex, const double *__restrict__ Ey,
        const double *__restrict__ Ez, const double *__restrict__ Bx,
        const double *__restrict__ By, const double *__restrict__ Bz,
        double econst, double bconst)
{
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        if constexpr (V == 2) {
            double2 a = ((double2 *)ux)[i], b = ((double2 *)uy)[i], cc = ((double2 *)uz)[i],
                    g = ((double2 *)ig)[i];
            double2 ex = ((const double2 *)Ex)[i], ey = ((const double2 *)Ey)[i],
                    ez = ((const double2 *)Ez)[i], bx = ((const double2 *)Bx)[i],
                    by = ((const double2 *)By)[i], bz = ((const double2 *)Bz)[i];
            vay(a.x, b.x, cc.x, g.x, ex.x, ey.x, ez.x, bx.x, by.x, bz.x, econst, bconst);
            vay(a.y, b.y, cc.y, g.y, ex.y, ey.y, ez.y, bx.y, by.y, bz.y, econst, bconst);
            ((double2 *)ux)[i] = a; ((double2 *)uy)[i] = b; ((double2 *)uz)[i] = cc;
            ((double2 *)ig)[i] = g;
        } else {
            double a = ux[i], b = uy[i], cc = uz[i], g = ig[i];
            vay(a, b, cc, g, Ex[i], Ey[i], Ez[i], Bx[i], By[i], Bz[i], econst, bconst);
            ux[i] = a; uy[i] = b; uz[i] = cc; ig[i] = g;
        }
    }
}

// fbpic/boundaries/particle_buffer_handling.py:536-556
__global__ __launch_bounds__(256) void k_shift_periodic(long n, double *__restrict__ z,
                                                        double zmin, double zmax)
{
    const double l_box = zmax - zmin;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double zi = z[i];
        bool ch = false;
        while (zi >= zmax) { zi -= l_box; ch = true; }
        while (zi < zmin) { zi += l_box; ch = true; }
        if (ch) z[i] = zi;
    }
}

// ------------------------------------------------------------------ gather
// Field gather for any number of azimuthal modes in ONE pass over the particles (the
// reference needs Nm launches + an erase for Nm != 2).  Semantics:
// fbpic/particles/gathering/threading_methods.py:25-201 (linear), :207-367 (cubic),
// inline_functions.py:9-187; exptheta_m by recurrence; modes are summed before the
// (r,t)->(x,y) rotation as in the Nm==2 kernels.
//
// MI355X design: particles arrive cell-sorted, so the 64 particles of a wave span only a
// few cells ("segments").  All particles of a cell read the SAME S x S grid nodes of the
// 6*Nm field arrays.  Instead of 48 (linear, Nm=2) 16-byte global loads per lane through
// the 64 B/clk vector-L1 path, the wave
//   1. finds its segments with one ballot on the (iz, ir) stencil origin,
//   2. stages, per segment, the S*S*6*Nm complex node values ONCE into an LDS panel
//      (lane = node value; below-axis mirror sign, r clamp and z wrap applied here),
//   3. lets every lane evaluate its stencil from its segment's panel with broadcast
//      ds_read_b128 (256 B/clk), so the kernel is bound by the particle streams
//      (24 B read + 48 B written per particle).
// Any order of particles gives the same result; an unsorted stream just has more segments.

// Optional fusion of the two kernels that follow the gather in the PIC cycle
// (main.py:469-490): Vay push_p with the fields still in registers, then push_x over dt_x.
// The particle E,B arrays are still written when their pointers are non-null.
struct PushArgs {
    double *x, *y, *z;            // writable aliases of the position arrays (push_x)
    double *ux, *uy, *uz, *ig;    // null -> gather only
    double econst, bconst;        // q dt / (m c), q dt / (2 m)
    double chdt;                  // c * dt_x ; 0 -> no position push
    // periodic wrap of z into [wzmin, wzmax) before the gather (k_shift_periodic folded in;
    // only with the position push, which rewrites z anyway); wzmax <= wzmin -> off
    double wzmin, wzmax;
    // cell + rank of the position after the NEXT push_x, for the counting sort that follows it
    // (fb_gather_push_rank_next); RK.count == null -> off
    RankNext RK;
    // Restriction of the pass to a contiguous range [*range_lo, *range_hi) of the (cell-sorted)
    // particle arrays (range_mode 1) or to its complement (2); the bounds are read on the
    // device - entries of the per-cell prefix sum - so that no offset travels to the host.
    // A null pointer stands for 0.  range_mode 0: every particle (fb_gather_push_range).
    const int *range_lo, *range_hi;
    int range_mode;
};


template <int SHAPE> struct GShape;
template <> struct GShape<FB_SHAPE_LINEAR> { static constexpr int S = 2, OFF = 0; };
template <> struct GShape<FB_SHAPE_CUBIC> { static constexpr int S = 4, OFF = 1; };

constexpr int G_NOKEY = -0x40000000;

// Tail of a gather chunk, shared by the gather kernels: (r,t) -> (x,y) rotation of the gathered
// fields, optional store, Vay push_p with the fields still in registers, push_x, and the cell /
// rank of the position after the NEXT push_x (see PushArgs).
// The rank of a particle = the value returned by the per-run atomic on its destination cell.  The
// round trip of that atomic is not waited for at the end of the chunk: the (cell, rank) pair is
// written at the start of the NEXT chunk, when the value has long arrived.
struct RankPending {
    long i;          // particle index of this lane in the chunk the ranks belong to (-1: none)
    int cell, base, run0;
};

__device__ __forceinline__ void rank_commit(RankPending &pd, int lane, const PushArgs &PA)
{
    if (!PA.RK.count) return;
    const int b = __shfl(pd.base, pd.run0);
    if (pd.i >= 0) {
        FB_NT_ST(pd.cell, PA.RK.cell + pd.i);
        FB_NT_ST(b + (lane - pd.run0), PA.RK.rank + pd.i);
    }
    pd.i = -1;
}

template <bool DEFER>
__device__ __forceinline__ void gather_finish(bool act, long i, int lane, double xj, double yj,
        double zj, double cs, double sn, const double *F,
        double *__restrict__ Ex, double *__restrict__ Ey, double *__restrict__ Ez,
        double *__restrict__ Bx, double *__restrict__ By, double *__restrict__ Bz,
        const PushArgs &PA, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        RankPending &pd,
        const double *pre = nullptr)     // ux, uy, uz, inv_gamma already in registers
{
    int rk_c = -1;
    if (act) {
        const double ex = cs * F[0] - sn * F[1], ey = sn * F[0] + cs * F[1], ez = F[2];
        const double bx = cs * F[3] - sn * F[4], by = sn * F[3] + cs * F[4], bz = F[5];
        if (Ex) { FB_NT_ST(ex, Ex + i); FB_NT_ST(ey, Ey + i); FB_NT_ST(ez, Ez + i); FB_NT_ST(bx, Bx + i); FB_NT_ST(by, By + i); FB_NT_ST(bz, Bz + i); }
        if (PA.ux) {
            // momenta are loaded here, not at the top of the chunk: 8 VGPRs less across the
            // stencil phase; the other waves of the SIMD cover the latency (measured: -3 %)
            double pux, puy, puz, pig;
            if (pre) { pux = pre[0]; puy = pre[1]; puz = pre[2]; pig = pre[3]; }
            else { pux = FB_NT_LD(PA.ux + i); puy = FB_NT_LD(PA.uy + i); puz = FB_NT_LD(PA.uz + i); pig = FB_NT_LD(PA.ig + i); }
            vay(pux, puy, puz, pig, ex, ey, ez, bx, by, bz, PA.econst, PA.bconst);
            FB_NT_ST(pux, PA.ux + i); FB_NT_ST(puy, PA.uy + i); FB_NT_ST(puz, PA.uz + i); FB_NT_ST(pig, PA.ig + i);
            if (PA.chdt != 0.) {
                // numba_methods.py:28-30 with push_x = push_y = push_z = 1
                const double xp = xj + PA.chdt * pig * 1. * pux;
                const double yp = yj + PA.chdt * pig * 1. * puy;
                const double zp = zj + PA.chdt * pig * 1. * puz;
                FB_NT_ST(xp, PA.x + i); FB_NT_ST(yp, PA.y + i); FB_NT_ST(zp, PA.z + i);
                if (PA.RK.count) {
                    // position after the coming push_x, cell as in k_cell_index / k_bin_rank
                    const double xq = xp + PA.RK.chdt * pig * PA.RK.px * pux;
                    const double yq = yp + PA.RK.chdt * pig * PA.RK.py * puy;
                    const double zq = zp + PA.RK.chdt * pig * PA.RK.pz * puz;
                    const double rq = sqrt(xq * xq + yq * yq);
                    int ir_upper = (int)ceil(invdr * (rq - rmin) - 0.5);
                    int iz_upper = (int)ceil(invdz * (zq - zmin) - 0.5);
                    if (ir_upper > Nr) ir_upper = Nr;
                    if (iz_upper < 0) iz_upper += Nz;
                    else if (iz_upper > Nz - 1) iz_upper -= Nz;
                    rk_c = ir_upper + iz_upper * (Nr + 1);
                }
            }
        }
    }
    if (PA.RK.count) {
        // one atomic per run of equal destination cells (wave-uniform branch)
        const int prev = __shfl_up(rk_c, 1);
        const bool rk_start = act && (lane == 0 || rk_c != prev);
        const unsigned long long rstarts = __ballot(rk_start);
        const unsigned long long actm = __ballot(act);
        const unsigned long long below = rstarts & ((2ull << lane) - 1ull);
        const int rk_run0 = 63 - __builtin_clzll(below | 1ull);
        int rk_base = 0;
        if (rk_start) {
            // length of the run = active lanes from here to the next start.  (The active lanes
            // are a prefix of the wave in a pass over all particles, but a suffix or a prefix
            // plus a suffix in a range-restricted pass: count them, do not subtract lane numbers.)
            const unsigned long long rest = (lane + 1 < 64) ? (rstarts >> (lane + 1)) : 0ull;
            const int span = rest ? (__builtin_ctzll(rest) + 1) : (64 - lane);
            const unsigned long long in_run = (span >= 64 ? ~0ull : ((1ull << span) - 1ull)) << lane;
            const int len = __popcll(actm & in_run);
            rk_base = atomicAdd(PA.RK.count + rk_c, len);
        }
        if constexpr (DEFER) {
            pd.i = act ? i : -1;
            pd.cell = rk_c; pd.base = rk_base; pd.run0 = rk_run0;
        } else {
            rk_base = __shfl(rk_base, rk_run0);
            if (act) {
                FB_NT_ST(rk_c, PA.RK.cell + i);
                FB_NT_ST(rk_base + (lane - rk_run0), PA.RK.rank + i);
            }
        }
    }
}


// NMT > 0: number of modes known at compile time (mode loop unrolled, exptheta_0 = 1 folded
// away); NMT = 0: run-time Nm.
//
// The kernel is bound by latency and instruction issue, not by HBM (rocprofv3 + ISA census,
// DESIGN.md section 6), so the staging of a chunk is arranged as ONE round trip to L2:
//   * the keys of the segments are wave-uniform scalars (v_readlane of the start lanes found
//     by the ballot): no LDS exchange, row/column arithmetic partly on the scalar unit;
//   * a lane's role in the staging (which field / node it fetches, its pointer and its
//     below-axis sign) does not depend on the chunk and is set up once per wave;
//   * the loads of all segments of a round are issued back to back, and the work that does
//     not need them (1/r, cos, sin, the products of the shape factors) sits between the
//     issue and the first use.
// (forcing 5 / 6 waves per SIMD with amdgpu_waves_per_eu: 107 VGPRs -> spills, measured
// 94 / 147 us against 96 us for the fused Nm = 2 kernel)
template <int SHAPE, int NMT>
__global__ __launch_bounds__(256) void k_gather(int Nm_arg, long n,
        const double *__restrict__ x, const double *__restrict__ y,
        const double *__restrict__ z, double rmax_gather,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        GatherGrids G, long rs,
        double *__restrict__ Ex, double *__restrict__ Ey, double *__restrict__ Ez,
        double *__restrict__ Bx, double *__restrict__ By, double *__restrict__ Bz,
        int maxseg_arg, int chunks_per_wave, PushArgs PA)
{
    constexpr int S = GShape<SHAPE>::S, OFF = GShape<SHAPE>::OFF;
    const int Nm = NMT ? NMT : Nm_arg;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar loop bounds
    const int NV = S * S * 6 * Nm;                 // complex node values of one segment
    const int PSTR = 2 * NV + 2;                   // panel stride in doubles (16-B pad)
    // FAST: one node value per lane and segment (linear shape, Nm <= 2)
    constexpr bool FAST = NMT > 0 && S * S * 6 * NMT <= 64;
    constexpr int NVL = NMT ? (S * S * 6 * NMT + 63) / 64 : 1;   // node values per lane
    const int maxseg = FAST ? 4 : maxseg_arg;
    double *panel = lds + (size_t)wave * ((size_t)maxseg * PSTR);

    // staging role of this lane, FAST path: node value o = lane of every segment
    bool st_on = false;
    int st_jr = 0, st_jz = 0;
    const cplx *st_ptr = nullptr;
    double st_sgn = 1.;
    if constexpr (FAST) {
        st_on = lane < S * S * 6 * NMT;
        const int o = st_on ? lane : 0;
        st_jr = o % S; st_jz = (o / S) % S;
        const int f = o / (S * S);
        st_ptr = G.g[f];
        // mirror below the axis: -(-1)^m for r,t components, +(-1)^m for z
        // (inline_functions.py:70-79, 151-158)
        const int m_ = f / 6, k_ = f - 6 * m_;
        const double flip = m1pow(m_);
        st_sgn = (k_ % 3 == 2) ? flip : -flip;
    }

    const long chunk0 = (xcd_block_id() * nwaves + wave) * chunks_per_wave;
    // software pipeline: the particle coordinates of chunk ch+1 are requested before the
    // work on chunk ch starts, so their HBM latency hides behind staging + stencil math
    double xn = 0., yn = 0., zn = 0.;
    if (chunk0 * 64 + lane < n) { xn = FB_NT_LD(x + (chunk0 * 64 + lane)); yn = FB_NT_LD(y + (chunk0 * 64 + lane)); zn = FB_NT_LD(z + (chunk0 * 64 + lane)); }
    RankPending pend = {-1, 0, 0, 0};
    bool have_next = true;        // xn, yn, zn hold the coordinates of the chunk about to start
    long r_lo = 0, r_hi = n;
    if (PA.range_mode) {
        r_lo = PA.range_lo ? (long)*PA.range_lo : 0;
        r_hi = PA.range_hi ? (long)*PA.range_hi : 0;
        if (r_hi < r_lo) r_hi = r_lo;
    }
    for (int ch = 0; ch < chunks_per_wave; ch++) {
        const long base = (chunk0 + ch) * 64;
        if (base >= n) break;
        const long i = base + lane;
        bool act = i < n;
        if (PA.range_mode) {
            const bool in_range = i >= r_lo && i < r_hi;
            act = act && (PA.range_mode == 1 ? in_range : !in_range);
            // chunks without a particle of this pass (wave-uniform test)
            const bool none = PA.range_mode == 1 ? (base + 64 <= r_lo || base >= r_hi)
                                                 : (base >= r_lo && base + 64 <= r_hi);
            if (none) { have_next = false; continue; }      // (nothing is loaded for it)
            if (!have_next && i < n) { xn = FB_NT_LD(x + i); yn = FB_NT_LD(y + i); zn = FB_NT_LD(z + i); }
            have_next = true;
        }
        double cs = 1., sn = 0., Sz[S], Sr[S];
        int kz = G_NOKEY, kr = G_NOKEY;
        bool inside = false;
        const double xj = xn, yj = yn;
        double zj = zn;
        if (PA.wzmax > PA.wzmin) {
            const double l_box = PA.wzmax - PA.wzmin;
            while (zj >= PA.wzmax) zj -= l_box;
            while (zj < PA.wzmin) zj += l_box;
        }
        if (ch + 1 < chunks_per_wave && i + 64 < n) { xn = FB_NT_LD(x + (i + 64)); yn = FB_NT_LD(y + (i + 64)); zn = FB_NT_LD(z + (i + 64)); }
        rank_commit(pend, lane, PA);            // ranks of the previous chunk
        double rj = 0.;
        if (act) {
            rj = sqrt(xj * xj + yj * yj);
            const double r_cell = invdr * (rj - rmin) - 0.5;
            const double z_cell = invdz * (zj - zmin) - 0.5;
            inside = rj < rmax_gather;
            kr = (int)floor(r_cell) - OFF;
            kz = (int)floor(z_cell) - OFF;
            if constexpr (SHAPE == FB_SHAPE_LINEAR) {
                // threading_methods.py:108-117
                Sr[0] = (kr + 1) - r_cell; Sr[1] = r_cell - kr;
                Sz[0] = (kz + 1) - z_cell; Sz[1] = z_cell - kz;
            } else {
                // threading_methods.py:312-321
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
            if (!inside) { kz = G_NOKEY; kr = G_NOKEY; }   // gathers nothing: no segment
        }
        // segments = runs of equal stencil origin among the lanes that gather
        const int pkz = __shfl_up(kz, 1), pkr = __shfl_up(kr, 1);
        const bool is_start = inside && (lane == 0 || kz != pkz || kr != pkr);
        const unsigned long long starts = __ballot(is_start);
        const int nseg = __popcll(starts);
        const int myseg = __popcll(starts & ((2ull << lane) - 1ull)) - 1;   // valid if inside
        unsigned long long rem = starts;          // start lanes of the segments not yet staged
        double F[6] = {0., 0., 0., 0., 0., 0.};

        // ---- staging of one round of segments: issue (loads in flight) / commit (to LDS)
        double2 vals[FAST ? 4 : NVL];
        double vsgn[FAST ? 4 : NVL];
        auto node_addr = [&](const cplx *fp, int skz, int skr, int jz, int jr, double sgn_below,
                             double &sgn) -> const cplx * {
            int row = skz + jz, col = skr + jr;
            if (row < 0) row += Nz; else if (row > Nz - 1) row -= Nz;
            sgn = 1.;
            if (col < 0) { col = -col - 1; sgn = sgn_below; }
            else if (col > Nr - 1) col = Nr - 1;
            return fp + (long)row * rs + col;
        };
        auto next_key = [&](int &skz, int &skr) {
            const int l = __builtin_ctzll(rem);
            rem &= rem - 1ull;
            skz = __builtin_amdgcn_readlane(kz, l);
            skr = __builtin_amdgcn_readlane(kr, l);
        };
        auto issue_fast = [&](int ns) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                vals[u] = make_double2(0., 0.);
                vsgn[u] = 1.;
                if (u < ns) {
                    int skz, skr;
                    next_key(skz, skr);
                    if (st_on) vals[u] = ldc(node_addr(st_ptr, skz, skr, st_jz, st_jr, st_sgn, vsgn[u]));
                }
            }
        };
        auto commit_fast = [&](int ns) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (u < ns && st_on) {
                    double2 v = vals[u];
                    v.x *= vsgn[u]; v.y *= vsgn[u];
                    *(double2 *)(panel + (size_t)u * PSTR + 2 * lane) = v;
                }
        };
        // general path: NV > 64 node values per segment, one segment at a time
        auto stage_general = [&](int ns) {
            for (int sg = 0; sg < ns; sg++) {
                int skz, skr;
                next_key(skz, skr);
                if constexpr (NMT > 0) {
#pragma unroll
                    for (int j = 0; j < NVL; j++) {
                        const int o = lane + 64 * j;
                        vals[j] = make_double2(0., 0.);
                        vsgn[j] = 1.;
                        if (o < NV) {
                            const int jr = o % S, jz = (o / S) % S, f = o / (S * S);
                            const int m_ = f / 6, k_ = f - 6 * m_;
                            const double flip = m1pow(m_);
                            vals[j] = ldc(node_addr(G.g[f], skz, skr, jz, jr,
                                                    (k_ % 3 == 2) ? flip : -flip, vsgn[j]));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NVL; j++) {
                        const int o = lane + 64 * j;
                        if (o < NV) {
                            double2 v = vals[j];
                            v.x *= vsgn[j]; v.y *= vsgn[j];
                            *(double2 *)(panel + (size_t)sg * PSTR + 2 * o) = v;
                        }
                    }
                } else {
                    for (int o = lane; o < NV; o += 64) {
                        const int jr = o % S, jz = (o / S) % S, f = o / (S * S);
                        const int m_ = f / 6, k_ = f - 6 * m_;
                        const double flip = m1pow(m_);
                        double sgn;
                        double2 v = ldc(node_addr(G.g[f], skz, skr, jz, jr,
                                                  (k_ % 3 == 2) ? flip : -flip, sgn));
                        v.x *= sgn; v.y *= sgn;
                        *(double2 *)(panel + (size_t)sg * PSTR + 2 * o) = v;
                    }
                }
            }
        };
        // ---- stencil evaluation of the lanes whose segment is in the staged round
        auto evaluate = [&](int s0, int ns) {
            if (inside && myseg >= s0 && myseg < s0 + ns) {
                const double *P = panel + (size_t)(myseg - s0) * PSTR;
                double er = 1., ei = 0.;            // exptheta_m = (cos - i sin)^m
#pragma unroll
                for (int m = 0; m < Nm; m++) {
                    const double factor = (m == 0) ? 1. : 2.;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const double *Pf = P + 2 * (m * 6 + k) * S * S;
                        double fr = 0., fi = 0.;
                        if constexpr (SHAPE == FB_SHAPE_LINEAR) {
#pragma unroll
                            for (int jz = 0; jz < S; jz++)
#pragma unroll
                                for (int jr = 0; jr < S; jr++) {
                                    const double2 v = *(const double2 *)(Pf + 2 * (jz * S + jr));
                                    const double w_ = Sz[jz] * Sr[jr];
                                    fr = __builtin_fma(w_, v.x, fr); fi = __builtin_fma(w_, v.y, fi);
                                }
                        } else {
#pragma unroll
                            for (int jr = 0; jr < S; jr++)
#pragma unroll
                                for (int jz = 0; jz < S; jz++) {
                                    const double2 v = *(const double2 *)(Pf + 2 * (jz * S + jr));
                                    const double w_ = Sz[jz] * Sr[jr];
                                    fr = __builtin_fma(w_, v.x, fr); fi = __builtin_fma(w_, v.y, fi);
                                }
                        }
                        // m = 0: exptheta = 1 + 0i and factor = 1, so the term is fr exactly
                        // (fr * 1 - fi * 0 for finite fields); skipping the products also lets
                        // the compiler drop the imaginary accumulation of mode 0
                        if (NMT && m == 0) F[k] += fr;
                        else F[k] += factor * (fr * er - fi * ei);
                    }
                    const double nr_ = er * cs - ei * (-sn);
                    const double ni_ = er * (-sn) + ei * cs;
                    er = nr_; ei = ni_;
                }
            }
        };

        const int ns0 = min(maxseg, nseg);
        if constexpr (FAST) issue_fast(ns0);
        // independent of the staged values: overlaps the L2 round trip of the loads above
        if (act && rj != 0.) { const double invr = 1. / rj; cs = xj * invr; sn = yj * invr; }
        if constexpr (FAST) commit_fast(ns0); else stage_general(ns0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        evaluate(0, ns0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int s0 = maxseg; s0 < nseg; s0 += maxseg) {     // more segments than one round holds
            const int ns = min(maxseg, nseg - s0);
            if constexpr (FAST) { issue_fast(ns); commit_fast(ns); } else stage_general(ns);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            evaluate(s0, ns);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        gather_finish<true>(act, i, lane, xj, yj, zj, cs, sn, F, Ex, Ey, Ez, Bx, By, Bz, PA,
                            invdz, zmin, Nz, invdr, rmin, Nr, pend);
    }
    rank_commit(pend, lane, PA);
}

// ------------------------------------------------------------------ cubic gather on the matrix cores
// With the cubic shape every particle reads 16 nodes x 6 Nm complex node values: evaluated lane by
// lane from the LDS panel (k_gather) that is 384 ds_read_b128 per particle for Nm = 4, and the LDS
// return path (128 B / clk / CU) is what bounds the kernel (SQ counters, 2048 x 512 x 64 ppc:
// 371 M LDS instructions in 5.5 ms = the LDS pipe ~ 100 % busy, VALU 48 %).  The particles of a
// segment (same stencil origin) all multiply the SAME node values, so the stencil sum is a small
// matrix product per segment,
//     D[column c][particle p] = sum over the 16 nodes of G[node][c] * W[node][p],
// with c = the 12 Nm real columns (field, mode, re / im) and W = Sz[jz] * Sr[jr]: it runs on
// v_mfma_f64_16x16x4_f64 with the node values as the A operand (read from the LDS panel, XOR-
// swizzled so that neither the staging writes nor the fragment reads conflict) and the weights
// of 16 particles as the B operand: 12 Nt ds_read_b64 per (segment, group of 16) instead of
// 96 Nm ds_read_b128 per particle.
// Measured (2048 x 512, 64 ppc, gather + push + rank): Nm = 4 5.56 -> 5.10 ms, Nm = 2 3.72 ->
// 3.37 ms.  Not more, because fp64 MFMA and fp64 VALU instructions do NOT overlap on gfx950
// (tools/overlap_probe.hip: one MFMA wave + one FMA wave per SIMD take the SUM of their solo
// times; both run on the same DP units, and both peak at 78.6 TFLOP/s): the matrix form saves
// the LDS traffic, not issue cycles - per chunk of 64 particles ~55 MFMAs x 64 cycles (groups
// that straddle a segment boundary are multiplied twice) + ~1100 VALU instructions (phase 1,
// staging, fold, push) = 8.0 k DP-pipe cycles, of which the kernel achieves 75 %.
//
// Column order: D row (l >> 4) + 4 r of tile t belongs to lane quarter a = l >> 4.  A quarter is
// given all the columns of ONE azimuthal mode (Nm = 3, 4: 6 complex fields, 3 tiles) or of one
// mode and one of E / B (Nm = 2: 3 complex fields, 2 tiles): lane (a, j) then holds, for particle
// j of the group, fr / fi of its fields in its own accumulator registers, multiplies by its mode's
// exptheta (staged per particle by phase 1) and the sum over the modes is two cross-quarter
// shuffles.  The quarter that holds particle 16 g + j as its OWN particle keeps the result, so
// the (r,t) -> (x,y) rotation, the push and the ranking continue lane = particle as in k_gather.
// sums over the lane quarters (lanes 16 / 32 apart) with the gfx950 row swaps: no LDS round trip
__device__ __forceinline__ double sum_xor32(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double sum_xor16(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// value of the lane 16 further / nearer (quarter a <-> a ^ 1)
__device__ __forceinline__ double other_xor16(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    // vdst keeps its even rows and receives the even rows of src in its odd rows; src the reverse
    const bool odd = (threadIdx.x >> 4) & 1;
    return odd ? __hiloint2double(b[0], a[0]) : __hiloint2double(b[1], a[1]);
}

template <int NMT> struct GMX {
    static_assert(NMT >= 2 && NMT <= 4, "matrix-core gather: 2 <= Nm <= 4");
    static constexpr int QM = (NMT == 2) ? 2 : 1;      // lane quarters per mode
    static constexpr int FPL = 6 / QM;                 // complex fields per lane
    static constexpr int NT = (2 * FPL + 3) / 4;       // 16-column tiles
    static constexpr int RS = 48;                      // panel row stride (doubles): rows of a
                                                       // fragment alternate between the bank halves
    static constexpr int NV = 16 * 6 * NMT;            // complex node values of a segment
    static constexpr int NVL = (NV + 63) / 64;         // ... per lane
    static constexpr int WROWS = 8 + 2 * (NMT - 1);    // Sz[4] | Sr[4] | (er, ei) of modes 1..
    static constexpr int WPAD = 65;
    static constexpr int WAVE_DOUBLES = 16 * RS + WROWS * WPAD + 1;
    // Panel element (node, column): the column is XOR-swizzled with node / 2 so that both access
    // patterns are free of bank conflicts - the staging writes (one column, 16 nodes per lane
    // quarter: with a plain row stride of 48 doubles every second node falls on the same bank,
    // measured 65 % of the LDS cycles lost) and the fragment reads (one node per quarter, 16
    // consecutive columns: the swizzle permutes them within their aligned block of 16).
    __host__ __device__ static constexpr int phys(int node, int col)
    {
        return node * RS + (col ^ (node >> 1));
    }
    // panel column of (mode m, component k, re / im)
    __host__ __device__ static constexpr int column(int m, int k, int ri)
    {
        const int a = (QM == 2) ? 2 * m + k / 3 : m;
        const int kk = (QM == 2) ? k % 3 : k;
        return 16 * (kk >> 1) + 4 * (2 * (kk & 1) + ri) + a;
    }
};

template <int NMT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_gather_cubic_mx(long n,
        const double *__restrict__ x, const double *__restrict__ y,
        const double *__restrict__ z, double rmax_gather,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        GatherGrids G, long rs,
        double *__restrict__ Ex, double *__restrict__ Ey, double *__restrict__ Ez,
        double *__restrict__ Bx, double *__restrict__ By, double *__restrict__ Bz,
        int chunks_per_wave, PushArgs PA)
{
    using C = GMX<NMT>;
    constexpr int NT = C::NT, FPL = C::FPL, RS = C::RS, NVL = C::NVL, WPAD = C::WPAD;
    typedef double double4_t __attribute__((ext_vector_type(4)));
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *panel = lds + (size_t)wave * C::WAVE_DOUBLES;      // G[node][column]
    double *Wp = panel + 16 * RS;                              // per-particle rows
    const int qa = lane >> 4, qj = lane & 15;
    // mode / field block of this lane's quarter (fold) ...
    const int q_m = (C::QM == 2) ? (qa >> 1) : qa;
    const int q_h = (C::QM == 2) ? (qa & 1) : 0;
    // ... and its staging role: node (jz, jr) = bits of the lane, field qa + 4 j of round j
    const int st_jr = lane & 3, st_jz = (lane >> 2) & 3;
    const cplx *st_ptr[NVL];
    int st_ofs[NVL];
    unsigned st_neg = 0u;         // bit j: the below-axis mirror of round j's field changes sign
#pragma unroll
    for (int j = 0; j < NVL; j++) {
        const int f = qa + 4 * j;
        const bool on = f < 6 * NMT;
        const int ff = on ? f : 0;
        const int m_ = ff / 6, k_ = ff - 6 * m_;
        st_ptr[j] = on ? G.g[ff] : nullptr;
        st_ofs[j] = C::phys(4 * st_jz + st_jr, C::column(m_, k_, 0));
        // mirror below the axis: -(-1)^m for r,t components, +(-1)^m for z
        // (inline_functions.py:70-79, 151-158)
        const bool neg = ((k_ % 3 == 2) ? m1pow(m_) : -m1pow(m_)) < 0.;
        st_neg |= neg ? (1u << j) : 0u;
    }
    // columns no field maps to (Nm = 3: the fourth quarter; Nm = 2: half of the second tile) stay 0
    for (int o = lane; o < 16 * RS; o += 64) panel[o] = 0.;

    const long chunk0 = (xcd_block_id() * nwaves + wave) * chunks_per_wave;
    double xn = 0., yn = 0., zn = 0.;
    if (chunk0 * 64 + lane < n) { xn = FB_NT_LD(x + (chunk0 * 64 + lane)); yn = FB_NT_LD(y + (chunk0 * 64 + lane)); zn = FB_NT_LD(z + (chunk0 * 64 + lane)); }
    RankPending pend = {-1, 0, 0, 0};
    for (int ch = 0; ch < chunks_per_wave; ch++) {
        const long base = (chunk0 + ch) * 64;
        if (base >= n) break;
        const long i = base + lane;
        const bool act = i < n;
        double cs = 1., sn = 0., Sz[4] = {0., 0., 0., 0.}, Sr[4] = {0., 0., 0., 0.};
        int kz = G_NOKEY, kr = G_NOKEY;
        bool inside = false;
        const double xj = xn, yj = yn;
        double zj = zn;
        if (PA.wzmax > PA.wzmin) {
            const double l_box = PA.wzmax - PA.wzmin;
            while (zj >= PA.wzmax) zj -= l_box;
            while (zj < PA.wzmin) zj += l_box;
        }
        if (ch + 1 < chunks_per_wave && i + 64 < n) { xn = FB_NT_LD(x + (i + 64)); yn = FB_NT_LD(y + (i + 64)); zn = FB_NT_LD(z + (i + 64)); }
        // momenta of the chunk: requested now, used by the push at the end (this kernel's
        // occupancy is set by its LDS panels, the 8 registers are free)
        double mom[4] = {0., 0., 0., 0.};
        if (PA.ux && act) { mom[0] = FB_NT_LD(PA.ux + i); mom[1] = FB_NT_LD(PA.uy + i); mom[2] = FB_NT_LD(PA.uz + i); mom[3] = FB_NT_LD(PA.ig + i); }
        double rj = 0., r_cell = 0., z_cell = 0.;
        if (act) {
            rj = sqrt(xj * xj + yj * yj);
            r_cell = invdr * (rj - rmin) - 0.5;
            z_cell = invdz * (zj - zmin) - 0.5;
            inside = rj < rmax_gather;
            if (inside) { kr = (int)floor(r_cell) - 1; kz = (int)floor(z_cell) - 1; }
        }
        const int pkz = __shfl_up(kz, 1), pkr = __shfl_up(kr, 1);
        const bool is_start = inside && (lane == 0 || kz != pkz || kr != pkr);
        unsigned long long rem = __ballot(is_start);
        // ---- staging of a segment: issue (node values in flight) / commit (to the LDS panel)
        double2 vals[NVL];
        bool st_below = false;
        int ps = 0, pe = 0;
        auto issue = [&]() {
            ps = __builtin_ctzll(rem);
            rem &= rem - 1ull;
            pe = rem ? __builtin_ctzll(rem) : 64;
            const int skz = __builtin_amdgcn_readlane(kz, ps), skr = __builtin_amdgcn_readlane(kr, ps);
            int row = skz + st_jz, col = skr + st_jr;
            if (row < 0) row += Nz; else if (row > Nz - 1) row -= Nz;
            st_below = col < 0;
            if (st_below) col = -col - 1; else if (col > Nr - 1) col = Nr - 1;
            const long off = (long)row * rs + col;
#pragma unroll
            for (int j = 0; j < NVL; j++) {
                vals[j] = make_double2(0., 0.);
                if ((64 * (j + 1) <= C::NV) || st_ptr[j]) vals[j] = ldc(st_ptr[j] + off);
            }
        };
        auto commit = [&]() {
#pragma unroll
            for (int j = 0; j < NVL; j++) {
                if ((64 * (j + 1) <= C::NV) || st_ptr[j]) {
                    const double sg_ = (st_below && ((st_neg >> j) & 1u)) ? -1. : 1.;
                    panel[st_ofs[j]] = sg_ * vals[j].x;
                    panel[st_ofs[j] ^ 4] = sg_ * vals[j].y;      // column + 4 (re -> im)
                }
            }
        };
        const bool any = rem != 0ull;
        if (any) issue();            // the first segment's L2 round trip overlaps the rest of phase 1
        if (act) {
            if (rj != 0.) { const double invr = 1. / rj; cs = xj * invr; sn = yj * invr; }
            if (inside) {
                // threading_methods.py:312-321
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
        }
        // per-particle rows: weights (0 for a particle that gathers nothing) and exptheta_m =
        // (cos - i sin)^m of modes 1 .. (recurrence of k_gather)
#pragma unroll
        for (int j = 0; j < 4; j++) { Wp[j * WPAD + lane] = Sz[j]; Wp[(4 + j) * WPAD + lane] = Sr[j]; }
        {
            double er = cs, ei = -sn;
#pragma unroll
            for (int m = 1; m < NMT; m++) {
                Wp[(8 + 2 * (m - 1)) * WPAD + lane] = er;
                Wp[(9 + 2 * (m - 1)) * WPAD + lane] = ei;
                const double nr_ = er * cs - ei * (-sn);
                const double ni_ = er * (-sn) + ei * cs;
                er = nr_; ei = ni_;
            }
        }
        double F[6] = {0., 0., 0., 0., 0., 0.};
        double4_t acc[NT];
        int gcur = -1;
        // finished group: mode factor x exptheta, sum over the quarters, keep the own particle
        auto fold = [&](int g) {
            const int p = 16 * g + qj;
            double er = 1., ei = 0.;
            if (q_m > 0) { er = Wp[(8 + 2 * (q_m - 1)) * WPAD + p]; ei = Wp[(9 + 2 * (q_m - 1)) * WPAD + p]; }
            const double factor = (q_m == 0) ? 1. : 2.;
            double Fq[FPL];
#pragma unroll
            for (int kk = 0; kk < FPL; kk++) {
                const double fr = acc[kk >> 1][2 * (kk & 1)], fi = acc[kk >> 1][2 * (kk & 1) + 1];
                double v = factor * (fr * er - fi * ei);
                if constexpr (C::QM == 1) v = sum_xor32(sum_xor16(v));
                else v = sum_xor32(v);                // the other mode of the same E / B block
                Fq[kk] = v;
            }
            if constexpr (C::QM == 1) {
                if (qa == g) {
#pragma unroll
                    for (int k = 0; k < 6; k++) F[k] = Fq[k];
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    const double other = other_xor16(Fq[kk]);       // the other block (E <-> B)
                    if (qa == g) { F[kk] = q_h ? other : Fq[kk]; F[3 + kk] = q_h ? Fq[kk] : other; }
                }
            }
        };
        bool more = any;
        while (more) {
            const int cps = ps, cpe = pe;
            commit();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the next segment's node values travel while this one is multiplied
            more = rem != 0ull;
            if (more) issue();
            for (int g = cps >> 4; g <= (cpe - 1) >> 4; g++) {
                if (g != gcur) {
                    if (gcur >= 0) fold(gcur);
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t] = (double4_t){0., 0., 0., 0.};
                    gcur = g;
                }
                // B operand: lane l -> W[node 4 jz + (l >> 4)][particle 16 g + (l & 15)], zero
                // outside the segment
                const int p = 16 * g + qj;
                const bool in = p >= cps && p < cpe;
                const double srv = in ? Wp[(4 + qa) * WPAD + p] : 0.;
#pragma unroll
                for (int jz = 0; jz < 4; jz++) {
                    const double b = Wp[jz * WPAD + p] * srv;
                    // A operand fragment: lane l -> G[node 4 jz + (l >> 4)][16 t + (l & 15)]; read
                    // per group instead of held across the segment: 18 registers less, which is
                    // what lets the next segment's loads be in flight (the LDS pipe is idle)
                    double ag[NT];
#pragma unroll
                    for (int t = 0; t < NT; t++) ag[t] = panel[C::phys(4 * jz + qa, 16 * t + qj)];
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[t], b, acc[t], 0, 0, 0);
                }
            }
        }
        if (gcur >= 0) fold(gcur);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (ranks not deferred to the next chunk as in k_gather: at 168 VGPRs the four registers of
        // the pending ranks spill - 5.2 -> 5.6 ms)
        gather_finish<false>(act, i, lane, xj, yj, zj, cs, sn, F, Ex, Ey, Ez, Bx, By, Bz, PA,
                             invdz, zmin, Nz, invdr, rmin, Nr, pend, mom);
    }
}

}  // namespace fb

using namespace fb;

extern "C" int fb_push_x(long n, double *x, double *y, double *z, const double *ux,
        const double *uy, const double *uz, const double *inv_gamma, double c, double dt,
        double px, double py, double pz, void *stream)
{
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double chdt = c * dt;
    bool v2 = (n % 2 == 0) && aligned16(x) && aligned16(y) && aligned16(z) && aligned16(ux) &&
              aligned16(uy) && aligned16(uz) && aligned16(inv_gamma);
    if (v2) {
        long nv = n / 2;
        hipLaunchKernelGGL(k_push_x<2>, dim3(stream_grid(nv)), dim3(256), 0, s, nv, x, y, z, ux,
                           uy, uz, inv_gamma, chdt, px, py, pz);
    } else {
        hipLaunchKernelGGL(k_push_x<1>, dim3(stream_grid(n)), dim3(256), 0, s, n, x, y, z, ux, uy,
                           uz, inv_gamma, chdt, px, py, pz);
    }
    FB_CHECK_LAUNCH("fb_push_x");
}

extern "C" int fb_push_p(long n, double *ux, double *uy, double *uz, double *inv_gamma,
        const double *Ex, const double *Ey, const double *Ez, const double *Bx,
        const double *By, const double *Bz, double q, double m, double c, double dt,
        void *stream)
{
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // fbpic/particles/push/numba_methods.py:41-42
    const double econst = q * dt / (m * c);
    const double bconst = 0.5 * q * dt / m;
    bool v2 = (n % 2 == 0) && aligned16(ux) && aligned16(uy) && aligned16(uz) &&
              aligned16(inv_gamma) && aligned16(Ex) && aligned16(Ey) && aligned16(Ez) &&
              aligned16(Bx) && aligned16(By) && aligned16(Bz);
    if (v2) {
        long nv = n / 2;
        hipLaunchKernelGGL(k_push_p<2>, dim3(stream_grid(nv)), dim3(256), 0, s, nv, ux, uy, uz,
                           inv_gamma, Ex, Ey, Ez, Bx, By, Bz, econst, bconst);
    } else {
        hipLaunchKernelGGL(k_push_p<1>, dim3(stream_grid(n)), dim3(256), 0, s, n, ux, uy, uz,
                           inv_gamma, Ex, Ey, Ez, Bx, By, Bz, econst, bconst);
    }
    FB_CHECK_LAUNCH("fb_push_p");
}

extern "C" int fb_shift_periodic(long n, double *z, double zmin, double zmax, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_shift_periodic, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream,
                       n, z, zmin, zmax);
    FB_CHECK_LAUNCH("fb_shift_periodic");
}

template <int NMT>
static int launch_gather_cubic_mx(long n, const double *x, const double *y, const double *z,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const GatherGrids &G, long row_stride, double *Ex, double *Ey, double *Ez, double *Bx,
        double *By, double *Bz, const PushArgs &PA, hipStream_t s, const char *where)
{
    using C = GMX<NMT>;
    // waves per workgroup so that as many waves as possible share the 160 KB of a CU
    const size_t wave_bytes = (size_t)C::WAVE_DOUBLES * 8;
    const int nwaves = lds_waves_per_workgroup(wave_bytes);
    const long nchunks = (n + 63) / 64;
    const long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    if (cpw < 1) cpw = 1;
    if (cpw > fb_cpw_cap(8)) cpw = fb_cpw_cap(8);
    const long total_waves = (nchunks + cpw - 1) / cpw;
    dim3 grid((unsigned)xcd_grid((total_waves + nwaves - 1) / nwaves)), block(64 * nwaves);
    hipLaunchKernelGGL((k_gather_cubic_mx<NMT>), grid, block, wave_bytes * nwaves, s, n, x, y, z,
                       rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, Ex, Ey, Ez,
                       Bx, By, Bz, cpw, PA);
    return check(hipGetLastError(), where);
}

static int launch_gather(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, double rmax_gather, double invdz, double zmin, int Nz, double invdr,
        double rmin, int Nr, const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        const PushArgs &PA, hipStream_t s, const char *where)
{
    if (n <= 0) return 0;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error(where, "Nm out of range"); return -1; }
    if (shape != FB_SHAPE_LINEAR && shape != FB_SHAPE_CUBIC) {
        set_error(where, "unknown shape");
        return -1;
    }
    GatherGrids G;
    for (int i = 0; i < 6 * Nm; i++) G.g[i] = (const cplx *)grids[i];
    for (int i = 6 * Nm; i < 6 * FB_MAX_MODES; i++) G.g[i] = nullptr;
    if (shape == FB_SHAPE_CUBIC && Nm >= 2 && Nm <= 4 && !getenv("FBPIC_AMD_GATHER_VALU") && !PA.range_mode) {
#define FB_MX(NMT) launch_gather_cubic_mx<NMT>(n, x, y, z, rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, G, \
                                               row_stride, Ex, Ey, Ez, Bx, By, Bz, PA, s, where)
        return Nm == 2 ? FB_MX(2) : (Nm == 3 ? FB_MX(3) : FB_MX(4));
#undef FB_MX
    }
    const int S = (shape == FB_SHAPE_LINEAR) ? 2 : 4;
    const size_t panel_bytes = (size_t)(2 * S * S * 6 * Nm + 2) * 8;
    // segments staged per round: up to 8, within ~16 KiB of LDS per wave
    int maxseg = (int)((16 * 1024) / panel_bytes);
    if (maxseg > 8) maxseg = 8;
    if (maxseg < 1) maxseg = 1;
    if (S * S * 6 * Nm <= 64) maxseg = 4;      // FAST path of k_gather: rounds of 4 segments
    const size_t wave_bytes = maxseg * panel_bytes;
    int nwaves = 4;
    while (nwaves > 1 && wave_bytes * nwaves > 64 * 1024) nwaves >>= 1;
    const long nchunks = (n + 63) / 64;
    // several rounds of waves (small tail), each long enough to pipeline its loads
    const long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    if (cpw < 1) cpw = 1;
    if (cpw > 64) cpw = 64;
    const long total_waves = (nchunks + cpw - 1) / cpw;
    dim3 grid((unsigned)xcd_grid((total_waves + nwaves - 1) / nwaves)), block(64 * nwaves);
#define FB_LAUNCH_GATHER(SH, NMT) \
    hipLaunchKernelGGL((k_gather<SH, NMT>), grid, block, wave_bytes * nwaves, s, Nm, n, x, y, z, \
                       rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, Ex, Ey, Ez, \
                       Bx, By, Bz, maxseg, cpw, PA)
    if (shape == FB_SHAPE_LINEAR) {
        if (Nm == 1) FB_LAUNCH_GATHER(FB_SHAPE_LINEAR, 1);
        else if (Nm == 2) FB_LAUNCH_GATHER(FB_SHAPE_LINEAR, 2);
        else if (Nm == 3) FB_LAUNCH_GATHER(FB_SHAPE_LINEAR, 3);
        else if (Nm == 4) FB_LAUNCH_GATHER(FB_SHAPE_LINEAR, 4);
        else FB_LAUNCH_GATHER(FB_SHAPE_LINEAR, 0);
    } else {
        if (Nm == 1) FB_LAUNCH_GATHER(FB_SHAPE_CUBIC, 1);
        else if (Nm == 2) FB_LAUNCH_GATHER(FB_SHAPE_CUBIC, 2);
        else if (Nm == 3) FB_LAUNCH_GATHER(FB_SHAPE_CUBIC, 3);
        else if (Nm == 4) FB_LAUNCH_GATHER(FB_SHAPE_CUBIC, 4);
        else FB_LAUNCH_GATHER(FB_SHAPE_CUBIC, 0);
    }
#undef FB_LAUNCH_GATHER
    return check(hipGetLastError(), where);
}

extern "C" int fb_gather(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, double rmax_gather, double invdz, double zmin, int Nz, double invdr,
        double rmin, int Nr, const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz, void *stream)
{
    PushArgs PA = {};
    return launch_gather(shape, Nm, n, x, y, z, rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr,
                         grids, row_stride, Ex, Ey, Ez, Bx, By, Bz, PA, (hipStream_t)stream,
                         "fb_gather");
}

static int gather_push_impl(const char *who, int shape, int Nm, long n, double *x, double *y, double *z,
        double *ux, double *uy, double *uz, double *inv_gamma,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        double q, double m, double c, double dt, double dt_x, double wrap_zmin, double wrap_zmax,
        const RankNext &RK, void *stream, const int *range_lo = nullptr, const int *range_hi = nullptr,
        int range_mode = 0)
{
    PushArgs PA;
    PA.RK = RK;
    PA.range_lo = range_lo; PA.range_hi = range_hi; PA.range_mode = range_mode;
    if (range_mode < 0 || range_mode > 2) { set_error(who, "range_mode must be 0, 1 or 2"); return -1; }
    PA.x = x; PA.y = y; PA.z = z;
    PA.ux = ux; PA.uy = uy; PA.uz = uz; PA.ig = inv_gamma;
    PA.econst = q * dt / (m * c);
    PA.bconst = 0.5 * q * dt / m;
    PA.chdt = c * dt_x;
    PA.wzmin = wrap_zmin; PA.wzmax = wrap_zmax;
    if (wrap_zmax > wrap_zmin && dt_x == 0.) {
        set_error(who, "the periodic wrap needs the position push (dt_x != 0)");
        return -1;
    }
    if (RK.count && dt_x == 0.) { set_error(who, "ranking needs the position push (dt_x != 0)"); return -1; }
    return launch_gather(shape, Nm, n, x, y, z, rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr,
                         grids, row_stride, Ex, Ey, Ez, Bx, By, Bz, PA, (hipStream_t)stream, who);
}

extern "C" int fb_gather_push(int shape, int Nm, long n, double *x, double *y, double *z,
        double *ux, double *uy, double *uz, double *inv_gamma,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        double q, double m, double c, double dt, double dt_x, double wrap_zmin, double wrap_zmax,
        void *stream)
{
    const RankNext none = {0., 0., 0., 0., nullptr, nullptr, nullptr};
    return gather_push_impl("fb_gather_push", shape, Nm, n, x, y, z, ux, uy, uz, inv_gamma, rmax_gather,
                            invdz, zmin, Nz, invdr, rmin, Nr, grids, row_stride, Ex, Ey, Ez, Bx, By, Bz,
                            q, m, c, dt, dt_x, wrap_zmin, wrap_zmax, none, stream);
}

extern "C" int fb_gather_push_rank_next_range(int shape, int Nm, long n, double *x, double *y, double *z,
        double *ux, double *uy, double *uz, double *inv_gamma,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        double q, double m, double c, double dt, double dt_x, double wrap_zmin, double wrap_zmax,
        double dt_push, double x_push, double y_push, double z_push, int ncell,
        void *sort_workspace, size_t workspace_bytes, int counts_are_zero,
        const int *range_lo, const int *range_hi, int range_mode, void *stream)
{
    const char *who = "fb_gather_push_rank_next";
    hipStream_t s = (hipStream_t)stream;
    if (ncell != Nz * (Nr + 1)) { set_error(who, "ncell != Nz*(Nr+1)"); return -1; }
    if (workspace_bytes < fb_bin_sort_workspace_bytes(n, ncell)) { set_error(who, "workspace too small"); return -1; }
    const BinSortWs W = carve_bin_sort_ws(sort_workspace, workspace_bytes, n, ncell);
    if (!counts_are_zero) {
        hipError_t e = hipMemsetAsync(W.count, 0, (size_t)ncell * sizeof(int), s);
        if (e != hipSuccess) return check(e, who);
    }
    if (n <= 0) return 0;
    // fbpic/particles/push/numba_methods.py:24-30: chdt = c * dt
    const RankNext RK = {c * dt_push, x_push, y_push, z_push, W.cell, W.rank, W.count};
    return gather_push_impl(who, shape, Nm, n, x, y, z, ux, uy, uz, inv_gamma, rmax_gather,
                            invdz, zmin, Nz, invdr, rmin, Nr, grids, row_stride, Ex, Ey, Ez, Bx, By, Bz,
                            q, m, c, dt, dt_x, wrap_zmin, wrap_zmax, RK, stream, range_lo, range_hi,
                            range_mode);
}

extern "C" int fb_gather_push_rank_next(int shape, int Nm, long n, double *x, double *y, double *z,
        double *ux, double *uy, double *uz, double *inv_gamma,
        double rmax_gather, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
        double q, double m, double c, double dt, double dt_x, double wrap_zmin, double wrap_zmax,
        double dt_push, double x_push, double y_push, double z_push, int ncell,
        void *sort_workspace, size_t workspace_bytes, int counts_are_zero, void *stream)
{
    return fb_gather_push_rank_next_range(shape, Nm, n, x, y, z, ux, uy, uz, inv_gamma, rmax_gather,
            invdz, zmin, Nz, invdr, rmin, Nr, grids, row_stride, Ex, Ey, Ez, Bx, By, Bz, q, m, c, dt,
            dt_x, wrap_zmin, wrap_zmax, dt_push, x_push, y_push, z_push, ncell, sort_workspace,
            workspace_bytes, counts_are_zero, nullptr, nullptr, 0, stream);
}
