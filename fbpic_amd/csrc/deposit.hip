// Charge / current deposition for gfx950.
//
// Design (MI355X-first; the reference's CUDA kernel is one thread per cell with serial
// loops and 4/16 atomics per thread, fbpic/particles/deposition/cuda_methods.py:84-194):
//
//   * particles are cell-sorted (sort.hip), so the macroparticles of one cell are
//     contiguous: each wave streams chunks of 64 consecutive particles with coalesced SoA
//     loads (lane = particle) and computes, per particle, the S x S shape factors for
//     mode 0 and modes >= 1 (Ruyten coefficient differs) and the NCOMP x NM complex mode
//     amplitudes.  These ~20 (linear) / ~44 (cubic) doubles per particle are staged in a
//     per-wave LDS panel (transposed + padded: conflict-free writes and reads);
//   * then the roles flip: lane o owns ONE output value o = (node jz,jr ; component ;
//     mode ; re/im) of the current cell and walks the 64 staged particles, accumulating
//     weight x amplitude in a REGISTER.  All particles of a cell hit the same S x S nodes,
//     so there is no atomic and no cross-lane reduction in the inner loop;
//   * when the cell of the next particle differs (wave-uniform test on the staged key),
//     the S*S*NCOMP*NM*2 registers are flushed with one global_atomic_add_f64 each, the
//     deposition guard cells of the reference (below-axis mirror, r clamp, periodic z:
//     fbpic/fields/numba_methods.py:409-461) being folded at that moment.  With ppc
//     particles per cell that is ~1/ppc of the atomics of a per-particle scatter, and they
//     are spread over distinct addresses.
//   * correctness never depends on the sort: an unsorted (or stale-sorted) stream only
//     flushes more often.
//
// Numerics: shape factors, Ruyten correction, axis flips and the mode recurrence restate
// fbpic/particles/deposition/particle_shapes.py:17-80 and threading_methods.py:27-650;
// each term is ((Sz*Sr)*flip)*amplitude as in the reference, only the summation order
// differs (as it does between the reference's own CPU and GPU paths): parity is
// 1e-13 * max|F| (tests/test_cpu_gpu_deposition.py:96).
#include "fb_common.h"
#include "dep_engine.h"
#include "cycle_dep.h"

namespace fb {

// NCOMP = 1 (rho) or 3 (Jr,Jt,Jz); this launch handles modes m0 .. m0+NM-1; Z0 <=> m0 == 0
template <int SHAPE, int NCOMP, int NM, bool Z0, bool RANK, bool PERM = false>
__global__ __launch_bounds__(256) void k_deposit(long n,
        const double *__restrict__ x, const double *__restrict__ y,
        const double *__restrict__ z, const double *__restrict__ w, double q,
        const double *__restrict__ ux, const double *__restrict__ uy,
        const double *__restrict__ uz, const double *__restrict__ inv_gamma, double c_light,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        DepGrids G, long rs, int m0,
        const double *__restrict__ beta0, const double *__restrict__ betah,
        int chunks_per_wave, unsigned long long *__restrict__ nflush, RankNext RK, PermArgs PM)
{
    static_assert(!RANK || NCOMP == 3, "ranking needs the momenta");
    static_assert(!PERM || NCOMP == 1, "the permuting front end belongs to the rho deposition");
    using E = DepEngine<SHAPE, NCOMP, NM, Z0>;
    using L = typename E::L;
    extern __shared__ double lds[];
    // wave index as a scalar: every loop bound below is then wave-uniform for the compiler
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    E eng;
    eng.init(lds + (size_t)wave * L::WAVE_DOUBLES, lane, G, rs, m0, Nz, Nr);
    const DepGeom geom = {invdz, zmin, Nz, invdr, rmin, Nr};

    const long chunk0 = (xcd_block_id() * nwaves + wave) * chunks_per_wave;
    // software pipeline: particle data of chunk ch+1 is requested before chunk ch is
    // processed, hiding the HBM latency behind the staging + accumulation work
    constexpr int NP = (NCOMP == 1 && !PERM) ? 4 : 8;
    double pn[NP];
    int idx_n = 0, idx_c = 0;        // PERM: source index of the next / current chunk's particle
    auto prefetch = [&](long ip) {
        if (ip < n) {
            if constexpr (PERM) {
#pragma unroll
                for (int k = 0; k < 8; k++) pn[k] = FB_NT_LDG(PM.src.p[k] + idx_n);
            } else {
                pn[0] = FB_NT_LD(x + ip); pn[1] = FB_NT_LD(y + ip); pn[2] = FB_NT_LD(z + ip); pn[3] = FB_NT_LD(w + ip);
                if constexpr (NCOMP == 3) { pn[4] = FB_NT_LD(ux + ip); pn[5] = FB_NT_LD(uy + ip); pn[6] = FB_NT_LD(uz + ip); pn[7] = FB_NT_LD(inv_gamma + ip); }
            }
        }
    };
    // PERM: two-stage pipeline - the index of chunk ch+2 is requested while the attributes
    // of chunk ch+1 (through the index requested one iteration earlier) are in flight
    if constexpr (PERM) { if (chunk0 * 64 + lane < n) idx_n = FB_NT_LD(PM.sidx + (chunk0 * 64 + lane)); }
    prefetch(chunk0 * 64 + lane);
    int idx_nn = 0;
    if constexpr (PERM) {
        idx_c = idx_n;
        if (chunks_per_wave > 1 && (chunk0 + 1) * 64 + lane < n) idx_nn = FB_NT_LD(PM.sidx + ((chunk0 + 1) * 64 + lane));
    }
    for (int ch = 0; ch < chunks_per_wave; ch++) {
        const long base = (chunk0 + ch) * 64;
        if (base >= n) break;
        const long ip = base + lane;
        const bool act = ip < n;
        int my_kz, my_kr, my_nb;
        int rk_c = -1, rk_run0 = 0, rk_base = 0;
        double pc[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) pc[k] = pn[k];
        if constexpr (PERM) {
            if (ch > 0) idx_c = idx_n;
            idx_n = idx_nn;
        }
        if (ch + 1 < chunks_per_wave) prefetch(ip + 64);
        if constexpr (PERM) {
            if (ch + 2 < chunks_per_wave && ip + 128 < n) idx_nn = FB_NT_LD(PM.sidx + (ip + 128));
        }
        double xj = pc[0], yj = pc[1], zj = pc[2];
        if constexpr (PERM) {
            if (act) {
                // pending push_x (numba_methods.py:25-30, expression of k_push_x), then every
                // attribute is written once, at its sorted slot
                const double g = pc[7];
                xj = pc[0] + PM.chdt * g * PM.px * pc[3];
                yj = pc[1] + PM.chdt * g * PM.py * pc[4];
                zj = pc[2] + PM.chdt * g * PM.pz * pc[5];
                FB_NT_ST(xj, PM.dst.p[0] + ip); FB_NT_ST(yj, PM.dst.p[1] + ip); FB_NT_ST(zj, PM.dst.p[2] + ip);
                FB_NT_ST(pc[3], PM.dst.p[3] + ip); FB_NT_ST(pc[4], PM.dst.p[4] + ip); FB_NT_ST(pc[5], PM.dst.p[5] + ip);
                FB_NT_ST(pc[6], PM.dst.p[6] + ip); FB_NT_ST(g, PM.dst.p[7] + ip);
                for (int k = 8; k < PM.nattr; k++) PM.dst.p[k][ip] = PM.src.p[k][idx_c];
                if (PM.cell_sorted) FB_NT_ST(FB_NT_LDG(PM.cell + idx_c), PM.cell_sorted + ip);
            }
        }
        const double wj = q * pc[PERM ? 6 : 3];
        if constexpr (NCOMP == 3)
            eng.stage(act, xj, yj, zj, wj, pc[4], pc[5], pc[6], pc[7], c_light, geom, beta0, betah,
                      my_kz, my_kr, my_nb);
        else
            eng.stage(act, xj, yj, zj, wj, 0., 0., 0., 0., 0., geom, beta0, betah, my_kz, my_kr, my_nb);
        if constexpr (RANK) {
            if (act) {
                // position after the coming push_x, cell as in k_cell_index / k_bin_rank
                const double g = pc[7];
                const double xq = xj + RK.chdt * g * RK.px * pc[4];
                const double yq = yj + RK.chdt * g * RK.py * pc[5];
                const double zq = zj + RK.chdt * g * RK.pz * pc[6];
                const double rq = sqrt(xq * xq + yq * yq);
                int ir_upper = (int)ceil(invdr * (rq - rmin) - 0.5);
                int iz_upper = (int)ceil(invdz * (zq - zmin) - 0.5);
                if (ir_upper > Nr) ir_upper = Nr;
                if (iz_upper < 0) iz_upper += Nz;
                else if (iz_upper > Nz - 1) iz_upper -= Nz;
                rk_c = ir_upper + iz_upper * (Nr + 1);
            }
            // one atomic per run of equal destination cells; its result is only needed at
            // the end of the chunk, so the round trip hides behind phase 2
            const int prev = __shfl_up(rk_c, 1);
            const bool rk_start = act && (lane == 0 || rk_c != prev);
            const unsigned long long rstarts = __ballot(rk_start);
            const int nact = __popcll(__ballot(act));
            const unsigned long long below = rstarts & ((2ull << lane) - 1ull);
            rk_run0 = 63 - __builtin_clzll(below | 1ull);
            if (rk_start) {
                const unsigned long long rest = (lane + 1 < 64) ? (rstarts >> (lane + 1)) : 0ull;
                const int len = rest ? (__builtin_ctzll(rest) + 1) : (nact - lane);
                rk_base = atomicAdd(RK.count + rk_c, len);
            }
        }
        wave_lds_release();
        eng.reduce((int)min((long)64, n - base), my_kz, my_kr, my_nb);
        if constexpr (RANK) {
            const int base_r = __shfl(rk_base, rk_run0);
            if (act) {
                RK.cell[ip] = rk_c;
                RK.rank[ip] = base_r + (lane - rk_run0);
            }
        }
        wave_lds_acquire();
    }
    eng.flush(false);
    // fragmentation statistic for the host's sort policy: 1024 counters (same-address
    // device atomics serialise at ~10 ns each; one shared counter would cost > 100 us)
    if (nflush && lane == 0)
        atomicAdd(nflush + ((blockIdx.x * nwaves + wave) & 1023), (unsigned long long)eng.my_flushes);
}

// push_x + counting sort + J AND rho deposition in one destination-ordered pass
// (fb_push_x_sort_deposit_J_rho): the particles are read through the inverse permutation as in
// k_deposit<.., PERM>; the current is deposited from the position BEFORE the pending push (where
// Simulation.step deposits J, main.py:515-517), the charge from the position after it
// (deposit('rho_next'), :528).  The two depositions run one after the other on the same LDS
// panel; the arithmetic of both overlaps the memory stalls of the permutation, and the
// stand-alone J pass (64 B per particle read again) disappears.  Modes 0 .. NM-1 (NM <= 4).
// Engines of the fused pass: J (3 components) and rho on the same per-wave panel.
// (Tried for the cubic shape with Nm >= 3: J as two engines, modes 0-1 and modes 2.., the panel
// shrinking from 33 to 24 rows = 13 KB per wave so that twelve instead of eight waves share a CU.
// The second engine stages the geometry again and the kernel needs 168 VGPRs with spills:
// 2048 x 512 x 64 ppc, Nm = 4: 5.70 ms unsplit, 6.78 ms split.)
template <int SHAPE, int NM> struct FusedPlan {
    using EJ = DepEngine<SHAPE, 3, NM, true>;
    using ER = DepEngine<SHAPE, 1, NM, true>;
    static constexpr int WAVE_DOUBLES = EJ::L::WAVE_DOUBLES > ER::L::WAVE_DOUBLES ? EJ::L::WAVE_DOUBLES
                                                                                   : ER::L::WAVE_DOUBLES;
};

template <int SHAPE, int NM>
__global__ __launch_bounds__(256) void k_perm_deposit_J_rho(long n, double q, double c_light,
        DepGeom gJ, DepGeom gR, DepGrids GJ, long rsJ, DepGrids GR, long rsR,
        const double *__restrict__ beta0, const double *__restrict__ betah,
        WaveRanges RG, PermArgs PM)
{
    using P = FusedPlan<SHAPE, NM>;
    using EJ = typename P::EJ;
    using ER = typename P::ER;
    constexpr int WAVE_DOUBLES = P::WAVE_DOUBLES;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    EJ ej;
    ER er;
    ej.init(lds + (size_t)wave * WAVE_DOUBLES, lane, GJ, rsJ, 0, gJ.Nz, gJ.Nr);
    er.init(lds + (size_t)wave * WAVE_DOUBLES, lane, GR, rsR, 0, gR.Nz, gR.Nr);

    long chunk0;
    int chunks_per_wave;
    if (!wave_range(RG, nwaves, wave, chunk0, chunks_per_wave)) return;
    double pn[8];
    int idx_n = 0, idx_c = 0, idx_nn = 0;
    auto prefetch = [&](long ip) {
        if (ip < n) {
#pragma unroll
            for (int k = 0; k < 8; k++) pn[k] = FB_NT_LDG(PM.src.p[k] + idx_n);
        }
    };
    if (chunk0 * 64 + lane < n) idx_n = FB_NT_LD(PM.sidx + (chunk0 * 64 + lane));
    prefetch(chunk0 * 64 + lane);
    idx_c = idx_n;
    if (chunks_per_wave > 1 && (chunk0 + 1) * 64 + lane < n) idx_nn = FB_NT_LD(PM.sidx + ((chunk0 + 1) * 64 + lane));
    for (int ch = 0; ch < chunks_per_wave; ch++) {
        const long base = (chunk0 + ch) * 64;
        if (base >= n) break;
        const long ip = base + lane;
        const bool act = ip < n;
        const int cnt = (int)min((long)64, n - base);
        double pc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) pc[k] = pn[k];
        if (ch > 0) idx_c = idx_n;
        idx_n = idx_nn;
        if (ch + 1 < chunks_per_wave) prefetch(ip + 64);
        if (ch + 2 < chunks_per_wave && ip + 128 < n) idx_nn = FB_NT_LD(PM.sidx + (ip + 128));
        const double wj = q * pc[6];
        int kz, kr, nb;
        // ---- J from the position before the push
        ej.stage(act, pc[0], pc[1], pc[2], wj, pc[3], pc[4], pc[5], pc[7], c_light, gJ, beta0, betah,
                 kz, kr, nb);
        wave_lds_release();
        ej.reduce(cnt, kz, kr, nb);
        wave_lds_acquire();
        // ---- pending push_x (expression of k_push_x), attributes written at the sorted slot
        double xj = pc[0], yj = pc[1], zj = pc[2];
        if (act) {
            const double g = pc[7];
            xj = pc[0] + PM.chdt * g * PM.px * pc[3];
            yj = pc[1] + PM.chdt * g * PM.py * pc[4];
            zj = pc[2] + PM.chdt * g * PM.pz * pc[5];
            FB_NT_ST(xj, PM.dst.p[0] + ip); FB_NT_ST(yj, PM.dst.p[1] + ip); FB_NT_ST(zj, PM.dst.p[2] + ip);
            FB_NT_ST(pc[3], PM.dst.p[3] + ip); FB_NT_ST(pc[4], PM.dst.p[4] + ip); FB_NT_ST(pc[5], PM.dst.p[5] + ip);
            FB_NT_ST(pc[6], PM.dst.p[6] + ip); FB_NT_ST(g, PM.dst.p[7] + ip);
            for (int k = 8; k < PM.nattr; k++) PM.dst.p[k][ip] = PM.src.p[k][idx_c];
            if (PM.cell_sorted) FB_NT_ST(FB_NT_LDG(PM.cell + idx_c), PM.cell_sorted + ip);
        }
        // ---- rho from the pushed position
        er.stage(act, xj, yj, zj, wj, 0., 0., 0., 0., 0., gR, beta0, betah, kz, kr, nb);
        wave_lds_release();
        er.reduce(cnt, kz, kr, nb);
        wave_lds_acquire();
    }
    ej.flush(false);
    er.flush(false);
}

// Round 5: the same pass with the merged engine of the one-pass cycle (cycle_dep.h) - J and rho staged
// together, ONE traversal of the runs.  The stream is in destination order, i.e. sorted by the cell of
// the pushed position: the runs are those of the rho keys (exact), and a particle whose position
// BEFORE the push has another stencil (it crosses a cell boundary within this half push) is a stray of
// the J engine only, written out directly.  Linear shape, both depositions on the same grid geometry,
// targets addressed from one base with the same strides (the in-step records); anything else takes
// k_perm_deposit_J_rho.  16 ppc (C3 / C4) halves the run length of C2 and doubles the flushes per
// particle: that is where one traversal instead of two pays most.
#ifndef FB_PERM_SPLIT_AT
#define FB_PERM_SPLIT_AT 6         // strays of the J engine per 64 particles from which a chunk is traversed twice
#endif
template <int NM>
__global__ __launch_bounds__(256) void k_perm_deposit_J_rho_merged(long n, double q, double c_light,
        DepGeom g, DepGrids GJ, DepGrids GR, long rs, cplx *gbase,
        const double *__restrict__ beta0, const double *__restrict__ betah,
        WaveRanges RG, PermArgs PM)
{
    using ED = CycleDep<NM>;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ED ed;
    ed.init(lds + (size_t)wave * ED::L::WAVE_DOUBLES, lane, GJ, rs, GR, rs, g.Nz, g.Nr, gbase);

    long chunk0;
    int chunks_per_wave;
    if (!wave_range(RG, nwaves, wave, chunk0, chunks_per_wave)) return;
    double pn[8];
    // (lanes beyond the last particle stage a harmless particle of weight 0)
    pn[0] = 1. / g.invdr; pn[1] = 0.; pn[2] = g.zmin + 1. / g.invdz; pn[3] = 0.; pn[4] = 0.; pn[5] = 0.; pn[6] = 0.; pn[7] = 1.;
    int idx_n = 0, idx_c = 0, idx_nn = 0;
    auto prefetch = [&](long ip) {
        if (ip < n) {
#pragma unroll
            for (int k = 0; k < 8; k++) pn[k] = FB_NT_LDG(PM.src.p[k] + idx_n);
        }
    };
    if (chunk0 * 64 + lane < n) idx_n = FB_NT_LD(PM.sidx + (chunk0 * 64 + lane));
    prefetch(chunk0 * 64 + lane);
    idx_c = idx_n;
    if (chunks_per_wave > 1 && (chunk0 + 1) * 64 + lane < n) idx_nn = FB_NT_LD(PM.sidx + ((chunk0 + 1) * 64 + lane));
    for (int ch = 0; ch < chunks_per_wave; ch++) {
        const long base = (chunk0 + ch) * 64;
        if (base >= n) break;
        const long ip = base + lane;
        const bool act = ip < n;
        const int cnt = (int)min((long)64, n - base);
        double pc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) pc[k] = pn[k];
        if (ch > 0) idx_c = idx_n;
        idx_n = idx_nn;
        if (ch + 1 < chunks_per_wave) prefetch(ip + 64);
        if (ch + 2 < chunks_per_wave && ip + 128 < n) idx_nn = FB_NT_LD(PM.sidx + (ip + 128));
        const double wj = act ? q * pc[6] : 0.;
        // pending push_x (expression of k_push_x), attributes written at the sorted slot
        const double gi = pc[7];
        const double xj = pc[0] + PM.chdt * gi * PM.px * pc[3];
        const double yj = pc[1] + PM.chdt * gi * PM.py * pc[4];
        const double zj = pc[2] + PM.chdt * gi * PM.pz * pc[5];
        // Ruyten coefficients of both positions (DepEngine::ruyten_index)
        const double r0 = sqrt(pc[0] * pc[0] + pc[1] * pc[1]), r1 = sqrt(xj * xj + yj * yj);
        const int irJ = min((int)ceil(g.invdr * (r0 - g.rmin) - 0.5), g.Nr);
        const int irR = min((int)ceil(g.invdr * (r1 - g.rmin) - 0.5), g.Nr);
        const double bJ0 = beta0[irJ], bJh = betah[irJ], bR0 = beta0[irR], bRh = betah[irR];
        if (act) {
            FB_NT_ST(xj, PM.dst.p[0] + ip); FB_NT_ST(yj, PM.dst.p[1] + ip); FB_NT_ST(zj, PM.dst.p[2] + ip);
            FB_NT_ST(pc[3], PM.dst.p[3] + ip); FB_NT_ST(pc[4], PM.dst.p[4] + ip); FB_NT_ST(pc[5], PM.dst.p[5] + ip);
            FB_NT_ST(pc[6], PM.dst.p[6] + ip); FB_NT_ST(gi, PM.dst.p[7] + ip);
            // (static indices: a run-time index into the by-value pointer tables sends them to scratch)
#pragma unroll
            for (int k = 8; k < 16; k++)
                if (k < PM.nattr) PM.dst.p[k][ip] = PM.src.p[k][idx_c];
            if (PM.cell_sorted) FB_NT_ST(FB_NT_LDG(PM.cell + idx_c), PM.cell_sorted + ip);
        }
        int jkz, jkr, jnb, rkz, rkr, rnb;
        ed.template stage<0>(pc[0], pc[1], pc[2], wj, pc[3], pc[4], pc[5], gi, c_light, g, bJ0, bJh, jkz, jkr, jnb);
        ed.template stage<1>(xj, yj, zj, wj, 0., 0., 0., 0., 0., g, bR0, bRh, rkz, rkr, rnb);
        const int pkz = __shfl_up(rkz, 1), pkr = __shfl_up(rkr, 1);
        const unsigned long long runstarts = __ballot(act && (lane == 0 || rkz != pkz || rkr != pkr));
        const bool homeJ = act && jkz == rkz && jkr == rkr && jnb == rnb;
        const unsigned long long smJ = __ballot(act && !homeJ);
        wave_lds_release();
        if (__popcll(smJ) > FB_PERM_SPLIT_AT) {
            // (C3: a quarter of the window's electrons are in the wake; scattering each of them cost
            // 0.95 ms per launch where the two engines of k_perm_deposit_J_rho take 0.70)
            ed.reduce_split(cnt, act, jkz, jkr, jnb, rkz, rkr, rnb);
        } else {
            if (smJ) {
                ed.scatter_strays(smJ, 0ull, jkz, jkr, jnb, rkz, rkr, rnb);
                ed.zero_amplitudes(!homeJ, false);
                wave_lds_release();
            }
            ed.reduce(cnt, runstarts, __ballot(act), rkz, rkr, rnb);
        }
        wave_lds_acquire();
    }
    ed.flush(false);
}

static int dep_waves_per_workgroup(size_t wave_bytes) { return lds_waves_per_workgroup(wave_bytes); }

template <int SHAPE, int NCOMP, int NM, bool Z0, bool RANK, bool PERM = false>
static int launch_z(long n, const double *x, const double *y, const double *z, const double *w,
        double q, const double *ux, const double *uy, const double *uz, const double *ig,
        double c, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const DepGrids &G, long rs, int m0, const double *b0, const double *bh,
        unsigned long long *nflush, const RankNext &RK, hipStream_t s,
        const PermArgs *PMp = nullptr)
{
    using L = DepLayout<SHAPE, NCOMP, NM, Z0>;
    PermArgs PM = {};
    if (PMp) PM = *PMp;
    // waves per workgroup: keep the LDS panel <= 64 KiB (1, 2 or 4 waves per workgroup and 32 /
    // 64 / 128 waves per CU in flight measured: 4 and 64 as good as any)
    const int nwaves = dep_waves_per_workgroup(L::wave_bytes());
    const long nchunks = (n + 63) / 64;
    // ~8 waves per SIMD-quad in flight over 256 CUs, each walking consecutive chunks so
    // that a cell straddling two chunks is not flushed twice
    long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    if (cpw < 1) cpw = 1;
    if (cpw > 64) cpw = 64;
    const long total_waves = (nchunks + cpw - 1) / cpw;
    const long nblocks = xcd_grid((total_waves + nwaves - 1) / nwaves);
    auto kern = k_deposit<SHAPE, NCOMP, NM, Z0, RANK, PERM>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * nwaves),
                       L::wave_bytes() * nwaves, s, n, x, y, z, w, q, ux, uy, uz, ig, c,
                       invdz, zmin, Nz, invdr, rmin, Nr, G, rs, m0, b0, bh, cpw,
                       (m0 == 0) ? nflush : nullptr, RK, PM);
    return check(hipGetLastError(), "fb_deposit");
}

template <int SHAPE, int NCOMP, int NM>
static int launch_one(long n, const double *x, const double *y, const double *z, const double *w,
        double q, const double *ux, const double *uy, const double *uz, const double *ig,
        double c, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const DepGrids &G, long rs, int m0, const double *b0, const double *bh,
        unsigned long long *nflush, const RankNext *RK, hipStream_t s, const PermArgs *PM = nullptr)
{
    const RankNext none = {0., 0., 0., 0., nullptr, nullptr, nullptr};
#define ZARGS n, x, y, z, w, q, ux, uy, uz, ig, c, invdz, zmin, Nz, invdr, rmin, Nr, G, rs, m0, b0, bh, nflush
    if (m0 == 0) {
        if constexpr (NCOMP == 3) {
            if (RK) return launch_z<SHAPE, NCOMP, NM, true, true>(ZARGS, *RK, s);
        }
        if constexpr (NCOMP == 1) {
            if (PM) return launch_z<SHAPE, NCOMP, NM, true, false, true>(ZARGS, none, s, PM);
        }
        return launch_z<SHAPE, NCOMP, NM, true, false>(ZARGS, none, s);
    }
    return launch_z<SHAPE, NCOMP, NM, false, false>(ZARGS, none, s);
#undef ZARGS
}

template <int SHAPE, int NCOMP>
static int launch_modes(int Nm, long n, const double *x, const double *y, const double *z,
        const double *w, double q, const double *ux, const double *uy, const double *uz,
        const double *ig, double c, double invdz, double zmin, int Nz, double invdr, double rmin,
        int Nr, const DepGrids &G, long rs, const double *b0, const double *bh,
        unsigned long long *nflush, const RankNext *RK, hipStream_t s, const PermArgs *PM = nullptr)
{
    int m0 = 0;
    while (m0 < Nm) {
        int left = Nm - m0, r;
#define ARGS n, x, y, z, w, q, ux, uy, uz, ig, c, invdz, zmin, Nz, invdr, rmin, Nr, G, rs, m0, b0, bh, nflush, RK, s, PM
        // (splitting a cubic 4-mode J launch into 2 + 2 or 1 + 1 + 1 + 1 modes for more waves per
        // SIMD changes nothing: 3.38 / 3.44 / 3.64 ms at 2048 x 512, 16 ppc)
        if (left >= 4) { r = launch_one<SHAPE, NCOMP, 4>(ARGS); m0 += 4; }
        else if (left == 3) { r = launch_one<SHAPE, NCOMP, 3>(ARGS); m0 += 3; }
        else if (left == 2) { r = launch_one<SHAPE, NCOMP, 2>(ARGS); m0 += 2; }
        else { r = launch_one<SHAPE, NCOMP, 1>(ARGS); m0 += 1; }
#undef ARGS
        if (r) return r;
        if (PM) {
            // modes beyond the first launch (Nm > 4) read the sorted arrays it has written
            x = PM->dst.p[0]; y = PM->dst.p[1]; z = PM->dst.p[2]; w = PM->dst.p[6];
            PM = nullptr;
        }
    }
    return 0;
}

}  // namespace fb

using namespace fb;

extern "C" int fb_deposit_rho(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, const double *w, double q, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, void *const *rho, long row_stride, long col_stride,
        const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh,
        unsigned long long *nflush, void *stream)
{
    (void)prefix_sum;
    if (n <= 0) return 0;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error("fb_deposit_rho", "Nm out of range"); return -1; }
    DepGrids G;
    G.cs = col_stride > 0 ? col_stride : 1;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) G.g[i] = i < Nm ? (cplx *)rho[i] : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (shape == FB_SHAPE_LINEAR)
        return launch_modes<FB_SHAPE_LINEAR, 1>(Nm, n, x, y, z, w, q, nullptr, nullptr, nullptr,
                nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0,
                ruyten_mh, nflush, nullptr, s);
    if (shape == FB_SHAPE_CUBIC)
        return launch_modes<FB_SHAPE_CUBIC, 1>(Nm, n, x, y, z, w, q, nullptr, nullptr, nullptr,
                nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0,
                ruyten_mh, nflush, nullptr, s);
    set_error("fb_deposit_rho", "unknown shape");
    return -1;
}

static int deposit_J_impl(const char *who, int shape, int Nm, long n, const double *x,
        const double *y, const double *z, const double *w, double q, const double *ux,
        const double *uy, const double *uz, const double *inv_gamma, double c, double invdz,
        double zmin, int Nz, double invdr, double rmin, int Nr, void *const *J, long row_stride,
        long col_stride, const double *ruyten_m0, const double *ruyten_mh,
        unsigned long long *nflush, const RankNext *RK, hipStream_t s)
{
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error(who, "Nm out of range"); return -1; }
    DepGrids G;
    G.cs = col_stride > 0 ? col_stride : 1;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) G.g[i] = i < 3 * Nm ? (cplx *)J[i] : nullptr;
    if (shape == FB_SHAPE_LINEAR)
        return launch_modes<FB_SHAPE_LINEAR, 3>(Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c,
                invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0, ruyten_mh, nflush, RK, s);
    if (shape == FB_SHAPE_CUBIC)
        return launch_modes<FB_SHAPE_CUBIC, 3>(Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c,
                invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0, ruyten_mh, nflush, RK, s);
    set_error(who, "unknown shape");
    return -1;
}

extern "C" int fb_deposit_J(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, const double *w, double q, const double *ux, const double *uy,
        const double *uz, const double *inv_gamma, double c, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, void *const *J, long row_stride, long col_stride,
        const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh,
        unsigned long long *nflush, void *stream)
{
    (void)prefix_sum;
    if (n <= 0) return 0;
    return deposit_J_impl("fb_deposit_J", shape, Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c,
                          invdz, zmin, Nz, invdr, rmin, Nr, J, row_stride, col_stride, ruyten_m0, ruyten_mh,
                          nflush, nullptr, (hipStream_t)stream);
}

extern "C" int fb_deposit_J_rank_next(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, const double *w, double q, const double *ux, const double *uy,
        const double *uz, const double *inv_gamma, double c, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, void *const *J, long row_stride, long col_stride,
        const double *ruyten_m0, const double *ruyten_mh, unsigned long long *nflush,
        double dt_push, double x_push, double y_push, double z_push, int ncell,
        void *sort_workspace, size_t workspace_bytes, int counts_are_zero, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (ncell != Nz * (Nr + 1)) { set_error("fb_deposit_J_rank_next", "ncell != Nz*(Nr+1)"); return -1; }
    if (workspace_bytes < fb_bin_sort_workspace_bytes(n, ncell)) {
        set_error("fb_deposit_J_rank_next", "workspace too small");
        return -1;
    }
    const BinSortWs W = carve_bin_sort_ws(sort_workspace, workspace_bytes, n, ncell);
    if (!counts_are_zero) {
        hipError_t e = hipMemsetAsync(W.count, 0, (size_t)ncell * sizeof(int), s);
        if (e != hipSuccess) return check(e, "fb_deposit_J_rank_next(memset)");
    }
    if (n <= 0) return 0;
    // fbpic/particles/push/numba_methods.py:24-30: chdt = c * dt
    const RankNext RK = {c * dt_push, x_push, y_push, z_push, W.cell, W.rank, W.count};
    return deposit_J_impl("fb_deposit_J_rank_next", shape, Nm, n, x, y, z, w, q, ux, uy, uz,
                          inv_gamma, c, invdz, zmin, Nz, invdr, rmin, Nr, J, row_stride, col_stride,
                          ruyten_m0, ruyten_mh, nflush, &RK, s);
}

extern "C" int fb_push_x_sort_deposit_rho(long n, int ncell, const double *x, const double *y,
        const double *z, const double *ux, const double *uy, const double *uz,
        const double *inv_gamma, double c, double dt, double x_push, double y_push, double z_push,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        int nattr, const double *const *src, double *const *dst,
        int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
        void *workspace, size_t workspace_bytes, int preranked,
        int shape, int Nm, double q, void *const *rho, long row_stride, long col_stride,
        const double *ruyten_m0, const double *ruyten_mh, void *stream)
{
    const char *who = "fb_push_x_sort_deposit_rho";
    hipStream_t s = (hipStream_t)stream;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error(who, "Nm out of range"); return -1; }
    if (shape != FB_SHAPE_LINEAR && shape != FB_SHAPE_CUBIC) { set_error(who, "unknown shape"); return -1; }
    if (!sorted_idx) { set_error(who, "sorted_idx (n ints) is required: it holds the permutation"); return -1; }
    if (nattr < 8) { set_error(who, "src / dst must hold x, y, z, ux, uy, uz, w, inv_gamma"); return -1; }
    // fbpic/particles/push/numba_methods.py:24-30: chdt = c * dt
    const PushX P = {ux, uy, uz, inv_gamma, c * dt, x_push, y_push, z_push};
    BinSortWs W;
    int r = bin_sort_prepare(who, true, preranked != 0, P, n, ncell, x, y, z, invdz, zmin, Nz, invdr,
                             rmin, Nr, nattr, src, prefix_sum, workspace, workspace_bytes, &W, s);
    if (r) return r;
    r = bin_sort_build_sidx(who, n, ncell, W, prefix_sum, sorted_idx, s);
    if (r || n <= 0) return r;
    PermArgs PM;
    PM.sidx = sorted_idx;
    for (int k = 0; k < 16; k++) { PM.src.p[k] = k < nattr ? src[k] : nullptr; PM.dst.p[k] = k < nattr ? dst[k] : nullptr; }
    PM.nattr = nattr;
    PM.chdt = P.chdt; PM.px = x_push; PM.py = y_push; PM.pz = z_push;
    PM.cell = W.cell;
    PM.cell_sorted = cell_idx_sorted;
    DepGrids G;
    G.cs = col_stride > 0 ? col_stride : 1;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) G.g[i] = i < Nm ? (cplx *)rho[i] : nullptr;
    if (shape == FB_SHAPE_LINEAR)
        return launch_modes<FB_SHAPE_LINEAR, 1>(Nm, n, x, y, z, src[6], q, nullptr, nullptr, nullptr,
                nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0,
                ruyten_mh, nullptr, nullptr, s, &PM);
    return launch_modes<FB_SHAPE_CUBIC, 1>(Nm, n, x, y, z, src[6], q, nullptr, nullptr, nullptr,
            nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0,
            ruyten_mh, nullptr, nullptr, s, &PM);
}

template <int SHAPE, int NM>
static int launch_perm_J_rho(long n, double q, double c, const DepGeom &gJ, const DepGeom &gR,
                             const DepGrids &GJ, long rsJ, const DepGrids &GR, long rsR,
                             const double *b0, const double *bh, const PermArgs &PM, hipStream_t s)
{
    const size_t wave_bytes = 8 * (size_t)FusedPlan<SHAPE, NM>::WAVE_DOUBLES;
    // one wave per workgroup, as for the one-pass kernel (cycle.hip): 0.208 against 0.218 ms per
    // launch at C2 for the 4-wave workgroups dep_waves_per_workgroup() picks, 5.69 against 5.74 ms
    // for the cubic pass of C5 (the stand-alone depositions measure the other way: 80 against 77 us)
    const int nwaves = 1;
    const long nchunks = (n + 63) / 64;
    const long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    if (cpw < 1) cpw = 1;
    if (cpw > fb_cpw_cap(16)) cpw = fb_cpw_cap(16);
    // graded ranges for the linear shape (C3: 0.75 -> 0.69 ms per launch); the cubic pass keeps the plain cut - its waves pay two
    // cubic engine set-ups, and short ranges at the end measured 5.41 -> 5.56 ms at C5 (profiles/r06_graded_ranges.txt)
    long nblocks;
    const WaveRanges RG = graded_ranges(nchunks, cpw, nwaves, &nblocks, SHAPE == FB_SHAPE_LINEAR ? 8 : 0);
    hipLaunchKernelGGL((k_perm_deposit_J_rho<SHAPE, NM>), dim3((unsigned)nblocks), dim3(64 * nwaves),
                       wave_bytes * nwaves, s, n, q, c, gJ, gR, GJ, rsJ, GR, rsR, b0, bh, RG, PM);
    return check(hipGetLastError(), "fb_push_x_sort_deposit_J_rho");
}

template <int NM>
static int launch_perm_J_rho_merged(long n, double q, double c, const DepGeom &g, const DepGrids &GJ,
                                    const DepGrids &GR, long rs, cplx *gbase, const double *b0,
                                    const double *bh, const PermArgs &PM, hipStream_t s)
{
    const size_t wave_bytes = 8 * (size_t)CycleDep<NM>::L::WAVE_DOUBLES;
    const int nwaves = 1;          // (as launch_perm_J_rho)
    const long nchunks = (n + 63) / 64;
    const long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    if (cpw < 1) cpw = 1;
    if (cpw > fb_cpw_cap(16)) cpw = fb_cpw_cap(16);
    long nblocks;
    const WaveRanges RG = graded_ranges(nchunks, cpw, nwaves, &nblocks);
    hipLaunchKernelGGL((k_perm_deposit_J_rho_merged<NM>), dim3((unsigned)nblocks), dim3(64 * nwaves),
                       wave_bytes * nwaves, s, n, q, c, g, GJ, GR, rs, gbase, b0, bh, RG, PM);
    return check(hipGetLastError(), "fb_push_x_sort_deposit_J_rho");
}

extern "C" int fb_push_x_sort_deposit_J_rho(long n, int ncell, const double *x, const double *y,
        const double *z, const double *ux, const double *uy, const double *uz,
        const double *inv_gamma, double c, double dt, double x_push, double y_push, double z_push,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        int nattr, const double *const *src, double *const *dst,
        int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
        void *workspace, size_t workspace_bytes, int preranked,
        int shape, int Nm, double q, double zmin_J, void *const *J, long J_row_stride,
        long J_col_stride, void *const *rho, long row_stride, long col_stride,
        const double *ruyten_m0, const double *ruyten_mh, int engine, void *stream)
{
    const char *who = "fb_push_x_sort_deposit_J_rho";
    hipStream_t s = (hipStream_t)stream;
    if (Nm < 1 || Nm > 4) { set_error(who, "Nm must be 1..4 (use the separate entry points beyond)"); return -1; }
    if (shape != FB_SHAPE_LINEAR && shape != FB_SHAPE_CUBIC) { set_error(who, "unknown shape"); return -1; }
    if (!sorted_idx) { set_error(who, "sorted_idx (n ints) is required: it holds the permutation"); return -1; }
    if (nattr < 8) { set_error(who, "src / dst must hold x, y, z, ux, uy, uz, w, inv_gamma"); return -1; }
    const PushX P = {ux, uy, uz, inv_gamma, c * dt, x_push, y_push, z_push};
    BinSortWs W;
    int r = bin_sort_prepare(who, true, preranked != 0, P, n, ncell, x, y, z, invdz, zmin, Nz, invdr,
                             rmin, Nr, nattr, src, prefix_sum, workspace, workspace_bytes, &W, s);
    if (r) return r;
    r = bin_sort_build_sidx(who, n, ncell, W, prefix_sum, sorted_idx, s);
    if (r || n <= 0) return r;
    PermArgs PM;
    PM.sidx = sorted_idx;
    for (int k = 0; k < 16; k++) { PM.src.p[k] = k < nattr ? src[k] : nullptr; PM.dst.p[k] = k < nattr ? dst[k] : nullptr; }
    PM.nattr = nattr;
    PM.chdt = P.chdt; PM.px = x_push; PM.py = y_push; PM.pz = z_push;
    PM.cell = W.cell;
    PM.cell_sorted = cell_idx_sorted;
    DepGrids GJ, GR;
    GJ.cs = J_col_stride > 0 ? J_col_stride : 1;
    GR.cs = col_stride > 0 ? col_stride : 1;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) {
        GJ.g[i] = i < 3 * Nm ? (cplx *)J[i] : nullptr;
        GR.g[i] = i < Nm ? (cplx *)rho[i] : nullptr;
    }
    const DepGeom gJ = {invdz, zmin_J, Nz, invdr, rmin, Nr}, gR = {invdz, zmin, Nz, invdr, rmin, Nr};
#ifndef FB_PERM_TWO_ENGINES
    // engine: 0 / 2 = the merged engine where it applies, 1 = the two engines one after the other (the
    // faster form where many particles change cell within the push - a laser wake: C3 0.70 against
    // 0.77-0.82 ms per launch, while a thermal plasma at 32 ppc takes 0.208 against 0.231 merged)
    if (engine != 1 && shape == FB_SHAPE_LINEAR && zmin_J == zmin && J_row_stride == row_stride && GJ.cs == GR.cs) {
        // one base for both targets (views of one record array, fields of one slab): merged engine
        cplx *bj = dep_grids_base(GJ, 3 * Nm, J_row_stride, Nz), *br = dep_grids_base(GR, Nm, row_stride, Nz);
        if (bj && br) {
            const uintptr_t lo = (uintptr_t)bj < (uintptr_t)br ? (uintptr_t)bj : (uintptr_t)br;
            uintptr_t hi = 0;
            for (int i = 0; i < 3 * Nm; i++) if ((uintptr_t)GJ.g[i] > hi) hi = (uintptr_t)GJ.g[i];
            for (int i = 0; i < Nm; i++) if ((uintptr_t)GR.g[i] > hi) hi = (uintptr_t)GR.g[i];
            if ((double)(hi - lo) + 16. * (double)row_stride * (double)(Nz + 1) < 4294967296.) {
#define LPM(NM_) launch_perm_J_rho_merged<NM_>(n, q, c, gR, GJ, GR, row_stride, (cplx *)lo, ruyten_m0, ruyten_mh, PM, s)
                return Nm == 1 ? LPM(1) : Nm == 2 ? LPM(2) : Nm == 3 ? LPM(3) : LPM(4);
#undef LPM
            }
        }
    }
#endif
#define LPJR(SH, NM_) launch_perm_J_rho<SH, NM_>(n, q, c, gJ, gR, GJ, J_row_stride, GR, row_stride, \
                                                 ruyten_m0, ruyten_mh, PM, s)
    if (shape == FB_SHAPE_LINEAR)
        return Nm == 1 ? LPJR(FB_SHAPE_LINEAR, 1) : Nm == 2 ? LPJR(FB_SHAPE_LINEAR, 2)
             : Nm == 3 ? LPJR(FB_SHAPE_LINEAR, 3) : LPJR(FB_SHAPE_LINEAR, 4);
    return Nm == 1 ? LPJR(FB_SHAPE_CUBIC, 1) : Nm == 2 ? LPJR(FB_SHAPE_CUBIC, 2)
         : Nm == 3 ? LPJR(FB_SHAPE_CUBIC, 3) : LPJR(FB_SHAPE_CUBIC, 4);
#undef LPJR
}
