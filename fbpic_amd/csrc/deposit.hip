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

namespace fb {

struct DepGrids { cplx *g[3 * FB_MAX_MODES]; };   // [comp + NCOMP*m]

template <int SHAPE> struct ShapeTraits;
template <> struct ShapeTraits<FB_SHAPE_LINEAR> { static constexpr int S = 2, H = 1; };
template <> struct ShapeTraits<FB_SHAPE_CUBIC> { static constexpr int S = 4, H = 2; };

// Longitudinal shape factors, particle_shapes.py:17-22, 44-58
template <int SHAPE>
__device__ __forceinline__ void shape_z(double z_cell, double *Sz)
{
    if constexpr (SHAPE == FB_SHAPE_LINEAR) {
        double s = ceil(z_cell) - z_cell;
        Sz[0] = s; Sz[1] = 1. - s;
    } else {
        int iz = (int)ceil(z_cell) - 2;
        double u = z_cell - iz - 1;
        double v = 1. - u;
        Sz[0] = (1. / 6.) * (v * (v * v));
        Sz[1] = (1. / 6.) * (3. * (u * (u * u)) - 6. * (u * u) + 4.);
        Sz[2] = (1. / 6.) * (3. * (v * (v * v)) - 6. * (v * v) + 4.);
        Sz[3] = (1. / 6.) * (u * (u * u));
    }
}

// Radial shape factors without the axis flip, particle_shapes.py:25-41, 61-80
template <int SHAPE>
__device__ __forceinline__ void shape_r(double r_cell, double beta_n, double *Sr)
{
    if constexpr (SHAPE == FB_SHAPE_LINEAR) {
        int ir = (int)ceil(r_cell) - 1;
        double u = r_cell - ir;
        double s = (1. - u) + beta_n * (1. - u) * u;
        Sr[0] = s; Sr[1] = 1. - s;
    } else {
        int ir = (int)ceil(r_cell) - 2;
        double u = r_cell - ir - 1;
        double v = 1. - u;
        Sr[0] = (1. / 6.) * (v * (v * v));
        double s1 = (1. / 6.) * (3. * (u * (u * u)) - 6. * (u * u) + 4.);
        s1 += beta_n * (1. - u) * u;
        Sr[1] = s1;
        double s2 = (1. / 6.) * (3. * (v * (v * v)) - 6. * (v * v) + 4.);
        s2 -= beta_n * (1. - u) * u;
        Sr[2] = s2;
        Sr[3] = (1. / 6.) * (u * (u * u));
    }
}

// fold an (unwrapped) node index pair into the physical grid
__device__ __forceinline__ void fold_node(int &iz, int &ir, int Nz, int Nr)
{
    if (iz < 0) iz += Nz; else if (iz > Nz - 1) iz -= Nz;
    if (iz < 0) iz += Nz; else if (iz > Nz - 1) iz -= Nz;
    if (ir < 0) ir = -ir - 1; else if (ir > Nr - 1) ir = Nr - 1;
}

constexpr int DEP_PAD = 65;          // panel row stride in doubles (64 particles + 1)
constexpr int DEP_NOKEY = -0x40000000;

template <int SHAPE, int NCOMP, int NM>
struct DepLayout {
    static constexpr int S = ShapeTraits<SHAPE>::S;
    static constexpr int NW = 2 * S * S;              // weights: [mode0 | modes>=1][jz][jr]
    static constexpr int NA = NCOMP * NM * 2;         // amplitudes: [comp][mode][re|im]
    static constexpr int NOUT = S * S * NA;           // outputs of one cell
    static constexpr int OPL = (NOUT + 63) / 64;      // outputs per lane
    static constexpr int WAVE_DOUBLES = (NW + NA) * DEP_PAD + 1;
    static constexpr size_t wave_bytes() { return (size_t)WAVE_DOUBLES * 8; }
};

// NCOMP = 1 (rho) or 3 (Jr,Jt,Jz); this launch handles modes m0 .. m0+NM-1
template <int SHAPE, int NCOMP, int NM>
__global__ __launch_bounds__(256) void k_deposit(long n,
        const double *__restrict__ x, const double *__restrict__ y,
        const double *__restrict__ z, const double *__restrict__ w, double q,
        const double *__restrict__ ux, const double *__restrict__ uy,
        const double *__restrict__ uz, const double *__restrict__ inv_gamma, double c_light,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        DepGrids G, long rs, int m0,
        const double *__restrict__ beta0, const double *__restrict__ betah,
        int chunks_per_wave, unsigned long long *__restrict__ nflush)
{
    using L = DepLayout<SHAPE, NCOMP, NM>;
    constexpr int S = L::S, H = ShapeTraits<SHAPE>::H;
    constexpr int NW = L::NW, NA = L::NA, NOUT = L::NOUT, OPL = L::OPL;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    double *Wl = lds + (size_t)wave * L::WAVE_DOUBLES;
    double *Al = Wl + NW * DEP_PAD;

    // When a cell has few output values (rho: 16 for the linear shape, Nm = 2) the wave is
    // split into NSUB groups of NP2 lanes; group g accumulates particles g, g+NSUB, ... of a
    // run and the groups are summed with xor-shuffles at the end of the run.
    constexpr int NP2 = (NOUT <= 16) ? 16 : (NOUT <= 32) ? 32 : 64;
    constexpr int NSUB = (OPL == 1) ? 64 / NP2 : 1;
    const int sub = (NSUB > 1) ? lane / NP2 : 0;
    // decode the outputs owned by this lane: o = lane + 64 j -> (jz, jr, comp, mode, re/im)
    int o_w0[OPL], o_wh[OPL], o_a[OPL], o_jr[OPL], o_jz[OPL], o_km[OPL];
    double o_sgn[OPL];
    bool o_ok[OPL], o_m0[OPL];
#pragma unroll
    for (int j = 0; j < OPL; j++) {
        int o = (NSUB > 1) ? (lane % NP2) : (lane + 64 * j);
        o_ok[j] = o < NOUT;
        if (!o_ok[j]) o = 0;
        const int ri = o & 1;
        int t = o >> 1;
        const int mm = t % NM; t /= NM;
        const int k = t % NCOMP; t /= NCOMP;
        const int jr = t % S, jz = t / S;
        const int m = m0 + mm;
        o_m0[j] = (m == 0);
        o_w0[j] = (jz * S + jr) * DEP_PAD;
        o_wh[j] = ((S + jz) * S + jr) * DEP_PAD;
        o_a[j] = ((k * NM + mm) * 2 + ri) * DEP_PAD;
        o_jr[j] = jr; o_jz[j] = jz;
        o_km[j] = (k + NCOMP * m) * 2 + ri;
        // rho, Jz: (-1)^m ; Jr, Jt: -(-1)^m (threading_methods.py:143-146, 289-302)
        const double flip = m1pow(m);
        o_sgn[j] = (NCOMP == 1 || k == 2) ? flip : -flip;
    }

    double acc[OPL], acc2[OPL];
#pragma unroll
    for (int j = 0; j < OPL; j++) { acc[j] = 0.; acc2[j] = 0.; }
    int cur_z = DEP_NOKEY, cur_r = DEP_NOKEY;
    unsigned int my_flushes = 0;      // wave-uniform: runs of equal cells seen by this wave

    // With the linear shape and one output per lane, two cells that follow each other along
    // r share one node column: its partial sums slide to the lanes of the lower column
    // instead of being flushed, halving the atomics of an r-ordered stream.
    constexpr bool SLIDE = (SHAPE == FB_SHAPE_LINEAR) && (OPL == 1) && (2 * S * NA <= 64);
    auto flush = [&](bool lower_column_only) {
        if (cur_z == DEP_NOKEY) return;
        my_flushes++;
#pragma unroll
        for (int j = 0; j < OPL; j++) {
            if (!o_ok[j] || acc[j] == 0. || sub != 0) continue;
            if (lower_column_only && o_jr[j] != 0) continue;
            int gz = cur_z + o_jz[j], gr = cur_r + o_jr[j];
            fold_node(gz, gr, Nz, Nr);
            double *g = (double *)(G.g[o_km[j] >> 1] + (long)gz * rs + gr) + (o_km[j] & 1);
            atomicAdd(g, acc[j]);
        }
    };

    const long chunk0 = ((long)blockIdx.x * nwaves + wave) * chunks_per_wave;
    // software pipeline: particle data of chunk ch+1 is requested before chunk ch is
    // processed, hiding the HBM latency behind the staging + accumulation work
    double pn[NCOMP == 1 ? 4 : 8];
    auto prefetch = [&](long ip) {
        if (ip < n) {
            pn[0] = x[ip]; pn[1] = y[ip]; pn[2] = z[ip]; pn[3] = w[ip];
            if constexpr (NCOMP == 3) { pn[4] = ux[ip]; pn[5] = uy[ip]; pn[6] = uz[ip]; pn[7] = inv_gamma[ip]; }
        }
    };
    prefetch(chunk0 * 64 + lane);
    for (int ch = 0; ch < chunks_per_wave; ch++) {
        const long base = (chunk0 + ch) * 64;
        if (base >= n) break;
        const long ip = base + lane;
        // ---- phase 1: lane = particle; stage weights / amplitudes, keep the cell key
        int my_kz = DEP_NOKEY, my_kr = DEP_NOKEY, my_nb = 0;
        double pc[NCOMP == 1 ? 4 : 8];
#pragma unroll
        for (int k = 0; k < (NCOMP == 1 ? 4 : 8); k++) pc[k] = pn[k];
        if (ch + 1 < chunks_per_wave) prefetch(ip + 64);
        if (ip < n) {
            const double xj = pc[0], yj = pc[1], zj = pc[2];
            const double wj = q * pc[3];
            const double rj = sqrt(xj * xj + yj * yj);
            double cs, sn;
            if (rj != 0.) { double invr = 1. / rj; cs = xj * invr; sn = yj * invr; }
            else { cs = 1.; sn = 0.; }
            double are[NCOMP], aim[NCOMP];
            if constexpr (NCOMP == 1) {
                are[0] = wj; aim[0] = 0.;
            } else {
                const double ig = pc[7];
                are[0] = wj * c_light * ig * (cs * pc[4] + sn * pc[5]); aim[0] = 0.;
                are[1] = wj * c_light * ig * (cs * pc[5] - sn * pc[4]); aim[1] = 0.;
                are[2] = wj * c_light * ig * pc[6]; aim[2] = 0.;
            }
            // mode recurrence (cos + i sin)^m, threading_methods.py:119-121, 264-267
            for (int m = 0; m < m0; m++) {
#pragma unroll
                for (int k = 0; k < NCOMP; k++) {
                    double re = cs * are[k] - sn * aim[k], im = cs * aim[k] + sn * are[k];
                    are[k] = re; aim[k] = im;
                }
            }
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
#pragma unroll
                for (int k = 0; k < NCOMP; k++) {
                    Al[((k * NM + mm) * 2 + 0) * DEP_PAD + lane] = are[k];
                    Al[((k * NM + mm) * 2 + 1) * DEP_PAD + lane] = aim[k];
                    double re = cs * are[k] - sn * aim[k], im = cs * aim[k] + sn * are[k];
                    are[k] = re; aim[k] = im;
                }
            }
            const double r_cell = invdr * (rj - rmin) - 0.5;
            const double z_cell = invdz * (zj - zmin) - 0.5;
            const int icr = (int)ceil(r_cell), icz = (int)ceil(z_cell);
            // lowest node of the stencil (unfolded)
            if constexpr (SHAPE == FB_SHAPE_LINEAR) { my_kr = min(icr - 1, Nr); my_kz = icz - 1; }
            else { my_kr = min(icr, Nr) - 2; my_kz = icz - 2; }
            const int ir_ruy = min(icr, Nr);
            double Sz[S], Sr0[S], Srh[S];
            shape_z<SHAPE>(z_cell, Sz);
            shape_r<SHAPE>(r_cell, beta0[ir_ruy], Sr0);
            shape_r<SHAPE>(r_cell, betah[ir_ruy], Srh);
#pragma unroll
            for (int jz = 0; jz < S; jz++)
#pragma unroll
                for (int jr = 0; jr < S; jr++) {
                    Wl[(jz * S + jr) * DEP_PAD + lane] = Sz[jz] * Sr0[jr];
                    Wl[((S + jz) * S + jr) * DEP_PAD + lane] = Sz[jz] * Srh[jr];
                }
            // number of stencil columns below the axis: index + (icr - H) < 0
            my_nb = H - icr;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2: lane = output value.  Segment boundaries (first particle of a new
        // cell) are found with one ballot; each segment is a branch-free register
        // accumulation over its staged particles.
        const int cnt = (int)min((long)64, n - base);
        const int prev_kz = __shfl_up(my_kz, 1), prev_kr = __shfl_up(my_kr, 1);
        bool is_start = (lane == 0) ? (my_kz != cur_z || my_kr != cur_r)
                                    : (my_kz != prev_kz || my_kr != prev_kr);
        const unsigned long long starts = __ballot(is_start && lane < cnt);
        int p = 0;
        while (p < cnt) {
            if ((starts >> p) & 1ull) {
                const int nz_ = __builtin_amdgcn_readlane(my_kz, p);
                const int nr_ = __builtin_amdgcn_readlane(my_kr, p);
#pragma unroll
                for (int j = 0; j < OPL; j++) { acc[j] += acc2[j]; acc2[j] = 0.; }
                if constexpr (NSUB > 1) {
#pragma unroll
                    for (int d = NP2; d < 64; d <<= 1) acc[0] += __shfl_xor(acc[0], d);
                }
                if (SLIDE && nz_ == cur_z && nr_ == cur_r + 1) {
                    flush(true);                     // column cur_r is complete
                    const double up = __shfl_down(acc[0], NA);   // column cur_r+1 carries on
                    acc[0] = (o_jr[0] == 0 && sub == 0) ? up : 0.;
                } else {
                    flush(false);
#pragma unroll
                    for (int j = 0; j < OPL; j++) acc[j] = 0.;
                }
                cur_z = nz_;
                cur_r = nr_;
            }
            const unsigned long long rest = (p + 1 < 64) ? (starts >> (p + 1)) : 0ull;
            int e = rest ? p + 1 + __builtin_ctzll(rest) : cnt;
            if (e > cnt) e = cnt;
            if (cur_r >= 0) {                       // no node of this cell is below the axis
                if constexpr (NSUB > 1) {
                    // group `sub` takes particles p+sub, p+sub+NSUB, ...; two chains
                    int q = p + sub;
                    for (; q + NSUB < e; q += 2 * NSUB) {
                        const double *wp = Wl + (o_m0[0] ? o_w0[0] : o_wh[0]) + q;
                        const double *ap = Al + o_a[0] + q;
                        const double w0 = wp[0], w1 = wp[NSUB], a0 = ap[0], a1 = ap[NSUB];
                        acc[0] = __builtin_fma(w0, a0, acc[0]);
                        acc2[0] = __builtin_fma(w1, a1, acc2[0]);
                    }
                    if (q < e) {
                        const double wv = o_m0[0] ? Wl[o_w0[0] + q] : Wl[o_wh[0] + q];
                        acc[0] = __builtin_fma(wv, Al[o_a[0] + q], acc[0]);
                    }
                    p = e;
                } else {
                // 4 particles per trip, two accumulator chains: all 8 LDS reads are in
                // flight before the first FMA, and the fp64 FMA latency is overlapped
                for (; p + 4 <= e; p += 4) {
#pragma unroll
                    for (int j = 0; j < OPL; j++) {
                        const double *wp = Wl + (o_m0[j] ? o_w0[j] : o_wh[j]) + p;
                        const double *ap = Al + o_a[j] + p;
                        const double w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                        const double a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
                        acc[j] = __builtin_fma(w0, a0, acc[j]);
                        acc2[j] = __builtin_fma(w1, a1, acc2[j]);
                        acc[j] = __builtin_fma(w2, a2, acc[j]);
                        acc2[j] = __builtin_fma(w3, a3, acc2[j]);
                    }
                }
                for (; p < e; p++) {
#pragma unroll
                    for (int j = 0; j < OPL; j++) {
                        const double wv = o_m0[j] ? Wl[o_w0[j] + p] : Wl[o_wh[j] + p];
                        acc[j] = __builtin_fma(wv, Al[o_a[j] + p], acc[j]);
                    }
                }
                }
            } else {
                for (; p < e; p++) {
                    const int nb = __builtin_amdgcn_readlane(my_nb, p);
#pragma unroll
                    for (int j = 0; j < OPL; j++) {
                        double wv = o_m0[j] ? Wl[o_w0[j] + p] : Wl[o_wh[j] + p];
                        if (o_jr[j] < nb) wv *= o_sgn[j];
                        if (sub == 0) acc[j] = __builtin_fma(wv, Al[o_a[j] + p], acc[j]);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int j = 0; j < OPL; j++) acc[j] += acc2[j];
    if constexpr (NSUB > 1) {
#pragma unroll
        for (int d = NP2; d < 64; d <<= 1) acc[0] += __shfl_xor(acc[0], d);
    }
    flush(false);
    // fragmentation statistic for the host's sort policy: 1024 counters (same-address
    // device atomics serialise at ~10 ns each; one shared counter would cost > 100 us)
    if (nflush && lane == 0)
        atomicAdd(nflush + ((blockIdx.x * nwaves + wave) & 1023), (unsigned long long)my_flushes);
}

template <int SHAPE, int NCOMP, int NM>
static int launch_one(long n, const double *x, const double *y, const double *z, const double *w,
        double q, const double *ux, const double *uy, const double *uz, const double *ig,
        double c, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const DepGrids &G, long rs, int m0, const double *b0, const double *bh,
        unsigned long long *nflush, hipStream_t s)
{
    using L = DepLayout<SHAPE, NCOMP, NM>;
    // waves per workgroup: keep the LDS panel <= 64 KiB
    int nwaves = 4;
    while (nwaves > 1 && L::wave_bytes() * nwaves > 64 * 1024) nwaves >>= 1;
    const long nchunks = (n + 63) / 64;
    // ~8 waves per SIMD-quad in flight over 256 CUs, each walking consecutive chunks so
    // that a cell straddling two chunks is not flushed twice
    long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    if (cpw < 1) cpw = 1;
    if (cpw > 64) cpw = 64;
    const long total_waves = (nchunks + cpw - 1) / cpw;
    const long nblocks = (total_waves + nwaves - 1) / nwaves;
    auto kern = k_deposit<SHAPE, NCOMP, NM>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * nwaves),
                       L::wave_bytes() * nwaves, s, n, x, y, z, w, q, ux, uy, uz, ig, c,
                       invdz, zmin, Nz, invdr, rmin, Nr, G, rs, m0, b0, bh, cpw, (m0 == 0) ? nflush : nullptr);
    return check(hipGetLastError(), "fb_deposit");
}

template <int SHAPE, int NCOMP>
static int launch_modes(int Nm, long n, const double *x, const double *y, const double *z,
        const double *w, double q, const double *ux, const double *uy, const double *uz,
        const double *ig, double c, double invdz, double zmin, int Nz, double invdr, double rmin,
        int Nr, const DepGrids &G, long rs, const double *b0, const double *bh,
        unsigned long long *nflush, hipStream_t s)
{
    int m0 = 0;
    while (m0 < Nm) {
        int left = Nm - m0, r;
#define ARGS n, x, y, z, w, q, ux, uy, uz, ig, c, invdz, zmin, Nz, invdr, rmin, Nr, G, rs, m0, b0, bh, nflush, s
        if (left >= 4) { r = launch_one<SHAPE, NCOMP, 4>(ARGS); m0 += 4; }
        else if (left == 3) { r = launch_one<SHAPE, NCOMP, 3>(ARGS); m0 += 3; }
        else if (left == 2) { r = launch_one<SHAPE, NCOMP, 2>(ARGS); m0 += 2; }
        else { r = launch_one<SHAPE, NCOMP, 1>(ARGS); m0 += 1; }
#undef ARGS
        if (r) return r;
    }
    return 0;
}

}  // namespace fb

using namespace fb;

extern "C" int fb_deposit_rho(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, const double *w, double q, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, void *const *rho, long row_stride,
        const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh,
        unsigned long long *nflush, void *stream)
{
    (void)prefix_sum;
    if (n <= 0) return 0;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error("fb_deposit_rho", "Nm out of range"); return -1; }
    DepGrids G;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) G.g[i] = i < Nm ? (cplx *)rho[i] : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (shape == FB_SHAPE_LINEAR)
        return launch_modes<FB_SHAPE_LINEAR, 1>(Nm, n, x, y, z, w, q, nullptr, nullptr, nullptr,
                nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0,
                ruyten_mh, nflush, s);
    if (shape == FB_SHAPE_CUBIC)
        return launch_modes<FB_SHAPE_CUBIC, 1>(Nm, n, x, y, z, w, q, nullptr, nullptr, nullptr,
                nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0,
                ruyten_mh, nflush, s);
    set_error("fb_deposit_rho", "unknown shape");
    return -1;
}

extern "C" int fb_deposit_J(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, const double *w, double q, const double *ux, const double *uy,
        const double *uz, const double *inv_gamma, double c, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, void *const *J, long row_stride,
        const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh,
        unsigned long long *nflush, void *stream)
{
    (void)prefix_sum;
    if (n <= 0) return 0;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error("fb_deposit_J", "Nm out of range"); return -1; }
    DepGrids G;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) G.g[i] = i < 3 * Nm ? (cplx *)J[i] : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (shape == FB_SHAPE_LINEAR)
        return launch_modes<FB_SHAPE_LINEAR, 3>(Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c,
                invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0, ruyten_mh, nflush, s);
    if (shape == FB_SHAPE_CUBIC)
        return launch_modes<FB_SHAPE_CUBIC, 3>(Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c,
                invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, ruyten_m0, ruyten_mh, nflush, s);
    set_error("fb_deposit_J", "unknown shape");
    return -1;
}
