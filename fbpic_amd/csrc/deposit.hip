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

// g[comp + NCOMP*m]: base of each target array; element (iz, ir) at g + iz*rs + ir*cs.
// cs = 1 for the (Nz, Nr) grids of the reference; cs = record length for a node-major
// ("array of structures") target in which all components and modes of one node share a
// cache line -- global atomics cost one L2 operation per LINE touched by an instruction
// (tools/atomic_probe.hip: 48 lanes on 48 lines 2.0 ns, on 3 lines 0.27 ns), so flushing a
// cell into 2-4 lines instead of 24-48 is what makes the J deposition's atomics cheap.
struct DepGrids { cplx *g[3 * FB_MAX_MODES]; long cs; };

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

// panel row stride in doubles: 64 particles + 1 (linear: a group of 16 consecutive particles per
// read); cubic: + 4, so that the 4 x 4 (row, particle) addresses of a column-block read fall on
// 16 different bank pairs
constexpr int dep_pad(int S) { return S == 4 ? 68 : 65; }
constexpr int DEP_NOKEY = -0x40000000;

// Phase 2 is a small matrix product per cell: out[node][amplitude] = sum over the particles
// of the cell of W[node][p] * A[amplitude][p].  It runs on the matrix cores with
// v_mfma_f64_4x4x4_4b_f64: 4 independent blocks of (4 nodes x 4 particles).(4 particles x 4
// amplitudes), i.e. 16 particles per instruction and 16 cycles per issue, operands taken
// straight from the staged LDS panels.  Lane layout (measured on gfx950 with
// tools/mfma4_probe.hip):  A operand lane l = A[i = l&3][k = l>>4] of block (l>>2)&3,
// B operand lane l = B[k = l>>4][j = l&3] of block (l>>2)&3, D lane l = D[i = l>>4][j = l&3]
// of block (l>>2)&3.
//
// Amplitude rows of the panel: the first mode of the launch, then the others.  When the
// first mode is m = 0 (Z0) its imaginary parts are identically zero and are not staged.
// Column tiles of 4 amplitudes never mix mode 0 with modes >= 1 because the two use
// different radial weights (Ruyten coefficients).
template <int SHAPE, int NCOMP, int NM, bool Z0>
struct DepLayout {
    static constexpr int S = ShapeTraits<SHAPE>::S;
    static constexpr int NPT = S * S;                 // nodes of one cell
    static constexpr int RG = NPT / 4;                // row groups of 4 nodes
    // shape-factor rows of the panel: Sz[jz] | Sr of mode 0 [jr] | Sr of modes >= 1 [jr]; the
    // node weights Sz[jz] * Sr[jr] are formed when the matrix operand is read (12 rows instead
    // of 2 x 16 products for the cubic shape: the panel of J, Nm = 4 shrinks from 27.5 to
    // 17 KB per wave and twice as many waves fit a CU; 6 instead of 8 for the linear shape)
    static constexpr int NW = 3 * S;
    static constexpr int R1 = Z0 ? NCOMP : 2 * NCOMP; // amplitude rows of the first mode
    static constexpr int T1 = (R1 + 3) / 4;
    static constexpr int RH = (NM - 1) * NCOMP * 2;   // rows of the other modes
    static constexpr int TH = (RH + 3) / 4;
    static constexpr int NT = T1 + TH;                // column tiles
    // amplitude rows actually staged: the panel holds no padding rows (a column tile whose
    // last columns are unused reads its last valid row again; those outputs are discarded).
    // J, linear, Nm = 2: 17 rows x 65 doubles = 8.8 KB per wave -> 4 workgroups of 4 waves per
    // CU instead of 3 with the padded 20 rows; rho: 11 rows -> 6 instead of 4.
    static constexpr int NA = R1 + RH;
    static constexpr int PAD = dep_pad(S);
    static constexpr int WAVE_DOUBLES = (NW + NA) * PAD + 1;
    static constexpr size_t wave_bytes() { return (size_t)WAVE_DOUBLES * 8; }
    // panel row of amplitude (component k, launch-local mode mm, re/im)
    __host__ __device__ static constexpr int row(int k, int mm, int ri)
    {
        return (mm == 0) ? (Z0 ? k : 2 * k + ri) : R1 + ((mm - 1) * NCOMP + k) * 2 + ri;
    }
    // panel row read by the lanes with column index jl of tile t
    __host__ __device__ static constexpr int tile_row(int t, int jl)
    {
        return (t < T1) ? ((4 * t + jl < R1) ? 4 * t + jl : R1 - 1)
                        : R1 + ((4 * (t - T1) + jl < RH) ? 4 * (t - T1) + jl : RH - 1);
    }
};

// rotate within rows of 16 lanes (DPP row_ror:N), used to add up the 4 MFMA blocks
template <int N>
__device__ __forceinline__ double row_ror(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (mov_dpp, not update_dpp(0, ..): a row rotation writes every lane, and the "old" operand of
    // update_dpp cost a v_mov_b32 0 per half - 4 of the 8 instructions of a block sum)
    lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Optional front end of the rho deposition (fb_push_x_sort_deposit_rho): the wave walks the
// particles in DESTINATION order of the counting sort.  Lane ip reads its 8 attributes through
// the inverse permutation sidx (nearly sequential: a particle moves at most a cell per step),
// evaluates the pending push_x in registers (same expression as k_push_x / k_scatter), writes
// the attributes contiguously at their sorted slot and deposits the charge of the pushed
// position.  One pass does what k_scatter (72 B read + 64 B written per particle) and the rho
// deposition (32 B read again, r / cos / sin / cell recomputed) did in two; the arithmetic of
// the deposition overlaps the memory stalls of the permutation.
struct PermArgs {
    const int *sidx;              // n, destination -> source
    CPtrs16 src;                  // x, y, z, ux, uy, uz, w, inv_gamma [, extra attributes]
    Ptrs16 dst;
    int nattr;
    double chdt, px, py, pz;
    const int *cell;              // source-ordered cell of the pushed position (sort workspace)
    int *cell_sorted;             // optional output
};

struct DepGeom { double invdz, zmin; int Nz; double invdr, rmin; int Nr; };

// One deposition "engine": the per-wave state and the two phases of the run-based deposition
// of NCOMP components x NM modes (modes m0 .. m0+NM-1; Z0 <=> m0 == 0).  A kernel may run
// several engines one after the other on the same LDS panel (k_perm_deposit_J_rho: J at the
// position before the push, rho at the position after it).
template <int SHAPE, int NCOMP, int NM, bool Z0>
struct DepEngine {
    using L = DepLayout<SHAPE, NCOMP, NM, Z0>;
    static constexpr int S = L::S, H = ShapeTraits<SHAPE>::H;
    static constexpr int NPT = L::NPT, RG = L::RG, NW = L::NW, NT = L::NT, T1 = L::T1;
    static constexpr int DEP_PAD = L::PAD;
    static constexpr bool NEED_W0 = Z0, NEED_WH = (!Z0) || (NM > 1);
    // Accumulator tile u = rg * NT + t holds, in lane l, node rg*4 + kl x amplitude row 4t + jl
    // of block bl.  At a flush the 4 blocks are added (row rotations) and the lanes of block b
    // write tile 4 q + b in round q: one atomic instruction per 4 tiles.
    static constexpr int NTILE = RG * NT, NQ = (NTILE + 3) / 4;
    // Cubic shape, CB ("column blocks"): the 4 blocks of the MFMA are the 4 node rows (jz) of the
    // cell instead of 4 groups of particles - block b multiplies W[nodes (b, jr = 0..3)][4
    // particles] by the amplitudes of the same 4 particles (B operand replicated over the
    // blocks: a broadcast LDS read).  One instruction covers 4 particles x 16 nodes x 4
    // amplitudes, the same 256 MACs, but every lane then holds ONE finished sum per column tile
    // (node (bl, kl), amplitude jl): a quarter of the accumulator registers (J, Nm = 4: 12
    // instead of 48) and no cross-block reduction at a flush, which was ~40 % of the kernel's VALU
    // instructions (24 tiles x (4 DPP moves + 2 adds + 2 selects) per run; SQ counters, 2048 x 512
    // x 64 ppc: 1187 VALU + 120 MFMA instructions per 64 particles, fp64 MFMA and VALU do not
    // overlap on gfx950, tools/overlap_probe.hip).
    static constexpr int STRAY_MAX = 4;          // longest run handled out of band (see reduce)
#ifdef FB_NO_STRAYS
    static constexpr bool STRAYS = false;
#else
    // cubic shape only: a flush of the linear shape is one atomic instruction, and the second
    // product / flush path costs the fused linear pass more than the strays do (205 -> 216 us)
    static constexpr bool STRAYS = (S == 4);
#endif
    static constexpr bool CB = (S == 4);
    static constexpr int NF = CB ? NT : NQ;      // values a lane flushes per run
    static constexpr int NAR = CB ? 1 : RG;

    double *Wl, *Al;
    int lane, jl, bl, kl, poff;    // poff: particle (within a group of 16) fed by this lane
    // grid base of this lane's amplitude (re or im part), advanced to the lane's node row
    // (+ 2 * f_jz * rs); the sign of its below-axis mirror and its validity are bit masks
    double *f_ptrz[NF];
    unsigned f_neg, f_okm;
    int f_jz[NF], f_jr[NF];
    long rs, cs, cs2;              // row / column stride in elements, column stride in doubles
    int Nz, Nr, m0;
    double acc[NAR][NT];
    int aoff[NT];                  // LDS offset of the amplitude row this lane feeds to tile t
    int cur_z, cur_r, cur_nb;
    unsigned int my_flushes;       // wave-uniform: runs of equal cells seen by this wave
    // Two cells that follow each other along r share S-1 of their S node columns.  The
    // partial sums of those columns are not flushed: their accumulator lanes simply take the
    // role of the next-lower column of the new cell (`off` rotates which radial weight feeds
    // which lane), and only the lowest column of the finished cell is written out - 1/S of
    // the atomics of an r-ordered stream (1/2 for the linear shape, 1/4 for the cubic one).
    // Logical column of a lane whose physical column index is j: (j - off) mod S.
    int off;

    __device__ __forceinline__ void init(double *panel, int lane_, const DepGrids &G, long rs_, int m0_,
                                         int Nz_, int Nr_)
    {
        Wl = panel;
        Al = panel + NW * DEP_PAD;
        lane = lane_;
        jl = lane & 3; bl = (lane >> 2) & 3; kl = lane >> 4;
        poff = 4 * bl + kl;
        rs = rs_; cs = G.cs; cs2 = 2 * G.cs; Nz = Nz_; Nr = Nr_; m0 = m0_;
        f_neg = 0u; f_okm = 0u;
#pragma unroll
        for (int qq = 0; qq < NF; qq++) {
            // CB: value qq of this lane = tile qq of its node (bl, kl); else tile 4 qq + bl
            const int u = CB ? bl * NT + qq : 4 * qq + bl;
            const int rg = u / NT, t = u % NT;
            int k, mm, ri;
            bool ok = u < NTILE;
            if (t < T1) {
                const int idx = 4 * t + jl;
                ok = ok && idx < L::R1;
                mm = 0;
                if (Z0) { k = idx; ri = 0; } else { k = idx >> 1; ri = idx & 1; }
            } else {
                const int idx = 4 * (t - T1) + jl;
                ok = ok && idx < L::RH;
                ri = idx & 1;
                k = (idx >> 1) % NCOMP;
                mm = 1 + (idx >> 1) / NCOMP;
            }
            if (!ok) { k = 0; mm = 0; ri = 0; }
            const int m = m0 + mm;
            f_okm |= ok ? (1u << qq) : 0u;
            double *fp = (double *)G.g[k + NCOMP * m] + ri;
            // rho, Jz: (-1)^m ; Jr, Jt: -(-1)^m (threading_methods.py:143-146, 289-302)
            const double flip = m1pow(m);
            f_neg |= (((NCOMP == 1 || k == 2) ? flip : -flip) < 0.) ? (1u << qq) : 0u;
            const int pt = (rg % RG) * 4 + kl;
            f_jz[qq] = pt / S; f_jr[qq] = pt % S;
            f_ptrz[qq] = fp + 2 * ((long)f_jz[qq] * rs);
        }
#pragma unroll
        for (int t = 0; t < NT; t++) aoff[t] = L::tile_row(t, jl) * DEP_PAD;
#pragma unroll
        for (int rg = 0; rg < NAR; rg++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[rg][t] = 0.;
        cur_z = DEP_NOKEY; cur_r = DEP_NOKEY; cur_nb = 0;
        my_flushes = 0;
        off = 0;
    }

    // flush the sums A of cell (cz, cr); keep_upper: only its lowest node column (the others
    // carry on); off_: rotation of the radial columns (see `off`)
    __device__ __forceinline__ void flush_acc(const double (&A)[NAR][NT], int cz, int cr, int cnb,
                                              int off_, bool keep_upper)
    {
        my_flushes++;
        const bool interior = cz >= 0 && cz + S <= Nz && cr >= 0 && cr + S <= Nr;
        // offset of the cell's lowest node in doubles: wave-uniform, scalar arithmetic
        const long cell_base2 = 2 * ((long)cz * rs + (long)cr * cs);
#pragma unroll
        for (int qq = 0; qq < NF; qq++) {
            double v = 0.;
            if constexpr (CB) {
                v = A[0][qq];          // already the sum over the particles of the run
            } else {
                // add the 4 blocks of each tile, then keep the tile this lane writes in this round
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int u = 4 * qq + b;
                    if (u < NTILE) {
                        double a = A[u / NT][u % NT];
                        a += row_ror<4>(a);
                        a += row_ror<8>(a);
                        v = (bl == b) ? a : v;
                    }
                }
            }
            const int jr = (f_jr[qq] - off_) & (S - 1);
            if (!((f_okm >> qq) & 1u) || v == 0. || (keep_upper && jr != 0)) continue;
            if (interior) {
                // all S x S nodes inside the grid (wave-uniform test): no guard folding, no
                // axis sign
                long joff = (jr & 1) ? cs2 : 0;
                if constexpr (S > 2) joff += (jr & 2) ? 2 * cs2 : 0;
                atomicAdd(f_ptrz[qq] + cell_base2 + joff, v);
            } else {
                int gz = cz + f_jz[qq], gr = cr + jr;
                fold_node(gz, gr, Nz, Nr);
                if (jr < cnb && ((f_neg >> qq) & 1u)) v = -v;   // node below the axis: signed fold
                atomicAdd(f_ptrz[qq] + 2 * ((long)(gz - f_jz[qq]) * rs + (long)gr * cs), v);
            }
        }
    }
    // flush the current cell
    __device__ __forceinline__ void flush(bool keep_upper)
    {
        if (cur_z == DEP_NOKEY) return;
        flush_acc(acc, cur_z, cur_r, cur_nb, off, keep_upper);
    }

    // A += W . amplitudes over the staged particles [p, e) (one run); off_ as in flush_acc
    __device__ __forceinline__ void product(double (&A)[NAR][NT], int p, int e, int off_)
    {
        // node fed by this lane in row group rg: rg*4 + (its logical column).  Linear shape:
        // jl = jz*2 + jr and only the jr bit rotates; cubic: jz = rg, jr = (jl - off) mod 4
        const int wrow = (S == 2) ? (jl ^ off_) : ((jl - off_) & 3);
        const int jr_row = (S == 2) ? (wrow & 1) : wrow;
        if constexpr (CB) {
            // steps of 4 particles; lane (bl, jl, kl): A = W[node (bl, jr_row)][particle 4 st + kl],
            // B = amplitude row of (tile, jl) of the same particle
            const int s1 = (e - 1) >> 2;
            for (int st = p >> 2; st <= s1; st++) {
                const int pi = 4 * st + kl;
                const bool in = (pi >= p) && (pi < e);
                const double sz = Wl[bl * DEP_PAD + pi];
                double w0 = 0., wh = 0.;
                if constexpr (NEED_W0) { const double v = Wl[(S + jr_row) * DEP_PAD + pi]; w0 = in ? sz * v : 0.; }
                if constexpr (NEED_WH) { const double v = Wl[(2 * S + jr_row) * DEP_PAD + pi]; wh = in ? sz * v : 0.; }
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const double av = Al[aoff[t] + pi];
                    double wv;
                    if constexpr (!NEED_WH) wv = w0;
                    else if constexpr (!NEED_W0) wv = wh;
                    else wv = (t < T1) ? w0 : wh;
                    A[0][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(wv, av, A[0][t], 0, 0, 0);
                }
            }
        } else {
            const int g1 = (e - 1) >> 4;
            for (int g = p >> 4; g <= g1; g++) {
                const int pi = 16 * g + poff;
                const bool in = (pi >= p) && (pi < e);
                double w0[RG], wh[RG];
                double sr0 = 0., srh = 0.;
                if constexpr (NEED_W0) { const double v = Wl[(S + jr_row) * DEP_PAD + pi]; sr0 = in ? v : 0.; }
                if constexpr (NEED_WH) { const double v = Wl[(2 * S + jr_row) * DEP_PAD + pi]; srh = in ? v : 0.; }
#pragma unroll
                for (int rg = 0; rg < RG; rg++) {
                    const double sz = Wl[((S == 2) ? (wrow >> 1) : rg) * DEP_PAD + pi];
                    if constexpr (NEED_W0) w0[rg] = sz * sr0;
                    if constexpr (NEED_WH) wh[rg] = sz * srh;
                }
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const double av = Al[aoff[t] + pi];
#pragma unroll
                    for (int rg = 0; rg < RG; rg++) {
                        double wv;
                        if constexpr (!NEED_WH) wv = w0[rg];
                        else if constexpr (!NEED_W0) wv = wh[rg];
                        else wv = (t < T1) ? w0[rg] : wh[rg];
                        A[rg][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(wv, av, A[rg][t], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- phase 1: lane = particle; stage weights / amplitudes, return the cell key.
    // u[0..2], ig, c_light are only read for NCOMP == 3.
    __device__ __forceinline__ void stage(bool act, double xj, double yj, double zj, double wj,
            double ux, double uy, double uz, double ig, double c_light, const DepGeom &g,
            const double *__restrict__ beta0, const double *__restrict__ betah,
            int &my_kz, int &my_kr, int &my_nb)
    {
        my_kz = DEP_NOKEY; my_kr = DEP_NOKEY; my_nb = 0;
        if (act) {
            const double rj = sqrt(xj * xj + yj * yj);
            double cs_, sn;
            if (rj != 0.) {
                // 1/r by hardware reciprocal + two Newton steps (< 1 ulp): the deposition is
                // compared at 1e-13, only the cell index below needs the exactly rounded r
                double r0 = __builtin_amdgcn_rcp(rj);
                r0 = __builtin_fma(r0, __builtin_fma(-rj, r0, 1.), r0);
                const double invr = __builtin_fma(r0, __builtin_fma(-rj, r0, 1.), r0);
                cs_ = xj * invr; sn = yj * invr;
            } else { cs_ = 1.; sn = 0.; }
            double are[NCOMP], aim[NCOMP];
            if constexpr (NCOMP == 1) {
                are[0] = wj; aim[0] = 0.;
            } else {
                are[0] = wj * c_light * ig * (cs_ * ux + sn * uy); aim[0] = 0.;
                are[1] = wj * c_light * ig * (cs_ * uy - sn * ux); aim[1] = 0.;
                are[2] = wj * c_light * ig * uz; aim[2] = 0.;
            }
            // mode recurrence (cos + i sin)^m, threading_methods.py:119-121, 264-267
            if constexpr (!Z0) {
                for (int m = 0; m < m0; m++) {
#pragma unroll
                    for (int k = 0; k < NCOMP; k++) {
                        double re = cs_ * are[k] - sn * aim[k], im = cs_ * aim[k] + sn * are[k];
                        are[k] = re; aim[k] = im;
                    }
                }
            }
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
#pragma unroll
                for (int k = 0; k < NCOMP; k++) {
                    Al[L::row(k, mm, 0) * DEP_PAD + lane] = are[k];
                    if (!(Z0 && mm == 0)) Al[L::row(k, mm, 1) * DEP_PAD + lane] = aim[k];
                    double re = cs_ * are[k] - sn * aim[k], im = cs_ * aim[k] + sn * are[k];
                    are[k] = re; aim[k] = im;
                }
            }
            const double r_cell = g.invdr * (rj - g.rmin) - 0.5;
            const double z_cell = g.invdz * (zj - g.zmin) - 0.5;
            const int icr = (int)ceil(r_cell), icz = (int)ceil(z_cell);
            // lowest node of the stencil (unfolded)
            if constexpr (SHAPE == FB_SHAPE_LINEAR) { my_kr = min(icr - 1, Nr); my_kz = icz - 1; }
            else { my_kr = min(icr, Nr) - 2; my_kz = icz - 2; }
            const int ir_ruy = min(icr, Nr);
            double Sz[S], Sr0[S], Srh[S];
            shape_z<SHAPE>(z_cell, Sz);
            if constexpr (NEED_W0) shape_r<SHAPE>(r_cell, beta0[ir_ruy], Sr0);
            if constexpr (NEED_WH) shape_r<SHAPE>(r_cell, betah[ir_ruy], Srh);
#pragma unroll
            for (int j = 0; j < S; j++) {
                Wl[j * DEP_PAD + lane] = Sz[j];
                if constexpr (NEED_W0) Wl[(S + j) * DEP_PAD + lane] = Sr0[j];
                if constexpr (NEED_WH) Wl[(2 * S + j) * DEP_PAD + lane] = Srh[j];
            }
            // number of stencil columns below the axis: index + (icr - H) < 0
            my_nb = H - icr;
        } else {
            // tail of the stream: finite amplitudes for the (masked) matrix operands
#pragma unroll
            for (int a = 0; a < L::NA; a++) Al[a * DEP_PAD + lane] = 0.;
        }
    }

    // ---- phase 2: runs of equal cells (boundaries found with one ballot) are reduced on the
    // matrix cores, 16 staged particles per instruction; particles of a group that belong to
    // another run are masked out of the weight operand.
    // (Measured with knock-out builds of the fused J + rho pass, 181 us: without phase 2 123 us,
    // without phase 1 as well 119 us = the permutation alone; without the atomics -11 us.
    // Phase 2 is a chain of short dependent steps per run that the other waves of the SIMD
    // only partly cover.  Requesting all LDS operands of a chunk up front costs 24 VGPRs, i.e.
    // one wave per SIMD, and loses: 208 -> 233 us.)
    __device__ __forceinline__ void reduce(int cnt, int my_kz, int my_kr, int my_nb)
    {
        const int prev_kz = __shfl_up(my_kz, 1), prev_kr = __shfl_up(my_kr, 1);
        bool is_start = (lane == 0) ? (my_kz != cur_z || my_kr != cur_r)
                                    : (my_kz != prev_kz || my_kr != prev_kr);
        const unsigned long long starts = __ballot(is_start && lane < cnt);
        int p = 0;
        while (p < cnt) {
            const unsigned long long rest = (p + 1 < 64) ? (starts >> (p + 1)) : 0ull;
            int e = rest ? p + 1 + __builtin_ctzll(rest) : cnt;
            if (e > cnt) e = cnt;
            if ((starts >> p) & 1ull) {
                const int nz_ = __builtin_amdgcn_readlane(my_kz, p);
                const int nr_ = __builtin_amdgcn_readlane(my_kr, p);
                if (nz_ == cur_z && nr_ == cur_r) {
                    // the current cell again (after a stray, below): carry on
                } else if (nz_ == cur_z && nr_ == cur_r + 1) {
                    flush(true);                     // column cur_r is complete
                    // the other columns carry on, one position lower in the new cell
                    const bool carry = (((kl & (S - 1)) - off) & (S - 1)) != 0;
#pragma unroll
                    for (int rg = 0; rg < NAR; rg++)
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[rg][t] = carry ? acc[rg][t] : 0.;
                    off = (off + 1) & (S - 1);
                    cur_r = nr_;
                    cur_nb = __builtin_amdgcn_readlane(my_nb, p);
                } else if (STRAYS && cur_z != DEP_NOKEY && e - p <= STRAY_MAX && e < cnt) {
                    // A stray: a few particles that sit among those of the current cell but
                    // deposit into another one (the stream is sorted by the cell of a position
                    // half a step or a step away; ~1-2 % of a thermal plasma's particles).
                    // Ending the current cell's run for them would flush it twice and lose its
                    // sliding columns: their sums go out directly instead, the current cell's
                    // accumulation continues.  (2048 x 512 x 64 ppc cubic Nm = 4, once the lattice
                    // has thermalised: J deposition 7.7 ms per step with every stray ending the run.)
                    double tmp[NAR][NT];
#pragma unroll
                    for (int rg = 0; rg < NAR; rg++)
#pragma unroll
                        for (int t = 0; t < NT; t++) tmp[rg][t] = 0.;
                    product(tmp, p, e, 0);
                    flush_acc(tmp, nz_, nr_, __builtin_amdgcn_readlane(my_nb, p), 0, false);
                    p = e;
                    continue;
                } else {
                    flush(false);
#pragma unroll
                    for (int rg = 0; rg < NAR; rg++)
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[rg][t] = 0.;
                    off = 0;
                    cur_z = nz_;
                    cur_r = nr_;
                    cur_nb = __builtin_amdgcn_readlane(my_nb, p);
                }
            }
            product(acc, p, e, off);
            p = e;
        }
    }
};

__device__ __forceinline__ void wave_lds_release()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_lds_acquire()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

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
                for (int k = 0; k < 8; k++) pn[k] = PM.src.p[k][idx_n];
            } else {
                pn[0] = x[ip]; pn[1] = y[ip]; pn[2] = z[ip]; pn[3] = w[ip];
                if constexpr (NCOMP == 3) { pn[4] = ux[ip]; pn[5] = uy[ip]; pn[6] = uz[ip]; pn[7] = inv_gamma[ip]; }
            }
        }
    };
    // PERM: two-stage pipeline - the index of chunk ch+2 is requested while the attributes
    // of chunk ch+1 (through the index requested one iteration earlier) are in flight
    if constexpr (PERM) { if (chunk0 * 64 + lane < n) idx_n = PM.sidx[chunk0 * 64 + lane]; }
    prefetch(chunk0 * 64 + lane);
    int idx_nn = 0;
    if constexpr (PERM) {
        idx_c = idx_n;
        if (chunks_per_wave > 1 && (chunk0 + 1) * 64 + lane < n) idx_nn = PM.sidx[(chunk0 + 1) * 64 + lane];
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
            if (ch + 2 < chunks_per_wave && ip + 128 < n) idx_nn = PM.sidx[ip + 128];
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
                PM.dst.p[0][ip] = xj; PM.dst.p[1][ip] = yj; PM.dst.p[2][ip] = zj;
                PM.dst.p[3][ip] = pc[3]; PM.dst.p[4][ip] = pc[4]; PM.dst.p[5][ip] = pc[5];
                PM.dst.p[6][ip] = pc[6]; PM.dst.p[7][ip] = g;
                for (int k = 8; k < PM.nattr; k++) PM.dst.p[k][ip] = PM.src.p[k][idx_c];
                if (PM.cell_sorted) PM.cell_sorted[ip] = PM.cell[idx_c];
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
        int chunks_per_wave, PermArgs PM)
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

    const long chunk0 = (xcd_block_id() * nwaves + wave) * chunks_per_wave;
    double pn[8];
    int idx_n = 0, idx_c = 0, idx_nn = 0;
    auto prefetch = [&](long ip) {
        if (ip < n) {
#pragma unroll
            for (int k = 0; k < 8; k++) pn[k] = PM.src.p[k][idx_n];
        }
    };
    if (chunk0 * 64 + lane < n) idx_n = PM.sidx[chunk0 * 64 + lane];
    prefetch(chunk0 * 64 + lane);
    idx_c = idx_n;
    if (chunks_per_wave > 1 && (chunk0 + 1) * 64 + lane < n) idx_nn = PM.sidx[(chunk0 + 1) * 64 + lane];
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
        if (ch + 2 < chunks_per_wave && ip + 128 < n) idx_nn = PM.sidx[ip + 128];
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
            PM.dst.p[0][ip] = xj; PM.dst.p[1][ip] = yj; PM.dst.p[2][ip] = zj;
            PM.dst.p[3][ip] = pc[3]; PM.dst.p[4][ip] = pc[4]; PM.dst.p[5][ip] = pc[5];
            PM.dst.p[6][ip] = pc[6]; PM.dst.p[7][ip] = g;
            for (int k = 8; k < PM.nattr; k++) PM.dst.p[k][ip] = PM.src.p[k][idx_c];
            if (PM.cell_sorted) PM.cell_sorted[ip] = PM.cell[idx_c];
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
    const int nwaves = dep_waves_per_workgroup(wave_bytes);
    const long nchunks = (n + 63) / 64;
    const long target_waves = 256L * 64;
    int cpw = (int)((nchunks + target_waves - 1) / target_waves);
    if (cpw < 1) cpw = 1;
    if (cpw > 64) cpw = 64;
    const long total_waves = (nchunks + cpw - 1) / cpw;
    const long nblocks = xcd_grid((total_waves + nwaves - 1) / nwaves);
    hipLaunchKernelGGL((k_perm_deposit_J_rho<SHAPE, NM>), dim3((unsigned)nblocks), dim3(64 * nwaves),
                       wave_bytes * nwaves, s, n, q, c, gJ, gR, GJ, rsJ, GR, rsR, b0, bh, cpw, PM);
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
        const double *ruyten_m0, const double *ruyten_mh, void *stream)
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
#define LPJR(SH, NM_) launch_perm_J_rho<SH, NM_>(n, q, c, gJ, gR, GJ, J_row_stride, GR, row_stride, \
                                                 ruyten_m0, ruyten_mh, PM, s)
    if (shape == FB_SHAPE_LINEAR)
        return Nm == 1 ? LPJR(FB_SHAPE_LINEAR, 1) : Nm == 2 ? LPJR(FB_SHAPE_LINEAR, 2)
             : Nm == 3 ? LPJR(FB_SHAPE_LINEAR, 3) : LPJR(FB_SHAPE_LINEAR, 4);
    return Nm == 1 ? LPJR(FB_SHAPE_CUBIC, 1) : Nm == 2 ? LPJR(FB_SHAPE_CUBIC, 2)
         : Nm == 3 ? LPJR(FB_SHAPE_CUBIC, 3) : LPJR(FB_SHAPE_CUBIC, 4);
#undef LPJR
}
