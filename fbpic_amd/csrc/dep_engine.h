// Deposition engine shared by deposit.hip (stand-alone / sort-fused depositions) and cycle.hip
// (the one-pass particle cycle): shape factors, LDS panel layout, the run-based MFMA reduction
// and the flush with guard-cell folding.  See the header comment of deposit.hip for the design.
#pragma once
#include "fb_common.h"

namespace fb {

// g[comp + NCOMP*m]: base of each target array; element (iz, ir) at g + iz*rs + ir*cs.
// cs = 1 for the (Nz, Nr) grids of the reference; cs = record length for a node-major
// ("array of structures") target in which all components and modes of one node share a
// cache line -- global atomics cost one L2 operation per LINE touched by an instruction
// (tools/atomic_probe.hip: 48 lanes on 48 lines 2.0 ns, on 3 lines 0.27 ns), so flushing a
// cell into 2-4 lines instead of 24-48 is what makes the J deposition's atomics cheap.
struct DepGrids { cplx *g[3 * FB_MAX_MODES]; long cs; };

// REL engines (DepEngine<..., true>) address the target as ONE scalar base + a 32-bit byte offset
// per lane (global_atomic_add_f64 v_offset, v_data, s[base]) with 32-bit offset arithmetic, instead
// of a 64-bit pointer per lane: for targets whose arrays all lie within 4 GiB of the lowest one
// (the node-major records; the fields of one slab).  dep_grids_base() returns that base, or null
// when the arrays are spread further apart (then only the pointer engines apply).
inline cplx *dep_grids_base(const DepGrids &G, int nused, long rs, int Nz)
{
    uintptr_t lo = ~(uintptr_t)0, hi = 0;
    for (int i = 0; i < nused; i++) {
        const uintptr_t a = (uintptr_t)G.g[i];
        if (a < lo) lo = a;
        if (a > hi) hi = a;
    }
    const double span = (double)(hi - lo) + 16. * (double)rs * (double)(Nz + 1);
    return (nused > 0 && span < 4294967296.) ? (cplx *)lo : nullptr;
}

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
template <int SHAPE, int NCOMP, int NM, bool Z0, bool REL = false>
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
    // REL: the same as a byte offset from `gbase`; strides as 32-bit byte counts
    unsigned f_off[NF];
    char *gbase;
    int rsB, csB;                  // bytes between z rows / between r columns of one array
    unsigned f_neg, f_okm;
    int f_jz[NF], f_jr[NF];
    long rs, cs, cs2;              // row / column stride in elements, column stride in doubles
    int Nz, Nr, m0;
    double acc[NAR][NT];
    int aoff[NT];                  // LDS offset of the amplitude row this lane feeds to tile t
    // direct scatter of a single particle (scatter_one): LDS offsets of the Sz row, the Sr row and
    // the amplitude row behind value qq of this lane
    int s_wz[NF], s_wr[NF], s_am[NF];
    int cur_z, cur_r, cur_nb;
    unsigned int my_flushes;       // wave-uniform: runs of equal cells seen by this wave
    // Two cells that follow each other along r share S-1 of their S node columns.  The
    // partial sums of those columns are not flushed: their accumulator lanes simply take the
    // role of the next-lower column of the new cell (`off` rotates which radial weight feeds
    // which lane), and only the lowest column of the finished cell is written out - 1/S of
    // the atomics of an r-ordered stream (1/2 for the linear shape, 1/4 for the cubic one).
    // Logical column of a lane whose physical column index is j: (j - off) mod S.
    int off;

    // gbase_: dep_grids_base() of the target (REL engines only)
    __device__ __forceinline__ void init(double *panel, int lane_, const DepGrids &G, long rs_, int m0_,
                                         int Nz_, int Nr_, cplx *gbase_ = nullptr)
    {
        gbase = (char *)gbase_;
        rsB = (int)(16 * rs_); csB = (int)(16 * G.cs);
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
            if constexpr (REL) f_off[qq] = (unsigned)((char *)f_ptrz[qq] - gbase);
            s_wz[qq] = f_jz[qq] * DEP_PAD;
            s_wr[qq] = (((t < T1) ? S : 2 * S) + f_jr[qq]) * DEP_PAD;
            s_am[qq] = L::tile_row(t, jl) * DEP_PAD;
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
        double vv[NF];
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
            vv[qq] = v;
        }
        flush_values(vv, cz, cr, cnb, off_, keep_upper);
    }
    // value qq of every lane -> its node of cell (cz, cr), one atomic instruction per qq
    __device__ __forceinline__ void flush_values(const double (&vv)[NF], int cz, int cr, int cnb,
                                                 int off_, bool keep_upper)
    {
        const bool interior = cz >= 0 && cz + S <= Nz && cr >= 0 && cr + S <= Nr;
        // offset of the cell's lowest node in doubles: wave-uniform, scalar arithmetic
        const long cell_base2 = REL ? 0 : 2 * ((long)cz * rs + (long)cr * cs);
        const int cell_baseB = cz * rsB + cr * csB;       // REL: the same in bytes
#pragma unroll
        for (int qq = 0; qq < NF; qq++) {
            double v = vv[qq];
            const int jr = (f_jr[qq] - off_) & (S - 1);
            if (!((f_okm >> qq) & 1u) || v == 0. || (keep_upper && jr != 0)) continue;
            if constexpr (REL) {
                unsigned voff;
                if (interior) {
                    voff = f_off[qq] + (unsigned)(cell_baseB + jr * csB);
                } else {
                    int gz = cz + f_jz[qq], gr = cr + jr;
                    fold_node(gz, gr, Nz, Nr);
                    if (jr < cnb && ((f_neg >> qq) & 1u)) v = -v;   // node below the axis: signed fold
                    voff = f_off[qq] + (unsigned)((gz - f_jz[qq]) * rsB + gr * csB);
                }
                atomicAdd((double *)(gbase + voff), v);
                continue;
            }
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
    // `pmask`: bit pi clear -> staged particle pi is left out (a stray of reduce_home)
    __device__ __forceinline__ void product(double (&A)[NAR][NT], int p, int e, int off_,
                                            unsigned long long pmask = ~0ull)
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
                const bool in = (pi >= p) && (pi < e) && ((pmask >> pi) & 1ull);
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
            // groups of 16 counted from the START of the run (a run of 16 is one group wherever it
            // begins; with groups aligned to the chunk it was two half-filled ones: 28 -> ~20 MFMA
            // instructions and a third fewer LDS round trips per 64 particles at 16-32 ppc)
            for (int b0 = p; b0 < e; b0 += 16) {
                const int pu = b0 + poff;
                const bool in = (pu < e) && ((pmask >> (pu & 63)) & 1ull);
                // (a masked lane may point beyond the chunk: it reads the last particle's row
                // entries - finite values times a zero weight)
                const int pi = min(pu, 63);
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
        double b0 = 0., bh = 0.;
        if (act) {
            const int ir_ruy = ruyten_index(xj, yj, g);
            if constexpr (NEED_W0) b0 = beta0[ir_ruy];
            if constexpr (NEED_WH) bh = betah[ir_ruy];
        }
        stage_with(act, xj, yj, zj, wj, ux, uy, uz, ig, c_light, g, b0, bh, my_kz, my_kr, my_nb);
    }
    // index of the particle's Ruyten coefficient (the one stage() reads): a kernel may fetch the
    // coefficients early and pass them to stage_with, so that the staging itself waits for no load
    __device__ __forceinline__ int ruyten_index(double xj, double yj, const DepGeom &g) const
    {
        const double rj = sqrt(xj * xj + yj * yj);
        const double r_cell = g.invdr * (rj - g.rmin) - 0.5;
        return min((int)ceil(r_cell), Nr);
    }
    __device__ __forceinline__ void stage_with(bool act, double xj, double yj, double zj, double wj,
            double ux, double uy, double uz, double ig, double c_light, const DepGeom &g,
            double beta0_v, double betah_v, int &my_kz, int &my_kr, int &my_nb)
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
            double Sz[S], Sr0[S], Srh[S];
            shape_z<SHAPE>(z_cell, Sz);
            if constexpr (NEED_W0) shape_r<SHAPE>(r_cell, beta0_v, Sr0);
            if constexpr (NEED_WH) shape_r<SHAPE>(r_cell, betah_v, Srh);
#pragma unroll
            for (int j = 0; j < S; j++) {
                Wl[j * DEP_PAD + lane] = Sz[j];
                if constexpr (NEED_W0) Wl[(S + j) * DEP_PAD + lane] = Sr0[j];
                if constexpr (NEED_WH) Wl[(2 * S + j) * DEP_PAD + lane] = Srh[j];
            }
            // number of stencil columns below the axis: index + (icr - H) < 0
            my_nb = H - icr;
        } else {
            // tail of the stream: finite amplitudes AND shape factors for the (masked) matrix
            // operands - the weight of a masked particle is (its Sz) x 0, and an Sz row that was
            // never written holds whatever the LDS held before the launch (a NaN pattern there
            // would reach the sums of the last, partial chunk)
#pragma unroll
            for (int a = 0; a < L::NA; a++) Al[a * DEP_PAD + lane] = 0.;
#pragma unroll
            for (int a = 0; a < NW; a++) Wl[a * DEP_PAD + lane] = 0.;
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

    // ---- phase 2 for a stream that was cell-sorted some steps ago (cycle.hip): the runs are those
    // of the HOME cells (cell of every particle at the last sort, contiguous by construction:
    // `runstarts` = lanes where it changes), and only the particles that still deposit into the
    // stencil of their home cell (`homem`) take part in them - a particle that has left its cell
    // does not break the run of its neighbours, it is deposited on its own by scatter_one.
    // hkz, hkr, hnb: stencil key of the lane's home cell (lane-varying, uniform inside a run).
    __device__ __forceinline__ void reduce_home(int cnt, unsigned long long runstarts,
                                                unsigned long long homem, int hkz, int hkr, int hnb)
    {
        int p = 0;
        while (p < cnt) {
            const unsigned long long rest = (p + 1 < 64) ? (runstarts >> (p + 1)) : 0ull;
            int e = rest ? p + 1 + __builtin_ctzll(rest) : cnt;
            if (e > cnt) e = cnt;
            const unsigned long long span = ((e - p) >= 64 ? ~0ull : ((1ull << (e - p)) - 1ull)) << p;
            if (homem & span) {
                const int nz_ = __builtin_amdgcn_readlane(hkz, p);
                const int nr_ = __builtin_amdgcn_readlane(hkr, p);
                if (nz_ == cur_z && nr_ == cur_r) {
                    // the run of the previous chunk goes on
                } else if (nz_ == cur_z && nr_ == cur_r + 1) {
                    flush(true);                     // column cur_r is complete
                    const bool carry = (((kl & (S - 1)) - off) & (S - 1)) != 0;
#pragma unroll
                    for (int rg = 0; rg < NAR; rg++)
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[rg][t] = carry ? acc[rg][t] : 0.;
                    off = (off + 1) & (S - 1);
                    cur_r = nr_;
                    cur_nb = __builtin_amdgcn_readlane(hnb, p);
                } else {
                    flush(false);
#pragma unroll
                    for (int rg = 0; rg < NAR; rg++)
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[rg][t] = 0.;
                    off = 0;
                    cur_z = nz_;
                    cur_r = nr_;
                    cur_nb = __builtin_amdgcn_readlane(hnb, p);
                }
                product(acc, p, e, off, homem);
            }
            p = e;
        }
    }

    // Direct deposition of ONE staged particle pl (wave-uniform) with stencil key (kz, kr, nb):
    // lane = (node, amplitude) as in a flush, value = (Sz * Sr) * amplitude from the staged
    // rows, one atomic instruction per NF - no accumulation, no cross-lane reduction.
    __device__ __forceinline__ void scatter_one(int pl, int kz, int kr, int nb)
    {
        double vv[NF];
#pragma unroll
        for (int qq = 0; qq < NF; qq++) {
            // non-CB layout: block bl of round qq writes tile 4 qq + bl; every lane of that
            // block holds the (node kl, column jl) value.  CB: lane (bl, kl) = node, tile qq.
            vv[qq] = (Wl[s_wz[qq] + pl] * Wl[s_wr[qq] + pl]) * Al[s_am[qq] + pl];
        }
        flush_values(vv, kz, kr, nb, 0, false);
    }
    // all particles of `straym` (bit = staged particle), keys read from their lanes
    __device__ __forceinline__ void scatter_strays(unsigned long long straym, int my_kz, int my_kr, int my_nb)
    {
        while (straym) {
            const int l = __builtin_ctzll(straym);
            straym &= straym - 1ull;
            scatter_one(l, __builtin_amdgcn_readlane(my_kz, l), __builtin_amdgcn_readlane(my_kr, l),
                        __builtin_amdgcn_readlane(my_nb, l));
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

}  // namespace fb
