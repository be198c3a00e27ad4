// Deposition engine of the one-pass particle cycle (cycle.hip), round 5: J (from x(n+1/2)) and rho
// (from x(n+1)) of a chunk are staged TOGETHER and reduced in ONE traversal of the home runs.
//
// Why a second engine next to DepEngine (dep_engine.h, still used by deposit.hip and by the
// 64-bit-pointer form of the cycle kernel): SQ counters of round 4 put 547 of the 1381 VALU and 414
// of the 596 SALU instructions per 64 particles into the two run reductions - each run was walked
// twice (once per engine: the same readlanes, key compares and flush decisions), and every flush
// added the 4 MFMA blocks of every tile with DPP rotations (8 VALU per tile) before one lane group
// could write it.  Here
//   * the 4 blocks of v_mfma_f64_4x4x4_4b are the 4 WEIGHT VARIANTS of a node - block v: 0 = J with
//     the mode-0 Ruyten weights, 1 = J with the weights of the modes >= 1, 2 = rho / mode 0,
//     3 = rho / modes >= 1 - so one instruction multiplies W_v[4 nodes][4 particles] by the
//     amplitudes of variant v of the same 4 particles for all four variants at once, and every
//     lane ends with ONE FINISHED sum per tile: (node l >> 4, amplitude 4 tile + (l & 3), variant
//     (l >> 2) & 3).  No cross-block sum, no second traversal, one flush decision per run, and the
//     J and rho values of a node (one 128-B record of the in-step target) leave in the same atomic
//     instruction.
//   * the panel holds per particle and engine the FIRST shape factor of each direction only
//     (linear shape: the second one is 1 - first, formed by an fma on read): s = Sz[0], t0 / th =
//     Sr[0] with the Ruyten coefficient of mode 0 / of the modes >= 1, then the mode-0 amplitudes
//     and cos / sin of the modes >= 1 (the amplitude of mode m is their product, formed on read).
//     Nm = 2: 15 rows x 67 doubles = 7.9 KB per wave (DepEngine: 7.8 KB for J alone) - with the
//     gather panel 12 776 B, the most that lets 12 one-wave workgroups share a CU (CycleDepLayout).
//   * two cells that follow each other along r share a node column: its sums MOVE to the lanes of
//     the lower column (v_permlane16_swap: node = lane >> 4, the two columns are 16 lanes apart)
//     instead of the lanes changing role - no per-lane state depends on the run.
//   * a stray (a particle that has left the stencil of its home cell, per engine) is written out
//     directly as before (lane = node x amplitude x variant); its mode-0 amplitudes are then
//     zeroed in the panel, so the products need no per-particle mask - only the last step of a run
//     masks the particles of the next one.
// Measured (MI355X, C2): 1121 VALU / 461 SALU per 64 particles against
// 1323 / 542 for the two engines, 2-3 % per launch - the kernel is bound by the latency chain of a
// wave at 3 waves per SIMD, not by issue.
// Lane layout of v_mfma_f64_4x4x4_4b (tools/mfma4_probe.hip): A operand lane l = A[i = l & 3][k = l >> 4]
// of block (l >> 2) & 3, B operand lane l = B[k = l >> 4][j = l & 3] of the same block, D lane l =
// D[i = l >> 4][j = l & 3] of the same block.
//
// Per-particle arithmetic (cos, sin, mode recurrence, shape factors, Ruyten term, cell keys) is that of
// DepEngine::stage_with, i.e. fbpic/particles/deposition/threading_methods.py:27-305 and
// particle_shapes.py:17-41; guard folding and axis signs are those of DepEngine::flush_values
// (fbpic/fields/numba_methods.py:409-461).  Only the summation order differs.
#pragma once
#include "dep_engine.h"

namespace fb {

template <int NM> struct CycleDepLayout {
    // amplitudes of a variant: J m0 (Jr, Jt, Jz: real), J modes >= 1 (3 components x re / im per mode),
    // rho m0, rho modes >= 1
    static constexpr int NAJ0 = 3, NAJH = 6 * (NM - 1), NAR0 = 1, NARH = 2 * (NM - 1);
    static constexpr int NTR = (NM > 1) ? 2 : 1;           // radial factor rows per engine: t0 [, th]
    // Rows of an engine: s | t0 [th] | mode-0 amplitudes a0 (3 / 1) | cos m theta, sin m theta of the
    // modes m >= 1.  The amplitude of mode m is a0 (cos m theta + i sin m theta) (threading_methods.py:
    // 119-121, 264-267: the recurrence starts from a real a0), formed when the matrix operand is read
    // as X . Y = (a0 row) . (cos | sin row, or the row of ones for mode 0): 15 rows at Nm = 2 where the
    // products themselves take 19 - the 12.8 KB per wave (with the gather panel) that let 12 waves
    // share a CU; with 14.2 KB only 10 did (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES, profiles/README.md) and
    // the kernel was slower than round 4's in spite of 250 fewer VALU instructions per 64 particles.
    static constexpr int NCS = 2 * (NM - 1);
    static constexpr int ROW_ONE = 0;
    static constexpr int ROW_SJ = 1, ROW_TJ = 2, ROW_AJ = 2 + NTR, ROW_CJ = ROW_AJ + 3;
    static constexpr int ROW_SR = ROW_CJ + NCS, ROW_TR = ROW_SR + 1, ROW_AR = ROW_SR + 1 + NTR, ROW_CR = ROW_AR + 1;
    static constexpr int NROWS = ROW_CR + NCS;
    // row stride in doubles, ODD: a 64-lane ds_read_b64 is served in four passes of 16 lanes, and the
    // 16 lanes of a pass (one particle, up to 15 different rows) must fall on 16 different 8-byte slots
    // of the 128-B LDS word - slot = (row PAD + particle) mod 16.  With 66 the rows of the two engines
    // (8 rows apart) aliased: SQ_LDS_BANK_CONFLICT 312 cycles per 64 particles; 65: 12; 67: 0
    // (profiles/r05_sq_onepass.txt).  15 rows x 67 + the 4.7 KB gather panel = 12 776 B per wave: 12
    // one-wave workgroups share a CU up to 12 800 B, 11 from 12 896 B on (tools/lds_probe.hip).
#ifndef FB_CD_PAD
#define FB_CD_PAD 67
#endif
    static constexpr int PAD = FB_CD_PAD;
    // tiles of 4 amplitudes: the largest variant decides (J, modes >= 1)
    static constexpr int NTL = (NAJH > 4) ? (NAJH + 3) / 4 : 1;
    static constexpr int WAVE_DOUBLES = NROWS * PAD + 4;
    __host__ __device__ static constexpr int namp(int v) { return v == 0 ? NAJ0 : v == 1 ? NAJH : v == 2 ? NAR0 : NARH; }
    // rows X (mode-0 amplitude) and Y (cos / sin of the mode, or ones) behind amplitude a of variant v
    __host__ __device__ static constexpr int row_x(int v, int a)
    {
        return v == 0 ? ROW_AJ + a : v == 1 ? ROW_AJ + (a >> 1) % 3 : ROW_AR;
    }
    __host__ __device__ static constexpr int row_y(int v, int a)
    {
        // modes >= 1: amplitude a -> mode 1 + (a >> 1) / ncomp, re (cos) / im (sin)
        return v == 0 || v == 2 ? ROW_ONE
             : v == 1 ? ROW_CJ + 2 * ((a >> 1) / 3) + (a & 1) : ROW_CR + 2 * (a >> 1) + (a & 1);
    }
};

template <int NM>
struct CycleDep {
    using L = CycleDepLayout<NM>;
    static constexpr int NTL = L::NTL, PAD = L::PAD;

    char *P;                       // the wave's panel (LDS)
    char *gbase;                   // lowest address of the J and rho targets (all within 4 GiB of it)
    int lane;
    int rsB, csB, Nz, Nr;
    // A / B operand roles: byte offsets of this lane's rows, particle (l >> 4) of a step included
    int aS, aT, aX[NTL], aY[NTL];
    int k8, kA;                    // 8 (l >> 4); l >> 4
    double cz0, cz1, cr0, cr1;     // A role, node l & 3: Sz[jz] = cz0 + cz1 s, Sr[jr] = cr0 + cr1 t
    // D role: target of tile t as a byte offset from gbase (node row jzD included), validity and
    // below-axis sign as bit masks, node (jzD, jrD), engine
    unsigned f_off[NTL];
    unsigned valid, neg;
    int jzD, jrD, jrDB;
    bool engR, evenrow;
    double acc[NTL];
    int cur_z, cur_r, cur_nb;

    // aS, aT, aB are absolute LDS byte addresses (the panel base folded in once): a fragment read is
    // one v_add (+ 8 q) per row and immediate offsets for the steps of a group
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    __device__ __forceinline__ double ld(int addr) const { return *(lds_cdouble *)(unsigned long)(unsigned)addr; }

    __device__ __forceinline__ void init(double *panel, int lane_, const DepGrids &GJ, long rsJ, const DepGrids &GR,
                                         long rsR, int Nz_, int Nr_, cplx *gbase_)
    {
        P = (char *)panel;
        gbase = (char *)gbase_;
        lane = lane_;
        // (both targets are views of one record array: same strides - checked by the host)
        rsB = (int)(16 * rsJ); csB = (int)(16 * GJ.cs);
        Nz = Nz_; Nr = Nr_;
        const int v = (lane >> 2) & 3, e = v >> 1, vh = v & 1, j = lane & 3;
        kA = lane >> 4; k8 = 8 * kA;
        const int pl = (int)(unsigned long)(__attribute__((address_space(3))) char *)P + k8;
        aS = (e ? L::ROW_SR : L::ROW_SJ) * PAD * 8 + pl;
        aT = ((e ? L::ROW_TR : L::ROW_TJ) + ((L::NTR > 1) ? vh : 0)) * PAD * 8 + pl;
        const int na = L::namp(v);
        valid = 0u; neg = 0u;
        const int iD = lane >> 4;
        jzD = iD >> 1; jrD = iD & 1; jrDB = jrD * csB;
        engR = e != 0;
        evenrow = jrD == 0;
#pragma unroll
        for (int t = 0; t < NTL; t++) {
            const int a = 4 * t + j;
            const bool ok = a < na;
            const int ac = na > 0 ? (ok ? a : na - 1) : 0;       // (columns beyond the variant: any finite rows)
            aX[t] = (na > 0 ? L::row_x(v, ac) : L::ROW_ONE) * PAD * 8 + pl;
            aY[t] = (na > 0 ? L::row_y(v, ac) : L::ROW_ONE) * PAD * 8 + pl;
            // amplitude a of variant v -> component, mode, re / im (rows as DepLayout::row)
            int comp = 0, m = 0, ri = 0;
            if (ok) {
                if (v == 0) { comp = a; }
                else if (v == 1) { ri = a & 1; comp = (a >> 1) % 3; m = 1 + (a >> 1) / 3; }
                else if (v == 3) { ri = a & 1; m = 1 + (a >> 1); }
            }
            // (selected with static indices: a lane-dependent index into the by-value pointer tables
            // would send them to scratch memory)
            double *fp = nullptr;
#pragma unroll
            for (int i = 0; i < 3 * NM; i++) fp = (!e && i == comp + 3 * m) ? (double *)GJ.g[i] : fp;
#pragma unroll
            for (int i = 0; i < NM; i++) fp = (e && i == m) ? (double *)GR.g[i] : fp;
            fp += ri;
            f_off[t] = (unsigned)((char *)fp - gbase) + (unsigned)(jzD * rsB);
            valid |= ok ? (1u << t) : 0u;
            // rho, Jz: (-1)^m ; Jr, Jt: -(-1)^m (threading_methods.py:143-146, 289-302)
            const double flip = m1pow(m);
            neg |= (((e || comp == 2) ? flip : -flip) < 0.) ? (1u << t) : 0u;
            acc[t] = 0.;
        }
        // (opaque to the optimiser: it would otherwise split the constant row offsets off again and
        // re-add them, and the panel base, in front of every read)
        asm volatile("" : "+v"(aS), "+v"(aT));
#pragma unroll
        for (int t = 0; t < NTL; t++) asm volatile("" : "+v"(aX[t]), "+v"(aY[t]));
        const int iA = lane & 3, jzA = iA >> 1, jrA = iA & 1;
        cz0 = jzA ? 1. : 0.; cz1 = jzA ? -1. : 1.;
        cr0 = jrA ? 1. : 0.; cr1 = jrA ? -1. : 1.;
        cur_z = DEP_NOKEY; cur_r = DEP_NOKEY; cur_nb = 0;
        // The last step of a run reads up to 3 particles beyond it (masked weights, but the values
        // must be finite): the columns from 64 on of every row and the 4 doubles behind the last row are
        // never staged - zero them once.
        if (lane < L::NROWS) {
#pragma unroll
            for (int cpad = 64; cpad < PAD; cpad++) *(double *)(P + (lane * PAD + cpad) * 8) = 0.;
        }
        if (lane < 4) *(double *)(P + (L::NROWS * PAD + lane) * 8) = 0.;
        // the row of ones (Y of the mode-0 amplitudes): written once
        *(double *)(P + (L::ROW_ONE * PAD + lane) * 8) = 1.;
    }

    // ---- phase 1, lane = particle: amplitudes, first shape factors, stencil key of engine E
    // (0: J, 1: rho) - the arithmetic of DepEngine::stage_with.  u, ig, c_light only for J.
    // stage_pre: everything that does not read the Ruyten coefficients
    template <int E>
    __device__ __forceinline__ void stage_pre(double xj, double yj, double zj, double wj,
            double ux, double uy, double uz, double ig, double c_light, const DepGeom &g,
            int &my_kz, int &my_kr, int &my_nb, double &r_cell_out)
    {
        constexpr int NC = (E == 0) ? 3 : 1;
        constexpr int R0 = (E == 0) ? L::ROW_AJ : L::ROW_AR;           // mode-0 amplitudes (real)
        constexpr int RC = (E == 0) ? L::ROW_CJ : L::ROW_CR;           // cos m theta, sin m theta, m >= 1
        double *Pd = (double *)P;
        const double rj = sqrt(xj * xj + yj * yj);
        double cs_, sn;
        if (rj != 0.) {
            double r0 = __builtin_amdgcn_rcp(rj);
            r0 = __builtin_fma(r0, __builtin_fma(-rj, r0, 1.), r0);
            const double invr = __builtin_fma(r0, __builtin_fma(-rj, r0, 1.), r0);
            cs_ = xj * invr; sn = yj * invr;
        } else { cs_ = 1.; sn = 0.; }
        if constexpr (NC == 1) {
            Pd[R0 * PAD + lane] = wj;
        } else {
            Pd[(R0 + 0) * PAD + lane] = wj * c_light * ig * (cs_ * ux + sn * uy);
            Pd[(R0 + 1) * PAD + lane] = wj * c_light * ig * (cs_ * uy - sn * ux);
            Pd[(R0 + 2) * PAD + lane] = wj * c_light * ig * uz;
        }
        {
            double er = cs_, ei = sn;                      // (cos + i sin)^m by the reference's recurrence
#pragma unroll
            for (int mm = 1; mm < NM; mm++) {
                Pd[(RC + 2 * (mm - 1)) * PAD + lane] = er;
                Pd[(RC + 2 * (mm - 1) + 1) * PAD + lane] = ei;
                const double re = cs_ * er - sn * ei, im = cs_ * ei + sn * er;
                er = re; ei = im;
            }
        }
        const double r_cell = g.invdr * (rj - g.rmin) - 0.5;
        const double z_cell = g.invdz * (zj - g.zmin) - 0.5;
        const int icr = (int)ceil(r_cell), icz = (int)ceil(z_cell);
        my_kr = min(icr - 1, Nr); my_kz = icz - 1;
        my_nb = 1 - icr;
        double Sz[2];
        shape_z<FB_SHAPE_LINEAR>(z_cell, Sz);
        constexpr int RS = (E == 0) ? L::ROW_SJ : L::ROW_SR;
        Pd[RS * PAD + lane] = Sz[0];
        r_cell_out = r_cell;
    }
    // ... and the part that reads the Ruyten coefficients (the radial shape factors of mode 0 and of the modes
    // >= 1): the one-pass kernel requests the coefficients in front of stage_pre and waits for vector memory
    // between the two parts, so that the request travels during the rest of the staging (round 6)
    template <int E>
    __device__ __forceinline__ void stage_post(double r_cell, double beta0_v, double betah_v)
    {
        double *Pd = (double *)P;
        double Sr0[2], Srh[2];
        constexpr int RT = (E == 0) ? L::ROW_TJ : L::ROW_TR;
        shape_r<FB_SHAPE_LINEAR>(r_cell, beta0_v, Sr0);
        Pd[RT * PAD + lane] = Sr0[0];
        if constexpr (NM > 1) {
            shape_r<FB_SHAPE_LINEAR>(r_cell, betah_v, Srh);
            Pd[(RT + 1) * PAD + lane] = Srh[0];
        }
    }
    template <int E>
    __device__ __forceinline__ void stage(double xj, double yj, double zj, double wj,
            double ux, double uy, double uz, double ig, double c_light, const DepGeom &g,
            double beta0_v, double betah_v, int &my_kz, int &my_kr, int &my_nb)
    {
        double r_cell;
        stage_pre<E>(xj, yj, zj, wj, ux, uy, uz, ig, c_light, g, my_kz, my_kr, my_nb, r_cell);
        stage_post<E>(r_cell, beta0_v, betah_v);
    }

    // zero the amplitudes of the lanes (= particles) that take no part in the runs of an engine
    __device__ __forceinline__ void zero_amplitudes(bool notJ, bool notR)
    {
        double *Pd = (double *)P;
        if (notJ) {
#pragma unroll
            for (int a = 0; a < 3; a++) Pd[(L::ROW_AJ + a) * PAD + lane] = 0.;
        }
        if (notR) Pd[L::ROW_AR * PAD + lane] = 0.;
    }

    // ---- move the staged column of particle `lane` to column `pos` (a permutation of 0 .. 63), every row but
    // the row of ones: the chunk's particles re-ordered inside the panel (cycle.hip, regrouped chunks).  LDS
    // operations of a wave execute in issue order, so the reads of a batch return the old columns before its
    // writes land.
    __device__ __forceinline__ void permute_columns(int pos)
    {
        double *Pd = (double *)P;
#pragma unroll
        for (int r0 = 1; r0 < L::NROWS; r0 += 4) {
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) if (r0 + k < L::NROWS) v[k] = Pd[(r0 + k) * PAD + lane];
#pragma unroll
            for (int k = 0; k < 4; k++) if (r0 + k < L::NROWS) Pd[(r0 + k) * PAD + pos] = v[k];
        }
    }

    // ---- flush of the current cell: one atomic instruction per tile.  keep_upper: only its lower
    // node column (the upper one carries on as the lower column of the next cell)
    __device__ __forceinline__ void flush(bool keep_upper)
    {
        if (cur_z == DEP_NOKEY) return;
        const int cz = cur_z, cr = cur_r;
        const bool interior = cz >= 0 && cz + 2 <= Nz && cr >= 0 && cr + 2 <= Nr;
        const int cell_baseB = cz * rsB + cr * csB;       // wave-uniform: scalar arithmetic
#pragma unroll
        for (int t = 0; t < NTL; t++) {
            double v = acc[t];
            if (!((valid >> t) & 1u) || v == 0. || (keep_upper && !evenrow)) continue;
            unsigned voff;
            if (interior) {
                voff = f_off[t] + (unsigned)(cell_baseB + jrDB);
            } else {
                int gz = cz + jzD, gr = cr + jrD;
                fold_node(gz, gr, Nz, Nr);
                if (jrD < cur_nb && ((neg >> t) & 1u)) v = -v;      // node below the axis: signed fold
                voff = f_off[t] + (unsigned)((gz - jzD) * rsB + gr * csB);
            }
#if defined(FB_KNOCK_FLUSH_ATOM)
            asm volatile("" :: "v"(voff), "v"(v));           // (timing experiment: the flush without its atomics)
#else
            atomicAdd((double *)(gbase + voff), v);
#endif
        }
    }

    // value of the lane 16 further on (node column jr = 1 of the same node row) for the lanes of
    // column 0; v_permlane16_swap: vdst keeps its even rows of 16 and receives the even rows of src in
    // its odd rows, src the reverse (particles.hip, other_xor16)
    __device__ __forceinline__ double upper_column(double v) const
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return __hiloint2double(b[1], a[1]);
    }

    // ---- one group of NST steps of 4 particles, first particle q (byte offset q8 = 8 q), rem = e - q
    // particles of the run left.  MASK: the last step holds particles of the next run.
    template <int NST, bool MASK>
    __device__ __forceinline__ void group(int q8, int rem)
    {
        double s[NST], t[NST], b[NST][NTL];
        const int oS = aS + q8, oT = aT + q8;
        int oX[NTL], oY[NTL];
#pragma unroll
        for (int tl = 0; tl < NTL; tl++) { oX[tl] = aX[tl] + q8; oY[tl] = aY[tl] + q8; }
#pragma unroll
        for (int u = 0; u < NST; u++) {
            s[u] = ld(oS + 32 * u);
            t[u] = ld(oT + 32 * u);
#pragma unroll
            for (int tl = 0; tl < NTL; tl++) b[u][tl] = ld(oX[tl] + 32 * u) * ld(oY[tl] + 32 * u);
        }
#pragma unroll
        for (int u = 0; u < NST; u++) {
            double w = __builtin_fma(s[u], cz1, cz0) * __builtin_fma(t[u], cr1, cr0);
            if (MASK && u == NST - 1) w = (kA < rem - 4 * u) ? w : 0.;
#pragma unroll
            for (int tl = 0; tl < NTL; tl++) acc[tl] = __builtin_amdgcn_mfma_f64_4x4x4f64(w, b[u][tl], acc[tl], 0, 0, 0);
        }
    }
    // acc += W . amplitudes over the staged particles [p, e) (one run)
    __device__ __forceinline__ void product(int p, int e)
    {
#ifndef FB_CD_GROUP
#define FB_CD_GROUP 4
#endif
        for (int q = p; q < e; q += 4 * FB_CD_GROUP) {
            const int rem = e - q;
            if (rem >= 4 * FB_CD_GROUP) group<FB_CD_GROUP, false>(8 * q, rem);
#if FB_CD_GROUP >= 4
            else if (rem > 12) group<4, true>(8 * q, rem);
#endif
#if FB_CD_GROUP >= 3
            else if (rem > 8) group<3, true>(8 * q, rem);
#endif
#if FB_CD_GROUP >= 2
            else if (rem > 4) group<2, true>(8 * q, rem);
#endif
            else group<1, true>(8 * q, rem);
        }
    }

    // ---- phase 2: ONE traversal of the home runs for both engines.  runstarts: lanes where the home
    // cell changes; homem: particles that take part in a run of either engine (the amplitudes of the
    // others are zero in the panel); hkz, hkr, hnb: stencil key of the lane's home cell.
    __device__ __forceinline__ void reduce(int cnt, unsigned long long runstarts, unsigned long long homem,
                                           int hkz, int hkr, int hnb)
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
#pragma unroll
                    for (int t = 0; t < NTL; t++) {
                        const double up = upper_column(acc[t]);
                        acc[t] = evenrow ? up : 0.;
                    }
                    cur_r = nr_;
                    cur_nb = __builtin_amdgcn_readlane(hnb, p);
                } else {
                    flush(false);
#pragma unroll
                    for (int t = 0; t < NTL; t++) acc[t] = 0.;
                    cur_z = nz_;
                    cur_r = nr_;
                    cur_nb = __builtin_amdgcn_readlane(hnb, p);
                }
                product(p, e);
            }
            p = e;
        }
    }

    // ---- a chunk regrouped by (J cell, rho cell) PAIRS (cycle.hip): every run of the sorted order holds
    // particles with one J stencil and one rho stencil - which may differ (a particle that crosses a cell
    // boundary within the second half push) -, so every particle takes part in a run of BOTH engines and
    // nothing is scattered one by one.  Each run is flushed on its own (no sliding columns: the two
    // engines' cells need not move together), the J tiles to the J cell, the rho tiles to the rho cell;
    // guard folding and axis signs per lane as in flush().
    __device__ __forceinline__ void flush_pair(int zJ, int rJ, int nJ, int zR, int rR, int nR)
    {
        const int cz = engR ? zR : zJ, cr = engR ? rR : rJ, nb = engR ? nR : nJ;
#pragma unroll
        for (int t = 0; t < NTL; t++) {
            double v = acc[t];
            if (!((valid >> t) & 1u) || v == 0.) continue;
            int gz = cz + jzD, gr = cr + jrD;
            fold_node(gz, gr, Nz, Nr);
            if (jrD < nb && ((neg >> t) & 1u)) v = -v;
            const unsigned voff = f_off[t] + (unsigned)((gz - jzD) * rsB + gr * csB);
            atomicAdd((double *)(gbase + voff), v);
        }
    }
    __device__ __forceinline__ void reduce_pairs(int cnt, unsigned long long starts, int jz, int jr, int jn,
                                                 int rz, int rr, int rn)
    {
        flush(false);                        // the run that was open when the chunk began
        cur_z = DEP_NOKEY; cur_r = DEP_NOKEY; cur_nb = 0;
        int p = 0;
        while (p < cnt) {
            const unsigned long long rest = (p + 1 < 64) ? (starts >> (p + 1)) : 0ull;
            int e = rest ? p + 1 + __builtin_ctzll(rest) : cnt;
            if (e > cnt) e = cnt;
#pragma unroll
            for (int t = 0; t < NTL; t++) acc[t] = 0.;
            product(p, e);
            flush_pair(__builtin_amdgcn_readlane(jz, p), __builtin_amdgcn_readlane(jr, p),
                       __builtin_amdgcn_readlane(jn, p), __builtin_amdgcn_readlane(rz, p),
                       __builtin_amdgcn_readlane(rr, p), __builtin_amdgcn_readlane(rn, p));
            p = e;
        }
#pragma unroll
        for (int t = 0; t < NTL; t++) acc[t] = 0.;
    }

    // ---- a chunk in which the stencils of the two engines differ for MANY particles (a laser wake:
    // fast particles change cell within the half push between the two depositions): two traversals,
    // each engine on the runs of ITS OWN keys, the other engine's mode-0 amplitudes zero meanwhile -
    // every particle takes part in a run, nothing is scattered one by one.  (The flush of a run writes
    // only the non-zero sums, i.e. those of the engine that is being traversed.)
    __device__ __forceinline__ void reduce_split(int cnt, bool act, int jkz, int jkr, int jnb,
                                                 int rkz, int rkr, int rnb)
    {
        double *Pd = (double *)P;
        const unsigned long long actm = __ballot(act);
        double a0[3];
#pragma unroll
        for (int a = 0; a < 3; a++) { a0[a] = Pd[(L::ROW_AJ + a) * PAD + lane]; Pd[(L::ROW_AJ + a) * PAD + lane] = 0.; }
        wave_lds_release();
        {
            const int pz = __shfl_up(rkz, 1), pr = __shfl_up(rkr, 1);
            reduce(cnt, __ballot(act && (lane == 0 || rkz != pz || rkr != pr)), actm, rkz, rkr, rnb);
        }
        wave_lds_acquire();
#pragma unroll
        for (int a = 0; a < 3; a++) Pd[(L::ROW_AJ + a) * PAD + lane] = a0[a];
        Pd[L::ROW_AR * PAD + lane] = 0.;
        wave_lds_release();
        {
            const int pz = __shfl_up(jkz, 1), pr = __shfl_up(jkr, 1);
            reduce(cnt, __ballot(act && (lane == 0 || jkz != pz || jkr != pr)), actm, jkz, jkr, jnb);
        }
    }

    // ---- strays: staged particle l (wave-uniform) written out directly - lane = (node l >> 4,
    // amplitude, variant), value = (Sz Sr) amplitude from the staged rows, one atomic instruction per
    // tile for BOTH engines where the particle is a stray of both (keys per engine).
    __device__ __forceinline__ void scatter_strays(unsigned long long smJ, unsigned long long smR,
            int jkz, int jkr, int jnb, int rkz, int rkr, int rnb)
    {
        unsigned long long um = smJ | smR;
        while (um) {
            const int l = __builtin_ctzll(um);
            um &= um - 1ull;
            const bool inJ = (smJ >> l) & 1ull, inR = (smR >> l) & 1ull;
            const int zJ = __builtin_amdgcn_readlane(jkz, l), rJ = __builtin_amdgcn_readlane(jkr, l);
            const int nJ = __builtin_amdgcn_readlane(jnb, l);
            const int zR = __builtin_amdgcn_readlane(rkz, l), rR = __builtin_amdgcn_readlane(rkr, l);
            const int nR = __builtin_amdgcn_readlane(rnb, l);
            const bool mine = engR ? inR : inJ;
            const int so = 8 * l - k8;
            // (D role, node l >> 4: the second factor of a direction is 1 - first)
            const double s_ = ld(aS + so), t_ = ld(aT + so);
            const double wz = jzD ? 1. - s_ : s_;
            const double wr = jrD ? 1. - t_ : t_;
            const double w = wz * wr;
            const bool intJ = zJ >= 0 && zJ + 2 <= Nz && rJ >= 0 && rJ + 2 <= Nr;
            const bool intR = zR >= 0 && zR + 2 <= Nz && rR >= 0 && rR + 2 <= Nr;
            const bool interior = (intJ || !inJ) && (intR || !inR);
            const int baseJ = zJ * rsB + rJ * csB, baseR = zR * rsB + rR * csB;
#pragma unroll
            for (int t = 0; t < NTL; t++) {
                double v = w * (ld(aX[t] + so) * ld(aY[t] + so));
                if (!mine || !((valid >> t) & 1u) || v == 0.) continue;
                unsigned voff;
                if (interior) {
                    voff = f_off[t] + (unsigned)((engR ? baseR : baseJ) + jrDB);
                } else {
                    const int kz = engR ? zR : zJ, kr = engR ? rR : rJ, nb = engR ? nR : nJ;
                    int gz = kz + jzD, gr = kr + jrD;
                    fold_node(gz, gr, Nz, Nr);
                    if (jrD < nb && ((neg >> t) & 1u)) v = -v;
                    voff = f_off[t] + (unsigned)((gz - jzD) * rsB + gr * csB);
                }
#if defined(FB_KNOCK_SCAT_ATOM)
                asm volatile("" :: "v"(voff), "v"(v));       // (timing experiment: strays without their atomics)
#else
                atomicAdd((double *)(gbase + voff), v);
#endif
            }
        }
    }
};

}  // namespace fb
