// The r-spectral part of a PIC step in ONE launch (single z-periodic domain, standard PSATD):
//   forward Hankel transform of the freshly deposited J and rho_next (divide-by-volume, (r,t) ->
//   (p,m) and the spectral filter riding along)          fields.py:313-368, hankel.py:182-243
//   -> curl-free current correction, PSATD push of E, B, rho_next -> rho_prev
//                                                         fields/numba_methods.py:63-185, 364-404
//   -> inverse Hankel transform of the new E, B          fields.py:370-429
// i.e. fb_hankel_rt_to_pm_scaled + fb_psatd_step_standard + fb_hankel(E, B), which at the
// headline size (1024 x 128, Nm = 2) are three dependent, under-filled launches of ~25 us each:
// every transform there is one "generation" of workgroups whose first loads and last stores
// overlap no MFMA work, and the spectral slab makes a full round trip between them.
//
// Here a workgroup owns 8 kz rows of ONE azimuthal mode (the solver does not couple modes) and
// all Nr columns; its 8 waves split the OUTPUT columns (16 each).  The PSATD update is local in
// (kz, kr), so after the forward products every lane holds J and rho_next of its own cells in its
// accumulators, updates them with E, B, rho_prev read once from the spectral slab, writes the
// slab, and passes the new E, B to the inverse products through LDS (the K dimension of the
// inverse transform runs over kr: all eight waves need all of it).  1024 x 128, Nm = 2: 256
// workgroups = one per CU, two waves per SIMD, 320 v_mfma_f64_16x16x4 per wave back to back.
// Round 5: the fields that share a Hankel matrix (Jz | rho; E and B of one component) are multiplied
// against one stream of its fragments, and the memory operations are ordered for the in-order
// vmcnt counter (see k_spect_cycle): 56 -> 50 us inside a step.  Nr <= 128 on purpose: every
// workgroup streams every matrix once, which only pays where a transform is one under-filled
// generation of workgroups.
//
// MFMA fragments (cdna_hip_programming.md section 3): A lane l -> A[i = l & 15][k = l >> 4],
// B lane l -> B[k = l >> 4][j = l & 15], D reg r of lane l -> D[i = (l >> 4) + 4 r][j = l & 15].
// Rows i of a tile = (kz_local, re | im): i = kz_local + 8 ri, so that a lane's four D registers
// are re / im of the SAME two cells (kz_local = l >> 4 and + 4): complete complex numbers in one
// lane, no shuffle before the cell-local update.
#include "fb_common.h"
#ifndef SC_KNOCK
#define SC_KNOCK 0                 // timing experiments (tools/sc_time.py): 1 .. 3 drop one part each
#endif

namespace fb {

#ifdef SC_TRACE
// timing experiment (tools/sc_trace.py): shader-clock stamps of wave 0 of every workgroup at the phase
// boundaries of k_spect_cycle<false>
__device__ unsigned long long sc_trace_buf[4096 * 8];
#define SC_STAMP(i) do { if (tid == 0) sc_trace_buf[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SC_STAMP(i) do { } while (0)
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cplx sc_add(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx sc_sub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx sc_rmul(double s, cplx a) { return {s * a.re, s * a.im}; }
__device__ __forceinline__ cplx sc_imul(cplx a) { return {-a.im, a.re}; }
__device__ __forceinline__ cplx sc_ld(const cplx *p) { double2 v = *(const double2 *)p; return {v.x, v.y}; }
__device__ __forceinline__ void sc_st(cplx *p, cplx v) { *(double2 *)p = make_double2(v.re, v.im); }
__device__ __forceinline__ void sc_stu(cplx *p, cplx v) { if (SC_KNOCK != 3) sc_st(p, v); }

constexpr int SC_TZ = 8;            // kz rows per workgroup
constexpr int SC_KMAX = 128;        // Nr <= 128
constexpr int SC_RS = SC_KMAX + 2;  // panel row stride in doubles: the 32 (row, k) pairs of half a
                                    // wave's ds_read_b64 fall on 32 different bank pairs
constexpr int SC_PANEL = 16 * SC_RS;
// (4 / 8 / 16 / 32 steps: 50.8 / 51.4 / 52.8 / 61.3 us when the launch is repeated back to back,
// tools/sc_time.py - but INSIDE a step, where the particle kernels have flushed the matrices out
// of L2: 64.8 / 57.9 / 56.7 / 65.7 us.  The depth is chosen in the bench, not in the loop.)
#ifndef SC_PF_STEPS
#define SC_PF_STEPS 16
#endif
constexpr int SC_PF = SC_PF_STEPS;   // matrix row groups requested ahead of the MFMA that uses them

struct SpectCycleArgs {
    // per mode m: src[4m..] = Jr, Jt, Jz, rho after the forward z-FFT (un-normalised)
    const cplx *src[4 * FB_MAX_MODES];
    const double *invvol[FB_MAX_MODES];
    const double *fwd[3 * FB_MAX_MODES];       // Hankel matrices of p (order m+1), m (m-1), 0 (m)
    const double *inv[3 * FB_MAX_MODES];       // their inverses
    const double *fz[FB_MAX_MODES], *fr[FB_MAX_MODES];     // spectral filter (both or none)
    cplx *f[11 * FB_MAX_MODES];                // Ep Em Ez Bp Bm Bz Jp Jm Jz rho_prev rho_next
    const double *t[8 * FB_MAX_MODES];         // rho_prev_coef rho_next_coef j_coef C S_w kr kz inv_k2
    cplx *out[6 * FB_MAX_MODES];               // E, B (p, m, z) in (kz, r) space
    long irs, srs, ors;                        // row strides of src, the spectral slab, out
    double dt, inv_dt, c2, eps0, mu0;
    int correct, use_true_rho, Nz, Nr;
};

// The matrix operand of a wave: column n0 + li of rows 4 s + lk, s = 0 .. K4 - 1, one 128-B row
// segment per quarter wave, straight from L2 (each wave reads a different column slice: nothing to
// share through LDS).  The stream of all products of the kernel is requested SC_PF steps ahead -
// across the products too: while product j runs its last steps, the first rows of product j + 1's
// matrix are already on their way (b carries over from call to call).
// Every load is unconditional (a select between the loaded value and 0 makes the compiler wait for
// the load it has just issued - measured: 130 us instead of 70 for the whole kernel): rows beyond
// Nr are clamped to the last row, where the A panel holds zeros (0 x finite = 0); columns beyond Nr
// are clamped too, their sums are never stored.
struct ScStream {
    double b[SC_PF];
};

__device__ __forceinline__ void sc_prime(ScStream &B, const double *__restrict__ mat, int Nr, int n0, int li, int lk)
{
    const int cc = min(n0 + li, Nr - 1);
#pragma unroll
    for (int p = 0; p < SC_PF; p++) B.b[p] = mat[(long)min(4 * p + lk, Nr - 1) * Nr + cc];
}

// acc[a] = A_a (16 rows x K, LDS panel `panel + a * pstride`) . M[:, this wave's 16 columns], a < NA:
// the NA fields that share a Hankel matrix (spectral_transformer.py:67-69: Jz | rho forward, E and B
// of the same (p | m | z) component backward) are multiplied against ONE stream of its fragments -
// round 4 streamed 10 matrices per workgroup for 6 different ones, each 512-B fragment feeding a
// single MFMA.  `next`: the matrix of the product that follows (its first SC_PF steps are
// requested here), or null.
template <int NA>
__device__ __forceinline__ void sc_product(const double *__restrict__ panel, int pstride,
                                           const double *__restrict__ mat, const double *__restrict__ next,
                                           ScStream &B, int Nr, int K4, int n0, int li, int lk,
                                           double4_t (&acc)[NA])
{
#pragma unroll
    for (int a = 0; a < NA; a++) acc[a] = (double4_t){0., 0., 0., 0.};
    const int cc = min(n0 + li, Nr - 1);
    const double *Arow = panel + li * SC_RS + lk;
    if (next == nullptr) next = mat;                  // (redundant loads at the very end)
    // (K4 is a multiple of SC_PF - the panel is zero beyond Nr - so that the body is straight-line
    // code: a conditional step makes every refill a copy behind an s_waitcnt vmcnt(0))
    for (int s0 = 0; s0 < K4; s0 += SC_PF) {
        // rows requested in this round: s0 + SC_PF ... of this matrix, or the first ones of the next
        const bool wrap = s0 + SC_PF >= K4;               // wave-uniform
        const double *src = wrap ? next : mat;
        const int sb = wrap ? 0 : s0 + SC_PF;
#pragma unroll
        for (int p = 0; p < SC_PF; p++) {
            double a[NA];
#pragma unroll
            for (int q = 0; q < NA; q++) a[q] = Arow[q * pstride + 4 * (s0 + p)];
            const double x = B.b[p];
#if SC_KNOCK != 1            // (1: timing experiment without the matrix stream)
            B.b[p] = src[(long)min(4 * (sb + p) + lk, Nr - 1) * Nr + cc];
#endif
#pragma unroll
            for (int q = 0; q < NA; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], x, acc[q], 0, 0, 0);
#ifdef SC_SGB
            // keep the step's shape in the schedule: operand reads, ONE matrix request, the MFMAs
            // (left alone the scheduler gathers the 16 requests of a round behind its MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NA, 0);
#endif
        }
    }
}

// Order of the kernel's memory operations (round 5).  Loads, stores and atomics of a wave retire
// IN ORDER through one counter (vmcnt): a matrix fragment requested after a batch of other
// accesses cannot be used before that batch has completed.  Round 4 requested the 30 values of the
// cell-local update (E, B, rho_prev and the tables of the lane's two cells: 46 MB over the launch)
// in front of the sources, so the forward products started only when 63 MB had arrived, and issued
// all 22 stores of the update in front of the inverse products, whose 17th step then waited for
// 46 MB to be written: the four phases ran one after the other (~14 + 8 + 12 + 11 us of the 58).
// Now the sources - the only thing the forward products need - go first; the update's operands
// are requested in three batches at the start of the three forward products (their first SC_PF
// matrix fragments are already in registers, so the batch gets that many MFMA steps to land and
// the transfer overlaps the products); J and rho are stored by the update itself, the new E, B
// in two batches at the start of the second and third inverse product.
//
// ONLY_CORRECT (fb_spect_cycle_standard with correct_currents = 2): the launch ends behind the
// curl-free correction - forward transforms + correction, J and rho_next stored, nothing else
// touched.  For z-decomposed runs, where the guard cells of the CORRECTED J are added between the
// correction and the push (main.py:530-542): the transform and the correction were two launches
// with a round trip of J, rho through the spectral slab between them.
template <bool ONLY_CORRECT>
__global__ __launch_bounds__(512) void k_spect_cycle(SpectCycleArgs A)
{
    constexpr int NTHREADS = 512;
    extern __shared__ double sc_lds[];
    const int m = blockIdx.y;
    const int zb = blockIdx.x * SC_TZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int Nz = A.Nz, Nr = A.Nr;
    const int K4 = (Nr + 4 * SC_PF - 1) / (4 * SC_PF) * SC_PF;     // MFMA steps over K (multiple of SC_PF)
    const int n0 = 16 * wave;
    SC_STAMP(0);

    // ---- sources -> LDS panels [field][i = kz_local + 8 ri][k = r]: p, m, z, rho
    // (p = (r - i t) / 2, m = (r + i t) / 2, each times 1 / volume: spectral_transformer.py:208-210
    // and the divide-by-volume pass, as in k_hankel<SCALED, PAIRED>)
    {
        const cplx *sr = A.src[4 * m], *st = A.src[4 * m + 1], *sz = A.src[4 * m + 2], *sq = A.src[4 * m + 3];
        const double *iv = A.invvol[m];
        constexpr int NE = SC_TZ * SC_KMAX / NTHREADS;
        cplx r_[NE], t_[NE], z_[NE], q_[NE];
        double s_[NE];
#pragma unroll
        for (int u = 0; u < NE; u++) {
            const int e = tid + u * NTHREADS;
            const int row = e >> 7, k = e & (SC_KMAX - 1);
            const long o = (long)min(zb + row, Nz - 1) * A.irs + min(k, Nr - 1);
            r_[u] = sc_ld(sr + o); t_[u] = sc_ld(st + o); z_[u] = sc_ld(sz + o); q_[u] = sc_ld(sq + o);
            s_[u] = iv[min(k, Nr - 1)];
        }
#pragma unroll
        for (int u = 0; u < NE; u++) {
            const int e = tid + u * NTHREADS;
            const int row = e >> 7, k = e & (SC_KMAX - 1);
            const bool in = (zb + row < Nz) && (k < Nr);
            const double s = in ? s_[u] : 0.;             // zeros outside the grid
            double *P0 = sc_lds + row * SC_RS + k;
            P0[0 * SC_PANEL] = 0.5 * (r_[u].re + t_[u].im) * s;  P0[0 * SC_PANEL + 8 * SC_RS] = 0.5 * (r_[u].im - t_[u].re) * s;
            P0[1 * SC_PANEL] = 0.5 * (r_[u].re - t_[u].im) * s;  P0[1 * SC_PANEL + 8 * SC_RS] = 0.5 * (r_[u].im + t_[u].re) * s;
            P0[2 * SC_PANEL] = z_[u].re * s;  P0[2 * SC_PANEL + 8 * SC_RS] = z_[u].im * s;
            P0[3 * SC_PANEL] = q_[u].re * s;  P0[3 * SC_PANEL + 8 * SC_RS] = q_[u].im * s;
        }
    }
    ScStream B;
    sc_prime(B, A.fwd[3 * m + 0], Nr, n0, li, lk);
    __syncthreads();
    SC_STAMP(1);

    // ---- this lane's 2 cells (kz = zb + lk + 4 h, kr = n0 + li): E, B, rho_prev and the tables,
    // requested in three batches in front of the three forward products.  Cells outside the grid
    // read a clamped address; their results are never stored.
    cplx *const *f = A.f + 11 * m;
    const double *const *tb = A.t + 8 * m;
    const double *fz = A.fz[m], *fr = A.fr[m];
    const int nc = min(n0 + li, Nr - 1);
    long co[2], ci[2];
    int zc[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        zc[h] = min(zb + lk + 4 * h, Nz - 1);
        co[h] = (long)zc[h] * A.srs + nc; ci[h] = (long)zc[h] * Nr + nc;
    }
    cplx c_f[2][7];                    // Ep Em Ez Bp Bm Bz rho_prev
    double c_t[2][8], c_cz[2];

    // ---- forward products: Jp, Jm (matrices of p, m), then Jz | rho against the matrix of order m
    double4_t aJ[4];
    {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if constexpr (!ONLY_CORRECT) {
#pragma unroll
                for (int j = 0; j < 6; j++) c_f[h][j] = sc_ld(f[j] + co[h]);
            }
            c_f[h][6] = sc_ld(f[9] + co[h]);
        }
        double4_t a1[1];
        sc_product<1>(sc_lds + 0 * SC_PANEL, SC_PANEL, A.fwd[3 * m + 0], A.fwd[3 * m + 1], B, Nr, K4, n0, li, lk, a1);
        aJ[0] = a1[0];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = (ONLY_CORRECT ? 5 : 0); j < 8; j++) c_t[h][j] = tb[j][ci[h]];
        sc_product<1>(sc_lds + 1 * SC_PANEL, SC_PANEL, A.fwd[3 * m + 1], A.fwd[3 * m + 2], B, Nr, K4, n0, li, lk, a1);
        aJ[1] = a1[0];
        // filter factor of the sources (fz[iz] * fr[ir] * F as in numba_filter_*, k_hankel epilogue)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            c_cz[h] = 1.0;
            if (fr) { const double cn = 1.0 * fr[nc]; c_cz[h] = fz[zc[h]] * cn; }
        }
        double4_t a2[2];
        // (the first rows of the first inverse matrix travel during the cell-local update)
        sc_product<2>(sc_lds + 2 * SC_PANEL, SC_PANEL, A.fwd[3 * m + 2], ONLY_CORRECT ? nullptr : A.inv[3 * m + 0], B,
                      Nr, K4, n0, li, lk, a2);
        aJ[2] = a2[0]; aJ[3] = a2[1];
    }
    SC_STAMP(2);
    if constexpr (ONLY_CORRECT) {
        // numba_correct_currents_curlfree_standard (fields/numba_methods.py:63-85), expressions of
        // k_psatd_step; rho_next goes to its own field (the push that follows the J exchange shifts it)
        const int n = n0 + li;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int zz = zb + lk + 4 * h;
            if (zz < Nz && n < Nr) {
                const long o = co[h];
                const double cz = c_cz[h];
                cplx jp = {cz * aJ[0][h], cz * aJ[0][2 + h]};
                cplx jm = {cz * aJ[1][h], cz * aJ[1][2 + h]};
                cplx jz = {cz * aJ[2][h], cz * aJ[2][2 + h]};
                const cplx rn = {cz * aJ[3][h], cz * aJ[3][2 + h]};
                const double krr = c_t[h][5], kzz = c_t[h][6];
                const cplx rp = c_f[h][6];
                const cplx t1 = sc_rmul(A.inv_dt, sc_sub(rn, rp));
                const cplx t2 = sc_rmul(kzz, sc_imul(jz));
                const cplx t3 = sc_rmul(krr, sc_sub(jp, jm));
                const cplx F = sc_rmul(-c_t[h][7], sc_add(sc_add(t1, t2), t3));
                jp = sc_add(jp, sc_rmul(0.5 * krr, F));
                jm = sc_add(jm, sc_rmul(-0.5 * krr, F));
                jz = sc_add(jz, sc_rmul(kzz, sc_imul(sc_rmul(-1., F))));
                sc_st(f[6] + o, jp); sc_st(f[7] + o, jm); sc_st(f[8] + o, jz);
                sc_st(f[10] + o, rn);
            }
        }
        return;
    }
    __syncthreads();                 // the source panels are dead: the E, B panels take their place

    // ---- cell-local update of this lane's 2 cells
    const int n = n0 + li;
    cplx nE[2][6];                   // new Ep Em Ez Bp Bm Bz (stored during the inverse products)
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int row = lk + 4 * h, zz = zb + row;
        cplx ep = {0., 0.}, em = {0., 0.}, ez = {0., 0.}, bp = {0., 0.}, bm = {0., 0.}, bz = {0., 0.};
        live[h] = SC_KNOCK != 2 && zz < Nz && n < Nr;       // (2: timing experiment without the update)
        if (live[h]) {
            const long o = co[h];
            const double cz = c_cz[h];
            cplx jp = {cz * aJ[0][h], cz * aJ[0][2 + h]};
            cplx jm = {cz * aJ[1][h], cz * aJ[1][2 + h]};
            cplx jz = {cz * aJ[2][h], cz * aJ[2][2 + h]};
            const cplx rn = {cz * aJ[3][h], cz * aJ[3][2 + h]};
            const double rpc = c_t[h][0], rnc = c_t[h][1], jc = c_t[h][2], Cc = c_t[h][3], Sw = c_t[h][4];
            const double krr = c_t[h][5], kzz = c_t[h][6];
            ep = c_f[h][0]; em = c_f[h][1]; ez = c_f[h][2];
            bp = c_f[h][3]; bm = c_f[h][4]; bz = c_f[h][5];
            const cplx rp = c_f[h][6];
            // k_psatd_step (fields.hip), same expressions
            if (A.correct) {
                const cplx t1 = sc_rmul(A.inv_dt, sc_sub(rn, rp));
                const cplx t2 = sc_rmul(kzz, sc_imul(jz));
                const cplx t3 = sc_rmul(krr, sc_sub(jp, jm));
                const cplx F = sc_rmul(-c_t[h][7], sc_add(sc_add(t1, t2), t3));
                jp = sc_add(jp, sc_rmul(0.5 * krr, F));
                jm = sc_add(jm, sc_rmul(-0.5 * krr, F));
                jz = sc_add(jz, sc_rmul(kzz, sc_imul(sc_rmul(-1., F))));
            }
            sc_stu(f[6] + o, jp); sc_stu(f[7] + o, jm); sc_stu(f[8] + o, jz);
            cplx rho_diff;
            if (A.use_true_rho) {
                rho_diff = sc_sub(sc_rmul(rnc, rn), sc_rmul(rpc, rp));
            } else {
                const cplx divE = sc_add(sc_rmul(krr, sc_sub(ep, em)), sc_rmul(kzz, sc_imul(ez)));
                const cplx divJ = sc_add(sc_rmul(krr, sc_sub(jp, jm)), sc_rmul(kzz, sc_imul(jz)));
                rho_diff = sc_sub(sc_rmul((rnc - rpc) * A.eps0, divE), sc_rmul(rnc * A.dt, divJ));
            }
            const cplx mihkBz = sc_rmul(0.5 * krr, sc_imul(sc_rmul(-1., bz)));
            const cplx nep = sc_add(sc_add(sc_rmul(Cc, ep), sc_rmul(0.5 * krr, rho_diff)),
                    sc_rmul(A.c2 * Sw, sc_sub(sc_add(mihkBz, sc_rmul(kzz, bp)), sc_rmul(A.mu0, jp))));
            const cplx nem = sc_add(sc_sub(sc_rmul(Cc, em), sc_rmul(0.5 * krr, rho_diff)),
                    sc_rmul(A.c2 * Sw, sc_sub(sc_sub(mihkBz, sc_rmul(kzz, bm)), sc_rmul(A.mu0, jm))));
            const cplx nez = sc_add(sc_sub(sc_rmul(Cc, ez), sc_rmul(kzz, sc_imul(rho_diff))),
                    sc_rmul(A.c2 * Sw, sc_sub(sc_add(sc_rmul(krr, sc_imul(bp)), sc_rmul(krr, sc_imul(bm))),
                                              sc_rmul(A.mu0, jz))));
            const cplx mihkEz = sc_rmul(0.5 * krr, sc_imul(sc_rmul(-1., ez)));
            const cplx mihkJz = sc_rmul(0.5 * krr, sc_imul(sc_rmul(-1., jz)));
            const cplx nbp = sc_add(sc_sub(sc_rmul(Cc, bp), sc_rmul(Sw, sc_add(mihkEz, sc_rmul(kzz, ep)))),
                    sc_rmul(jc, sc_add(mihkJz, sc_rmul(kzz, jp))));
            const cplx nbm = sc_add(sc_sub(sc_rmul(Cc, bm), sc_rmul(Sw, sc_sub(mihkEz, sc_rmul(kzz, em)))),
                    sc_rmul(jc, sc_sub(mihkJz, sc_rmul(kzz, jm))));
            const cplx nbz = sc_add(sc_sub(sc_rmul(Cc, bz),
                                           sc_rmul(Sw, sc_add(sc_rmul(krr, sc_imul(ep)), sc_rmul(krr, sc_imul(em))))),
                    sc_rmul(jc, sc_add(sc_rmul(krr, sc_imul(jp)), sc_rmul(krr, sc_imul(jm)))));
            sc_stu(f[9] + o, rn);                       // push_rho: rho_prev <- rho_next
            sc_stu(f[10] + o, {0., 0.});
            ep = nep; em = nem; ez = nez; bp = nbp; bm = nbm; bz = nbz;
        }
        nE[h][0] = ep; nE[h][1] = em; nE[h][2] = ez; nE[h][3] = bp; nE[h][4] = bm; nE[h][5] = bz;
        // new E, B -> the A panels of the inverse products (zeros outside the grid); panel order
        // Ep Bp | Em Bm | Ez Bz: the two fields of a product lie next to each other
        if (n < SC_KMAX) {
            double *P0 = sc_lds + row * SC_RS + n;
            P0[0 * SC_PANEL] = ep.re; P0[0 * SC_PANEL + 8 * SC_RS] = ep.im;
            P0[1 * SC_PANEL] = bp.re; P0[1 * SC_PANEL + 8 * SC_RS] = bp.im;
            P0[2 * SC_PANEL] = em.re; P0[2 * SC_PANEL + 8 * SC_RS] = em.im;
            P0[3 * SC_PANEL] = bm.re; P0[3 * SC_PANEL + 8 * SC_RS] = bm.im;
            P0[4 * SC_PANEL] = ez.re; P0[4 * SC_PANEL + 8 * SC_RS] = ez.im;
            P0[5 * SC_PANEL] = bz.re; P0[5 * SC_PANEL + 8 * SC_RS] = bz.im;
        }
    }
    SC_STAMP(3);
    __syncthreads();
    SC_STAMP(4);

    // ---- inverse products, E and B of a component against one matrix stream, written to the
    // (kz, r) slab the backward z-FFT reads (out[6 m + j], j = Ep Em Ez Bp Bm Bz)
#pragma unroll
    for (int j = 0; j < 3; j++) {
        // the new spectral E, B of the update leave in two batches (see the top of the kernel)
        if (j == 1) {
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (live[h]) { sc_stu(f[0] + co[h], nE[h][0]); sc_stu(f[1] + co[h], nE[h][1]); sc_stu(f[2] + co[h], nE[h][2]); }
        }
        if (j == 2) {
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (live[h]) { sc_stu(f[3] + co[h], nE[h][3]); sc_stu(f[4] + co[h], nE[h][4]); sc_stu(f[5] + co[h], nE[h][5]); }
        }
        double4_t acc[2];
        sc_product<2>(sc_lds + 2 * j * SC_PANEL, SC_PANEL, A.inv[3 * m + j], j < 2 ? A.inv[3 * m + j + 1] : nullptr,
                      B, Nr, K4, n0, li, lk, acc);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            cplx *o_ = A.out[6 * m + j + 3 * q];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int zz = zb + lk + 4 * h;
                if (zz < Nz && n < Nr)
                    sc_st(o_ + (long)zz * A.ors + n, {1.0 * acc[q][h], 1.0 * acc[q][2 + h]});
            }
        }
        SC_STAMP(5 + j);
    }
}

}  // namespace fb

using namespace fb;

#ifdef SC_TRACE
extern "C" int fb_debug_sc_trace(unsigned long long *host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sc_trace_buf), (size_t)n * 8);
}
#endif

extern "C" int fb_spect_cycle_supported(int Nm, int Nr)
{
    return Nm >= 1 && Nm <= FB_MAX_MODES && Nr >= 1 && Nr <= SC_KMAX;
}

extern "C" int fb_spect_cycle_standard(int Nm, const void *const *src, long src_row_stride,
        const double *const *invvol, const double *const *fwd_mats, const double *const *inv_mats,
        const double *const *filter_z, const double *const *filter_r,
        void *const *fields, long spect_row_stride, const double *const *tables,
        double dt, int correct_currents, int use_true_rho, double c, double epsilon_0, double mu_0,
        void *const *out, long out_row_stride, int Nz, int Nr, void *stream)
{
    const char *who = "fb_spect_cycle_standard";
    if (!fb_spect_cycle_supported(Nm, Nr)) { set_error(who, "Nm <= 8 and Nr <= 128 (use the separate entry points)"); return -1; }
    SpectCycleArgs A;
    for (int i = 0; i < 4 * FB_MAX_MODES; i++) A.src[i] = i < 4 * Nm ? (const cplx *)src[i] : nullptr;
    for (int i = 0; i < FB_MAX_MODES; i++) {
        A.invvol[i] = i < Nm ? invvol[i] : nullptr;
        A.fz[i] = (i < Nm && filter_z) ? filter_z[i] : nullptr;
        A.fr[i] = (i < Nm && filter_r) ? filter_r[i] : nullptr;
        if ((A.fz[i] == nullptr) != (A.fr[i] == nullptr)) { set_error(who, "filter_z and filter_r go together"); return -1; }
    }
    if (correct_currents != 2 && (!inv_mats || !out)) { set_error(who, "inv_mats and out are required"); return -1; }
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) {
        A.fwd[i] = i < 3 * Nm ? fwd_mats[i] : nullptr;
        A.inv[i] = (i < 3 * Nm && inv_mats) ? inv_mats[i] : nullptr;
    }
    for (int i = 0; i < 11 * FB_MAX_MODES; i++) A.f[i] = i < 11 * Nm ? (cplx *)fields[i] : nullptr;
    for (int i = 0; i < 8 * FB_MAX_MODES; i++) A.t[i] = i < 8 * Nm ? tables[i] : nullptr;
    for (int i = 0; i < 6 * FB_MAX_MODES; i++) A.out[i] = (i < 6 * Nm && out) ? (cplx *)out[i] : nullptr;
    A.irs = src_row_stride; A.srs = spect_row_stride; A.ors = out_row_stride;
    A.dt = dt; A.inv_dt = 1. / dt; A.c2 = c * c; A.eps0 = epsilon_0; A.mu0 = mu_0;
    A.correct = correct_currents; A.use_true_rho = use_true_rho; A.Nz = Nz; A.Nr = Nr;
    const size_t lds_bytes = (size_t)6 * SC_PANEL * 8;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_spect_cycle<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return check(e, who);
        e = hipFuncSetAttribute((const void *)k_spect_cycle<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(4 * SC_PANEL * 8));
        if (e != hipSuccess) return check(e, who);
        attr_done = true;
    }
    dim3 grid((Nz + SC_TZ - 1) / SC_TZ, Nm);
    // 8 waves of one 16-column tile each (4 waves of two tiles: 58 against 52 us at C2, round 4)
    if (correct_currents == 2)
        hipLaunchKernelGGL(k_spect_cycle<true>, grid, dim3(512), (size_t)4 * SC_PANEL * 8, (hipStream_t)stream, A);
    else
        hipLaunchKernelGGL(k_spect_cycle<false>, grid, dim3(512), lds_bytes, (hipStream_t)stream, A);
    return check(hipGetLastError(), who);
}
