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
// inverse transform runs over kr: all four waves need all of it).  1024 x 128, Nm = 2: 256
// workgroups = one per CU, two waves per SIMD, 320 v_mfma_f64_16x16x4 per wave back to back.
//
// MFMA fragments (cdna_hip_programming.md section 3): A lane l -> A[i = l & 15][k = l >> 4],
// B lane l -> B[k = l >> 4][j = l & 15], D reg r of lane l -> D[i = (l >> 4) + 4 r][j = l & 15].
// Rows i of a tile = (kz_local, re | im): i = kz_local + 8 ri, so that a lane's four D registers
// are re / im of the SAME two cells (kz_local = l >> 4 and + 4): complete complex numbers in one
// lane, no shuffle before the cell-local update.
#include "fb_common.h"
#ifndef SC_KNOCK
#define SC_KNOCK 0                 // timing experiments (tools/sc_time.py): 1 .. 4 drop one part each
#endif

namespace fb {

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
constexpr int SC_PF = 16;            // matrix row groups requested ahead of the MFMA that uses them

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

// The matrix operand of a wave: column n0 + 16 t + li (t = 0, 1) of rows 4 s + lk, s = 0 .. K4 - 1,
// one 128-B row segment per quarter wave, straight from L2 (each wave reads a different column
// slice: nothing to share through LDS).  The stream of all products of the kernel is requested
// SC_PF steps ahead - across the products too: while product j runs its last steps, the first rows
// of product j + 1's matrix are already on their way (b0 / b1 carry over from call to call).
// Every load is unconditional (a select between the loaded value and 0 makes the compiler wait for
// the load it has just issued - measured: 130 us instead of 70 for the whole kernel): rows beyond
// Nr are clamped to the last row, where the A panel holds zeros (0 x finite = 0); columns beyond Nr
// are clamped too, their sums are never stored.
// NT: 16-column tiles per wave (a workgroup has 8 / NT waves)
template <int NT> struct ScStream {
    double b[NT][SC_PF];
};

template <int NT>
__device__ __forceinline__ void sc_prime(ScStream<NT> &B, const double *__restrict__ mat, int Nr, int n0, int li, int lk)
{
#pragma unroll
    for (int p = 0; p < SC_PF; p++) {
        const long ro = (long)min(4 * p + lk, Nr - 1) * Nr;
#pragma unroll
        for (int t = 0; t < NT; t++) B.b[t][p] = mat[ro + min(n0 + 16 * t + li, Nr - 1)];
    }
}

// acc = A (16 rows x K, LDS panel) . M[:, this wave's 32 columns]; `next`: the matrix of the product
// that follows (its first SC_PF steps are requested here), or null
template <int NT>
__device__ __forceinline__ void sc_product(const double *__restrict__ panel, const double *__restrict__ mat,
                                           const double *__restrict__ next, ScStream<NT> &B,
                                           int Nr, int K4, int n0, int li, int lk, double4_t (&acc)[NT])
{
    int cc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        acc[t] = (double4_t){0., 0., 0., 0.};
        cc[t] = min(n0 + 16 * t + li, Nr - 1);
    }
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
            const double a = Arow[4 * (s0 + p)];
            double x[NT];
            const long ro = (long)min(4 * (sb + p) + lk, Nr - 1) * Nr;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                x[t] = B.b[t][p];
#if SC_KNOCK != 1            // (1: timing experiment without the matrix stream)
                B.b[t][p] = src[ro + cc[t]];
#endif
            }
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[t], acc[t], 0, 0, 0);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(512 / NT) void k_spect_cycle(SpectCycleArgs A)
{
    constexpr int NTHREADS = 512 / NT;
    extern __shared__ double sc_lds[];
    const int m = blockIdx.y;
    const int zb = blockIdx.x * SC_TZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int Nz = A.Nz, Nr = A.Nr;
    const int K4 = (Nr + 4 * SC_PF - 1) / (4 * SC_PF) * SC_PF;     // MFMA steps over K (multiple of SC_PF)
    const int n0 = 16 * NT * wave;

    ScStream<NT> B;
    sc_prime(B, A.fwd[3 * m + 0], Nr, n0, li, lk);     // (in flight while the panels are filled)

    // ---- E, B, rho_prev and the coefficient tables of this lane's 4 cells (kz = zb + lk + 4 h,
    // kr = n0 + 16 t + li): requested NOW, used after the forward products (with one wave per SIMD
    // nothing else would hide their latency: 58 -> 33 us without the update, measured).  Cells
    // outside the grid read a clamped address; their results are never stored.
    cplx *const *f = A.f + 11 * m;
    const double *const *tb = A.t + 8 * m;
    const double *fz = A.fz[m], *fr = A.fr[m];
    cplx c_f[2 * NT][7];               // Ep Em Ez Bp Bm Bz rho_prev
    double c_t[2 * NT][8], c_cz[2 * NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = 2 * t + h;
            const int zc = min(zb + lk + 4 * h, Nz - 1), nc = min(n0 + 16 * t + li, Nr - 1);
            const long o = (long)zc * A.srs + nc, idx = (long)zc * Nr + nc;
#pragma unroll
            for (int j = 0; j < 6; j++) c_f[q][j] = sc_ld(f[j] + o);
            c_f[q][6] = sc_ld(f[9] + o);
#pragma unroll
            for (int j = 0; j < 8; j++) c_t[q][j] = tb[j][idx];
            // filter factor of the sources (fz[iz] * fr[ir] * F as in numba_filter_*, k_hankel epilogue)
            c_cz[q] = 1.0;
            if (fr) { const double cn = 1.0 * fr[nc]; c_cz[q] = fz[zc] * cn; }
        }

    // ---- sources -> LDS panels [field][i = kz_local + 8 ri][k = r]: p, m, z, rho
    // (p = (r - i t) / 2, m = (r + i t) / 2, each times 1 / volume: spectral_transformer.py:208-210
    // and the divide-by-volume pass, as in k_hankel<SCALED, PAIRED>)
    {
        const cplx *sr = A.src[4 * m], *st = A.src[4 * m + 1], *sz = A.src[4 * m + 2], *sq = A.src[4 * m + 3];
        const double *iv = A.invvol[m];
        for (int e = tid; e < SC_TZ * SC_KMAX; e += NTHREADS) {
            const int row = e >> 7, k = e & (SC_KMAX - 1);
            const int zz = zb + row;
            cplx p = {0., 0.}, mm = {0., 0.}, z = {0., 0.}, q = {0., 0.};
            if (zz < Nz && k < Nr) {
                const long o = (long)zz * A.irs + k;
                const cplx r_ = sc_ld(sr + o), t_ = sc_ld(st + o);
                const double s_ = iv[k];
                p = {0.5 * (r_.re + t_.im) * s_, 0.5 * (r_.im - t_.re) * s_};
                mm = {0.5 * (r_.re - t_.im) * s_, 0.5 * (r_.im + t_.re) * s_};
                z = sc_ld(sz + o); z = {z.re * s_, z.im * s_};
                q = sc_ld(sq + o); q = {q.re * s_, q.im * s_};
            }
            double *P0 = sc_lds + row * SC_RS + k;
            P0[0 * SC_PANEL] = p.re;  P0[0 * SC_PANEL + 8 * SC_RS] = p.im;
            P0[1 * SC_PANEL] = mm.re; P0[1 * SC_PANEL + 8 * SC_RS] = mm.im;
            P0[2 * SC_PANEL] = z.re;  P0[2 * SC_PANEL + 8 * SC_RS] = z.im;
            P0[3 * SC_PANEL] = q.re;  P0[3 * SC_PANEL + 8 * SC_RS] = q.im;
        }
    }
    __syncthreads();

    // ---- forward products: Jp, Jm (matrices of p, m), Jz, rho (matrix of order m)
    double4_t aJ[4][NT];
    sc_product(sc_lds + 0 * SC_PANEL, A.fwd[3 * m + 0], A.fwd[3 * m + 1], B, Nr, K4, n0, li, lk, aJ[0]);
    sc_product(sc_lds + 1 * SC_PANEL, A.fwd[3 * m + 1], A.fwd[3 * m + 2], B, Nr, K4, n0, li, lk, aJ[1]);
    sc_product(sc_lds + 2 * SC_PANEL, A.fwd[3 * m + 2], A.fwd[3 * m + 2], B, Nr, K4, n0, li, lk, aJ[2]);
    // (the first rows of the first inverse matrix travel during the cell-local update)
    sc_product(sc_lds + 3 * SC_PANEL, A.fwd[3 * m + 2], A.inv[3 * m + 0], B, Nr, K4, n0, li, lk, aJ[3]);
    __syncthreads();                 // the source panels are dead: the E, B panels take their place

    // ---- cell-local update of this lane's 2 NT cells
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + 16 * t + li;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int row = lk + 4 * h, zz = zb + row;
            cplx ep = {0., 0.}, em = {0., 0.}, ez = {0., 0.}, bp = {0., 0.}, bm = {0., 0.}, bz = {0., 0.};
            if (SC_KNOCK != 2 && zz < Nz && n < Nr) {       // (2: timing experiment without the update)
                const int q = 2 * t + h;
                const long o = (long)zz * A.srs + n;
                const double cz = c_cz[q];
                cplx jp = {cz * aJ[0][t][h], cz * aJ[0][t][2 + h]};
                cplx jm = {cz * aJ[1][t][h], cz * aJ[1][t][2 + h]};
                cplx jz = {cz * aJ[2][t][h], cz * aJ[2][t][2 + h]};
                const cplx rn = {cz * aJ[3][t][h], cz * aJ[3][t][2 + h]};
                const double rpc = c_t[q][0], rnc = c_t[q][1], jc = c_t[q][2], Cc = c_t[q][3], Sw = c_t[q][4];
                const double krr = c_t[q][5], kzz = c_t[q][6];
                ep = c_f[q][0]; em = c_f[q][1]; ez = c_f[q][2];
                bp = c_f[q][3]; bm = c_f[q][4]; bz = c_f[q][5];
                const cplx rp = c_f[q][6];
                // k_psatd_step (fields.hip), same expressions
                if (A.correct) {
                    const cplx t1 = sc_rmul(A.inv_dt, sc_sub(rn, rp));
                    const cplx t2 = sc_rmul(kzz, sc_imul(jz));
                    const cplx t3 = sc_rmul(krr, sc_sub(jp, jm));
                    const cplx F = sc_rmul(-c_t[q][7], sc_add(sc_add(t1, t2), t3));
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
                sc_stu(f[0] + o, nep); sc_stu(f[1] + o, nem); sc_stu(f[2] + o, nez);
                sc_stu(f[3] + o, nbp); sc_stu(f[4] + o, nbm); sc_stu(f[5] + o, nbz);
                sc_stu(f[9] + o, rn);                       // push_rho: rho_prev <- rho_next
                sc_stu(f[10] + o, {0., 0.});
                ep = nep; em = nem; ez = nez; bp = nbp; bm = nbm; bz = nbz;
            }
            // new E, B -> the A panels of the inverse products (zeros outside the grid)
            if (n < SC_KMAX) {
                double *P0 = sc_lds + row * SC_RS + n;
                P0[0 * SC_PANEL] = ep.re; P0[0 * SC_PANEL + 8 * SC_RS] = ep.im;
                P0[1 * SC_PANEL] = em.re; P0[1 * SC_PANEL + 8 * SC_RS] = em.im;
                P0[2 * SC_PANEL] = ez.re; P0[2 * SC_PANEL + 8 * SC_RS] = ez.im;
                P0[3 * SC_PANEL] = bp.re; P0[3 * SC_PANEL + 8 * SC_RS] = bp.im;
                P0[4 * SC_PANEL] = bm.re; P0[4 * SC_PANEL + 8 * SC_RS] = bm.im;
                P0[5 * SC_PANEL] = bz.re; P0[5 * SC_PANEL + 8 * SC_RS] = bz.im;
            }
        }
    }
    #if SC_KNOCK == 4
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif

    // ---- inverse products, written to the (kz, r) slab the backward z-FFT reads
#pragma unroll 1
    for (int j = 0; j < 6; j++) {
        double4_t acc[NT];
        sc_product(sc_lds + j * SC_PANEL, A.inv[3 * m + (j % 3)], j < 5 ? A.inv[3 * m + ((j + 1) % 3)] : nullptr,
                   B, Nr, K4, n0, li, lk, acc);
        cplx *o_ = A.out[6 * m + j];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int n = n0 + 16 * t + li;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int zz = zb + lk + 4 * h;
                if (zz < Nz && n < Nr)
                    sc_st(o_ + (long)zz * A.ors + n, {1.0 * acc[t][h], 1.0 * acc[t][2 + h]});
            }
        }
    }
}

}  // namespace fb

using namespace fb;

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
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) {
        A.fwd[i] = i < 3 * Nm ? fwd_mats[i] : nullptr;
        A.inv[i] = i < 3 * Nm ? inv_mats[i] : nullptr;
    }
    for (int i = 0; i < 11 * FB_MAX_MODES; i++) A.f[i] = i < 11 * Nm ? (cplx *)fields[i] : nullptr;
    for (int i = 0; i < 8 * FB_MAX_MODES; i++) A.t[i] = i < 8 * Nm ? tables[i] : nullptr;
    for (int i = 0; i < 6 * FB_MAX_MODES; i++) A.out[i] = i < 6 * Nm ? (cplx *)out[i] : nullptr;
    A.irs = src_row_stride; A.srs = spect_row_stride; A.ors = out_row_stride;
    A.dt = dt; A.inv_dt = 1. / dt; A.c2 = c * c; A.eps0 = epsilon_0; A.mu0 = mu_0;
    A.correct = correct_currents; A.use_true_rho = use_true_rho; A.Nz = Nz; A.Nr = Nr;
    const size_t lds_bytes = (size_t)6 * SC_PANEL * 8;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_spect_cycle<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return check(e, who);
        attr_done = true;
    }
    dim3 grid((Nz + SC_TZ - 1) / SC_TZ, Nm);
    // 8 waves of one 16-column tile each (NT = 2, 4 waves of two tiles: 58 against 52 us at C2)
    hipLaunchKernelGGL(k_spect_cycle<1>, grid, dim3(512), lds_bytes, (hipStream_t)stream, A);
    return check(hipGetLastError(), who);
}
