// Grid kernels for gfx950 (interpolation grid + spectral grid), all HBM/cache-bound
// element-wise passes over complex128 (Nz, Nr) views with an explicit row stride so the
// same kernels run on the reference's contiguous arrays and on z-major field slabs.
// One lane = one complex element (16-byte accesses, r contiguous -> coalesced).
#include "fb_common.h"

namespace fb {

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx rmul(double s, cplx a) { return {s * a.re, s * a.im}; }
__device__ __forceinline__ cplx imul(cplx a) { return {-a.im, a.re}; }   // i * a
__device__ __forceinline__ cplx ld(const cplx *p) { double2 v = *(const double2 *)p; return {v.x, v.y}; }
__device__ __forceinline__ void st(cplx *p, cplx v) { *(double2 *)p = make_double2(v.re, v.im); }
// Non-temporal loads for the solver step of a BIG grid (its slab + tables exceed the 256 MB Infinity Cache: every
// operand is read once per step, and what the step WRITES - the new E, B - is what the next launch reads; see
// FB_NT_LD in fb_common.h).  Small grids keep plain loads: their tables and fields live in the cache from step to step.
typedef double fld_v2d __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ cplx ldx(const cplx *p)
{
#ifndef FB_NO_NT
    if constexpr (NT) { const fld_v2d v = __builtin_nontemporal_load((const fld_v2d *)p); return {v.x, v.y}; }
#endif
    return ld(p);
}
template <bool NT> __device__ __forceinline__ double ldx(const double *p)
{
#ifndef FB_NO_NT
    if constexpr (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}

#define FB_GRID_LOOP(idx, iz, ir)                                                     \
    const long ncell_ = (long)Nz * Nr;                                                \
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < ncell_;        \
         idx += (long)gridDim.x * blockDim.x)                                         \
        for (int iz = (int)(idx / Nr), ir = (int)(idx - (long)iz * Nr), once_ = 1; once_; once_ = 0)

// fields/cuda_methods.py:18-66
__global__ __launch_bounds__(256) void k_erase(int nf, Ptrs48 P, long rs, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        for (int f = 0; f < nf; f++) st((cplx *)P.p[f] + (long)iz * rs + ir, {0., 0.});
    }
}

// fields/cuda_methods.py:68-118 ; CPU: F *= invvol[newaxis, :]
__global__ __launch_bounds__(256) void k_divide(int nf, Ptrs48 P, long rs,
                                                const double *__restrict__ invvol, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const double v = invvol[ir];
        for (int f = 0; f < nf; f++) {
            cplx *p = (cplx *)P.p[f] + (long)iz * rs + ir;
            cplx a = ld(p);
            st(p, {a.re * v, a.im * v});
        }
    }
}

// fields/numba_methods.py:14-60 : field = fz[iz]*fr[ir]*field
__global__ __launch_bounds__(256) void k_filter(int nf, Ptrs48 P, long rs,
        const double *__restrict__ fz, const double *__restrict__ fr, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const double f_ = fz[iz] * fr[ir];
        for (int f = 0; f < nf; f++) {
            cplx *p = (cplx *)P.p[f] + (long)iz * rs + ir;
            cplx a = ld(p);
            st(p, {f_ * a.re, f_ * a.im});
        }
    }
}

__global__ __launch_bounds__(256) void k_scale(int nf, Ptrs48 P, long rs, double s, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        for (int f = 0; f < nf; f++) {
            cplx *p = (cplx *)P.p[f] + (long)iz * rs + ir;
            cplx a = ld(p);
            st(p, {a.re * s, a.im * s});
        }
    }
}

// fields/spectral_transform/numba_methods.py:60-103 (in-place safe: reads before writes)
__global__ __launch_bounds__(256) void k_rt_to_pm(int np, CPtrs48 R, CPtrs48 T, Ptrs48 Pp, Ptrs48 Pm,
                                                  long rs, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        for (int f = 0; f < np; f++) {
            cplx vr = ld((const cplx *)R.p[f] + o), vt = ld((const cplx *)T.p[f] + o);
            st((cplx *)Pp.p[f] + o, {0.5 * (vr.re + vt.im), 0.5 * (vr.im - vt.re)});
            st((cplx *)Pm.p[f] + o, {0.5 * (vr.re - vt.im), 0.5 * (vr.im + vt.re)});
        }
    }
}
__global__ __launch_bounds__(256) void k_pm_to_rt(int np, CPtrs48 Pp, CPtrs48 Pm, Ptrs48 R, Ptrs48 T,
                                                  long rs, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        for (int f = 0; f < np; f++) {
            cplx vp = ld((const cplx *)Pp.p[f] + o), vm = ld((const cplx *)Pm.p[f] + o);
            st((cplx *)R.p[f] + o, {vp.re + vm.re, vp.im + vm.im});
            const double dre = vp.re - vm.re, dim = vp.im - vm.im;
            st((cplx *)T.p[f] + o, {-dim, dre});
        }
    }
}

// fields/numba_methods.py:63-85
__global__ __launch_bounds__(256) void k_correct_currents(const cplx *__restrict__ rho_prev,
        const cplx *__restrict__ rho_next, cplx *__restrict__ Jp, cplx *__restrict__ Jm,
        cplx *__restrict__ Jz, long rs, const double *__restrict__ kz,
        const double *__restrict__ kr, const double *__restrict__ inv_k2, double inv_dt,
        int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        const double kzz = kz[idx], krr = kr[idx];
        cplx jp = ld(Jp + o), jm = ld(Jm + o), jz = ld(Jz + o);
        cplx t1 = rmul(inv_dt, csub(ld(rho_next + o), ld(rho_prev + o)));
        cplx t2 = rmul(kzz, imul(jz));
        cplx t3 = rmul(krr, csub(jp, jm));
        cplx F = rmul(-inv_k2[idx], cadd(cadd(t1, t2), t3));
        st(Jp + o, cadd(jp, rmul(0.5 * krr, F)));
        st(Jm + o, cadd(jm, rmul(-0.5 * krr, F)));
        st(Jz + o, cadd(jz, rmul(kzz, imul(rmul(-1., F)))));
    }
}

// fields/numba_methods.py:118-185
__global__ __launch_bounds__(256) void k_push_eb(cplx *__restrict__ Ep, cplx *__restrict__ Em,
        cplx *__restrict__ Ez, cplx *__restrict__ Bp, cplx *__restrict__ Bm, cplx *__restrict__ Bz,
        const cplx *__restrict__ Jp, const cplx *__restrict__ Jm, const cplx *__restrict__ Jz,
        const cplx *__restrict__ rho_prev, const cplx *__restrict__ rho_next, long rs,
        const double *__restrict__ rho_prev_coef, const double *__restrict__ rho_next_coef,
        const double *__restrict__ j_coef, const double *__restrict__ C,
        const double *__restrict__ S_w, const double *__restrict__ kr,
        const double *__restrict__ kz, double dt, int use_true_rho,
        double c2, double eps0, double mu0, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        const double krr = kr[idx], kzz = kz[idx], Cc = C[idx], Sw = S_w[idx], jc = j_coef[idx];
        const double rnc = rho_next_coef[idx], rpc = rho_prev_coef[idx];
        const cplx ep = ld(Ep + o), em = ld(Em + o), ez = ld(Ez + o);
        const cplx bp = ld(Bp + o), bm = ld(Bm + o), bz = ld(Bz + o);
        const cplx jp = ld(Jp + o), jm = ld(Jm + o), jz = ld(Jz + o);
        cplx rho_diff;
        if (use_true_rho) {
            rho_diff = csub(rmul(rnc, ld(rho_next + o)), rmul(rpc, ld(rho_prev + o)));
        } else {
            cplx divE = cadd(rmul(krr, csub(ep, em)), rmul(kzz, imul(ez)));
            cplx divJ = cadd(rmul(krr, csub(jp, jm)), rmul(kzz, imul(jz)));
            rho_diff = csub(rmul((rnc - rpc) * eps0, divE), rmul(rnc * dt, divJ));
        }
        const cplx mihkBz = rmul(0.5 * krr, imul(rmul(-1., bz)));
        st(Ep + o, cadd(cadd(rmul(Cc, ep), rmul(0.5 * krr, rho_diff)),
                        rmul(c2 * Sw, csub(cadd(mihkBz, rmul(kzz, bp)), rmul(mu0, jp)))));
        st(Em + o, cadd(csub(rmul(Cc, em), rmul(0.5 * krr, rho_diff)),
                        rmul(c2 * Sw, csub(csub(mihkBz, rmul(kzz, bm)), rmul(mu0, jm)))));
        st(Ez + o, cadd(csub(rmul(Cc, ez), rmul(kzz, imul(rho_diff))),
                        rmul(c2 * Sw, csub(cadd(rmul(krr, imul(bp)), rmul(krr, imul(bm))),
                                           rmul(mu0, jz)))));
        const cplx mihkEz = rmul(0.5 * krr, imul(rmul(-1., ez)));
        const cplx mihkJz = rmul(0.5 * krr, imul(rmul(-1., jz)));
        st(Bp + o, cadd(csub(rmul(Cc, bp), rmul(Sw, cadd(mihkEz, rmul(kzz, ep)))),
                        rmul(jc, cadd(mihkJz, rmul(kzz, jp)))));
        st(Bm + o, cadd(csub(rmul(Cc, bm), rmul(Sw, csub(mihkEz, rmul(kzz, em)))),
                        rmul(jc, csub(mihkJz, rmul(kzz, jm)))));
        st(Bz + o, cadd(csub(rmul(Cc, bz), rmul(Sw, cadd(rmul(krr, imul(ep)), rmul(krr, imul(em))))),
                        rmul(jc, cadd(rmul(krr, imul(jp)), rmul(krr, imul(jm))))));
    }
}

// Fused spectral step for ALL modes in one launch (single-domain fast path):
// curl-free current correction (numba_methods.py:63-85) -> PSATD push of E,B
// (:118-185) -> rho_prev <- rho_next, rho_next <- 0 (spectral_grid.py:407-421).
// All three are cell-local, so the corrected J never leaves registers.  Per mode:
// 11 field pointers in SpectralGrid order and 8 table pointers.
struct PsatdModes {
    cplx *f[11 * FB_MAX_MODES];
    const double *t[8 * FB_MAX_MODES];   // rho_prev_coef, rho_next_coef, j_coef, C, S_w, kr, kz, inv_k2
};

// shift != null and n_move != 0: the moving window's translation of the grid by n_move cells
// (fb_shift_spect: E, B, rho_prev and J times shift[iz]^n_move, moving_window.py:176-239) is
// applied to the values this kernel writes anyway - one sweep over the spectral slab less per
// step of a moving-window run.
template <bool NT>
__global__ __launch_bounds__(256) void k_psatd_step(PsatdModes M, long rs, double dt, double inv_dt,
        int correct, int use_true_rho, double c2, double eps0, double mu0, int Nz, int Nr,
        const cplx *__restrict__ shift, int n_move)
{
    const int m = blockIdx.y;
    cplx *const *f = M.f + 11 * m;
    const double *const *t = M.t + 8 * m;
    const bool moving = shift != nullptr && n_move != 0;
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        cplx pw = {1., 0.};
        if (moving) {
            const cplx sh = ld(shift + iz);
            const int na = n_move < 0 ? -n_move : n_move;
            for (int i = 0; i < na; i++) pw = {pw.re * sh.re - pw.im * sh.im, pw.re * sh.im + pw.im * sh.re};
            if (n_move < 0) pw.im = -pw.im;
        }
        auto stw = [&](cplx *p, cplx a) {      // store, translated when the window moves
            if (moving) a = {a.re * pw.re - a.im * pw.im, a.re * pw.im + a.im * pw.re};
            st(p, a);
        };
        const double rpc = ldx<NT>(t[0] + idx), rnc = ldx<NT>(t[1] + idx), jc = ldx<NT>(t[2] + idx), Cc = ldx<NT>(t[3] + idx), Sw = ldx<NT>(t[4] + idx);
        const double krr = ldx<NT>(t[5] + idx), kzz = ldx<NT>(t[6] + idx);
        const cplx ep = ldx<NT>(f[0] + o), em = ldx<NT>(f[1] + o), ez = ldx<NT>(f[2] + o);
        const cplx bp = ldx<NT>(f[3] + o), bm = ldx<NT>(f[4] + o), bz = ldx<NT>(f[5] + o);
        cplx jp = ldx<NT>(f[6] + o), jm = ldx<NT>(f[7] + o), jz = ldx<NT>(f[8] + o);
        const cplx rp = ldx<NT>(f[9] + o), rn = ldx<NT>(f[10] + o);
        if (correct) {
            cplx t1 = rmul(inv_dt, csub(rn, rp));
            cplx t2 = rmul(kzz, imul(jz));
            cplx t3 = rmul(krr, csub(jp, jm));
            cplx F = rmul(-ldx<NT>(t[7] + idx), cadd(cadd(t1, t2), t3));
            jp = cadd(jp, rmul(0.5 * krr, F));
            jm = cadd(jm, rmul(-0.5 * krr, F));
            jz = cadd(jz, rmul(kzz, imul(rmul(-1., F))));
            if (correct == 2) {                  // correction only (the J guard exchange follows)
                st(f[6] + o, jp); st(f[7] + o, jm); st(f[8] + o, jz);
                continue;
            }
            stw(f[6] + o, jp); stw(f[7] + o, jm); stw(f[8] + o, jz);
        } else if (moving) {
            stw(f[6] + o, jp); stw(f[7] + o, jm); stw(f[8] + o, jz);
        }
        cplx rho_diff;
        if (use_true_rho) {
            rho_diff = csub(rmul(rnc, rn), rmul(rpc, rp));
        } else {
            cplx divE = cadd(rmul(krr, csub(ep, em)), rmul(kzz, imul(ez)));
            cplx divJ = cadd(rmul(krr, csub(jp, jm)), rmul(kzz, imul(jz)));
            rho_diff = csub(rmul((rnc - rpc) * eps0, divE), rmul(rnc * dt, divJ));
        }
        const cplx mihkBz = rmul(0.5 * krr, imul(rmul(-1., bz)));
        stw(f[0] + o, cadd(cadd(rmul(Cc, ep), rmul(0.5 * krr, rho_diff)),
                          rmul(c2 * Sw, csub(cadd(mihkBz, rmul(kzz, bp)), rmul(mu0, jp)))));
        stw(f[1] + o, cadd(csub(rmul(Cc, em), rmul(0.5 * krr, rho_diff)),
                          rmul(c2 * Sw, csub(csub(mihkBz, rmul(kzz, bm)), rmul(mu0, jm)))));
        stw(f[2] + o, cadd(csub(rmul(Cc, ez), rmul(kzz, imul(rho_diff))),
                          rmul(c2 * Sw, csub(cadd(rmul(krr, imul(bp)), rmul(krr, imul(bm))),
                                             rmul(mu0, jz)))));
        const cplx mihkEz = rmul(0.5 * krr, imul(rmul(-1., ez)));
        const cplx mihkJz = rmul(0.5 * krr, imul(rmul(-1., jz)));
        stw(f[3] + o, cadd(csub(rmul(Cc, bp), rmul(Sw, cadd(mihkEz, rmul(kzz, ep)))),
                          rmul(jc, cadd(mihkJz, rmul(kzz, jp)))));
        stw(f[4] + o, cadd(csub(rmul(Cc, bm), rmul(Sw, csub(mihkEz, rmul(kzz, em)))),
                          rmul(jc, csub(mihkJz, rmul(kzz, jm)))));
        stw(f[5] + o, cadd(csub(rmul(Cc, bz), rmul(Sw, cadd(rmul(krr, imul(ep)), rmul(krr, imul(em))))),
                          rmul(jc, cadd(rmul(krr, imul(jp)), rmul(krr, imul(jm))))));
        stw(f[9] + o, rn);
        st(f[10] + o, {0., 0.});
    }
}

// ---- Galilean / comoving-current PSATD (fields/numba_methods.py:217-241, 278-355) --------
// Same cell-local structure as the standard scheme; the Theta coefficients T_eb, T_cc, T_rho,
// the corrected-current coefficient and the three source coefficients are complex tables
// (contiguous (Nz,Nr) complex128), C, S_w, kr, kz, inv_k2 real.
__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

__global__ __launch_bounds__(256) void k_correct_currents_comoving(const cplx *__restrict__ rho_prev,
        const cplx *__restrict__ rho_next, cplx *__restrict__ Jp, cplx *__restrict__ Jm,
        cplx *__restrict__ Jz, long rs, const double *__restrict__ kz, const double *__restrict__ kr,
        const double *__restrict__ inv_k2, const cplx *__restrict__ j_corr_coef,
        const cplx *__restrict__ T_eb, const cplx *__restrict__ T_cc, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        const double kzz = kz[idx], krr = kr[idx];
        const cplx jp = ld(Jp + o), jm = ld(Jm + o), jz = ld(Jz + o);
        const cplx t1 = cmul(cmul(ld(T_cc + idx), ld(j_corr_coef + idx)),
                             csub(ld(rho_next + o), cmul(ld(rho_prev + o), ld(T_eb + idx))));
        const cplx t2 = rmul(kzz, imul(jz));
        const cplx t3 = rmul(krr, csub(jp, jm));
        const cplx F = rmul(-inv_k2[idx], cadd(cadd(t1, t2), t3));
        st(Jp + o, cadd(jp, rmul(0.5 * krr, F)));
        st(Jm + o, cadd(jm, rmul(-0.5 * krr, F)));
        st(Jz + o, cadd(jz, rmul(kzz, imul(rmul(-1., F)))));
    }
}

// fields/numba_methods.py:87-116 (cross-deposition correction, standard PSATD): Dz + Dxy is
// the error of the continuity equation, split with the help of two extra charge densities
// deposited at (z[n], x[n+1]) [rho_next_xy] and (z[n+1], x[n]) [rho_next_z].
// COMOVING: numba_methods.py:243-275, with the complex tables of the Galilean scheme.
template <bool COMOVING>
__global__ __launch_bounds__(256) void k_correct_currents_cross(const cplx *__restrict__ rho_prev,
        const cplx *__restrict__ rho_next, const cplx *__restrict__ rho_next_z,
        const cplx *__restrict__ rho_next_xy, cplx *__restrict__ Jp, cplx *__restrict__ Jm,
        cplx *__restrict__ Jz, long rs, const double *__restrict__ kz, const double *__restrict__ kr,
        double inv_dt, const cplx *__restrict__ j_corr_coef, const cplx *__restrict__ T_eb,
        const cplx *__restrict__ T_cc, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        const double kzz = kz[idx], krr = kr[idx];
        const cplx jp = ld(Jp + o), jm = ld(Jm + o), jz = ld(Jz + o);
        const cplx rn = ld(rho_next + o), rz = ld(rho_next_z + o), rxy = ld(rho_next_xy + o);
        cplx rp = ld(rho_prev + o), sz, sxy;
        if (COMOVING) {
            const cplx teb = ld(T_eb + idx);
            const cplx txy = cmul(teb, rxy);
            rp = cmul(teb, rp);
            sz = csub(cadd(csub(rn, txy), rz), rp);
            sxy = csub(csub(cadd(rn, txy), rz), rp);
            const cplx h = cmul(rmul(0.5, ld(T_cc + idx)), ld(j_corr_coef + idx));
            sz = cmul(h, sz);
            sxy = cmul(h, sxy);
        } else {
            sz = rmul(0.5 * inv_dt, csub(cadd(csub(rn, rxy), rz), rp));
            sxy = rmul(0.5 * inv_dt, csub(cadd(csub(rn, rz), rxy), rp));
        }
        const cplx Dz = cadd(rmul(kzz, imul(jz)), sz);
        const cplx Dxy = cadd(rmul(krr, csub(jp, jm)), sxy);
        if (krr != 0.) {
            const double inv_kr = 1. / krr;
            st(Jp + o, cadd(jp, rmul(inv_kr, rmul(-0.5, Dxy))));
            st(Jm + o, cadd(jm, rmul(inv_kr, rmul(0.5, Dxy))));
        }
        if (kzz != 0.) {
            const double inv_kz = 1. / kzz;
            st(Jz + o, cadd(jz, rmul(inv_kz, imul(Dz))));
        }
    }
}

__global__ __launch_bounds__(256) void k_push_eb_comoving(cplx *__restrict__ Ep, cplx *__restrict__ Em,
        cplx *__restrict__ Ez, cplx *__restrict__ Bp, cplx *__restrict__ Bm, cplx *__restrict__ Bz,
        const cplx *__restrict__ Jp, const cplx *__restrict__ Jm, const cplx *__restrict__ Jz,
        const cplx *__restrict__ rho_prev, const cplx *__restrict__ rho_next, long rs,
        const cplx *__restrict__ rho_prev_coef, const cplx *__restrict__ rho_next_coef,
        const cplx *__restrict__ j_coef, const double *__restrict__ C, const double *__restrict__ S_w,
        const cplx *__restrict__ T_eb, const cplx *__restrict__ T_cc, const cplx *__restrict__ T_rho,
        const double *__restrict__ kr, const double *__restrict__ kz, double V, int use_true_rho,
        double c2, double eps0, double mu0, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        const double krr = kr[idx], kzz = kz[idx], Cc = C[idx], Sw = S_w[idx];
        const cplx jc = ld(j_coef + idx), Teb = ld(T_eb + idx), Tcc = ld(T_cc + idx);
        const cplx rnc = ld(rho_next_coef + idx), rpc = ld(rho_prev_coef + idx);
        const cplx ep = ld(Ep + o), em = ld(Em + o), ez = ld(Ez + o);
        const cplx bp = ld(Bp + o), bm = ld(Bm + o), bz = ld(Bz + o);
        const cplx jp = ld(Jp + o), jm = ld(Jm + o), jz = ld(Jz + o);
        cplx rho_diff;
        if (use_true_rho) {
            rho_diff = csub(cmul(rnc, ld(rho_next + o)), cmul(rpc, ld(rho_prev + o)));
        } else {
            const cplx divE = cadd(rmul(krr, csub(ep, em)), rmul(kzz, imul(ez)));
            const cplx divJ = cadd(rmul(krr, csub(jp, jm)), rmul(kzz, imul(jz)));
            const cplx a = rmul(eps0, csub(cmul(Teb, rnc), rpc));
            rho_diff = cadd(cmul(a, divE), cmul(cmul(ld(T_rho + idx), rnc), divJ));
        }
        const cplx TC = rmul(Cc, Teb);
        const cplx jikzV = cmul(jc, cplx{0., kzz * V});
        const cplx TS = rmul(c2 * Sw, Teb);
        const cplx mihkBz = rmul(0.5 * krr, imul(rmul(-1., bz)));
        st(Ep + o, cadd(cadd(cadd(cmul(TC, ep), rmul(0.5 * krr, rho_diff)), cmul(jikzV, jp)),
                        cmul(TS, csub(cadd(mihkBz, rmul(kzz, bp)), rmul(mu0, cmul(Tcc, jp))))));
        st(Em + o, cadd(cadd(csub(cmul(TC, em), rmul(0.5 * krr, rho_diff)), cmul(jikzV, jm)),
                        cmul(TS, csub(csub(mihkBz, rmul(kzz, bm)), rmul(mu0, cmul(Tcc, jm))))));
        st(Ez + o, cadd(cadd(csub(cmul(TC, ez), rmul(kzz, imul(rho_diff))), cmul(jikzV, jz)),
                        cmul(TS, csub(cadd(rmul(krr, imul(bp)), rmul(krr, imul(bm))),
                                      rmul(mu0, cmul(Tcc, jz))))));
        const cplx TSb = rmul(Sw, Teb);
        const cplx mihkEz = rmul(0.5 * krr, imul(rmul(-1., ez)));
        const cplx mihkJz = rmul(0.5 * krr, imul(rmul(-1., jz)));
        st(Bp + o, cadd(csub(cmul(TC, bp), cmul(TSb, cadd(mihkEz, rmul(kzz, ep)))),
                        cmul(jc, cadd(mihkJz, rmul(kzz, jp)))));
        st(Bm + o, cadd(csub(cmul(TC, bm), cmul(TSb, csub(mihkEz, rmul(kzz, em)))),
                        cmul(jc, csub(mihkJz, rmul(kzz, jm)))));
        st(Bz + o, cadd(csub(cmul(TC, bz), cmul(TSb, cadd(rmul(krr, imul(ep)), rmul(krr, imul(em))))),
                        cmul(jc, cadd(rmul(krr, imul(jp)), rmul(krr, imul(jm))))));
    }
}

// boundaries/moving_window.py:204-239 (shift_spect_array_cpu): F[iz,:] *= shift[iz]^n_move,
// the power by repeated multiplication, conjugated for n_move < 0
__global__ __launch_bounds__(256) void k_shift_spect(int nf, Ptrs48 P, long rs,
        const cplx *__restrict__ shift, int n_move, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const cplx sh = ld(shift + iz);
        cplx pw = {1., 0.};
        const int na = n_move < 0 ? -n_move : n_move;
        for (int i = 0; i < na; i++) pw = {pw.re * sh.re - pw.im * sh.im, pw.re * sh.im + pw.im * sh.re};
        if (n_move < 0) pw.im = -pw.im;
        for (int f = 0; f < nf; f++) {
            cplx *p = (cplx *)P.p[f] + (long)iz * rs + ir;
            const cplx a = ld(p);
            st(p, {a.re * pw.re - a.im * pw.im, a.re * pw.im + a.im * pw.re});
        }
    }
}

// fields/spectral_grid.py:407-421
__global__ __launch_bounds__(256) void k_push_rho(cplx *__restrict__ rho_prev,
                                                  cplx *__restrict__ rho_next, long rs, int Nz, int Nr)
{
    FB_GRID_LOOP(idx, iz, ir) {
        const long o = (long)iz * rs + ir;
        st(rho_prev + o, ld(rho_next + o));
        st(rho_next + o, {0., 0.});
    }
}

static inline int grid_for(int Nz, int Nr) { return stream_grid((long)Nz * Nr, 256, 256 * 8); }

template <class T, class U>
static bool fill48(T &dst, U *const *src, int n, const char *where)
{
    if (n < 0 || n > 48) { set_error(where, "more than 48 fields"); return false; }
    for (int i = 0; i < 48; i++) dst.p[i] = i < n ? src[i] : nullptr;
    return true;
}

}  // namespace fb

using namespace fb;

extern "C" int fb_erase(int nf, void *const *ptrs, long rs, int Nz, int Nr, void *stream)
{
    Ptrs48 P;
    if (!fill48(P, ptrs, nf, "fb_erase")) return -1;
    if (nf == 0) return 0;
    hipLaunchKernelGGL(k_erase, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream, nf, P, rs, Nz, Nr);
    FB_CHECK_LAUNCH("fb_erase");
}

extern "C" int fb_divide_by_volume(int nf, void *const *ptrs, long rs, const double *invvol,
                                   int Nz, int Nr, void *stream)
{
    Ptrs48 P;
    if (!fill48(P, ptrs, nf, "fb_divide_by_volume")) return -1;
    if (nf == 0) return 0;
    hipLaunchKernelGGL(k_divide, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream, nf, P, rs,
                       invvol, Nz, Nr);
    FB_CHECK_LAUNCH("fb_divide_by_volume");
}

extern "C" int fb_filter(int nf, void *const *ptrs, long rs, const double *fz, const double *fr,
                         int Nz, int Nr, void *stream)
{
    Ptrs48 P;
    if (!fill48(P, ptrs, nf, "fb_filter")) return -1;
    if (nf == 0) return 0;
    hipLaunchKernelGGL(k_filter, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream, nf, P, rs,
                       fz, fr, Nz, Nr);
    FB_CHECK_LAUNCH("fb_filter");
}

extern "C" int fb_scale(int nf, void *const *ptrs, long rs, double factor, int Nz, int Nr,
                        void *stream)
{
    Ptrs48 P;
    if (!fill48(P, ptrs, nf, "fb_scale")) return -1;
    if (nf == 0) return 0;
    hipLaunchKernelGGL(k_scale, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream, nf, P, rs,
                       factor, Nz, Nr);
    FB_CHECK_LAUNCH("fb_scale");
}

extern "C" int fb_rt_to_pm(int np, const void *const *r, const void *const *t, void *const *p,
                           void *const *m, long rs, int Nz, int Nr, void *stream)
{
    CPtrs48 R, T;
    Ptrs48 P, M;
    if (!fill48(R, r, np, "fb_rt_to_pm") || !fill48(T, t, np, "fb_rt_to_pm") ||
        !fill48(P, p, np, "fb_rt_to_pm") || !fill48(M, m, np, "fb_rt_to_pm")) return -1;
    if (np == 0) return 0;
    hipLaunchKernelGGL(k_rt_to_pm, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream, np, R, T,
                       P, M, rs, Nz, Nr);
    FB_CHECK_LAUNCH("fb_rt_to_pm");
}

extern "C" int fb_pm_to_rt(int np, const void *const *p, const void *const *m, void *const *r,
                           void *const *t, long rs, int Nz, int Nr, void *stream)
{
    CPtrs48 P, M;
    Ptrs48 R, T;
    if (!fill48(P, p, np, "fb_pm_to_rt") || !fill48(M, m, np, "fb_pm_to_rt") ||
        !fill48(R, r, np, "fb_pm_to_rt") || !fill48(T, t, np, "fb_pm_to_rt")) return -1;
    if (np == 0) return 0;
    hipLaunchKernelGGL(k_pm_to_rt, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream, np, P, M,
                       R, T, rs, Nz, Nr);
    FB_CHECK_LAUNCH("fb_pm_to_rt");
}

extern "C" int fb_correct_currents_curlfree_standard(const void *rho_prev, const void *rho_next,
        void *Jp, void *Jm, void *Jz, long rs, const double *kz, const double *kr,
        const double *inv_k2, double inv_dt, int Nz, int Nr, void *stream)
{
    hipLaunchKernelGGL(k_correct_currents, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream,
                       (const cplx *)rho_prev, (const cplx *)rho_next, (cplx *)Jp, (cplx *)Jm,
                       (cplx *)Jz, rs, kz, kr, inv_k2, inv_dt, Nz, Nr);
    FB_CHECK_LAUNCH("fb_correct_currents_curlfree_standard");
}

extern "C" int fb_push_eb_standard(void *Ep, void *Em, void *Ez, void *Bp, void *Bm, void *Bz,
        const void *Jp, const void *Jm, const void *Jz, const void *rho_prev,
        const void *rho_next, long rs, const double *rho_prev_coef, const double *rho_next_coef,
        const double *j_coef, const double *C, const double *S_w, const double *kr,
        const double *kz, double dt, int use_true_rho, double c, double epsilon_0, double mu_0,
        int Nz, int Nr, void *stream)
{
    hipLaunchKernelGGL(k_push_eb, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream,
                       (cplx *)Ep, (cplx *)Em, (cplx *)Ez, (cplx *)Bp, (cplx *)Bm, (cplx *)Bz,
                       (const cplx *)Jp, (const cplx *)Jm, (const cplx *)Jz,
                       (const cplx *)rho_prev, (const cplx *)rho_next, rs, rho_prev_coef,
                       rho_next_coef, j_coef, C, S_w, kr, kz, dt, use_true_rho, c * c,
                       epsilon_0, mu_0, Nz, Nr);
    FB_CHECK_LAUNCH("fb_push_eb_standard");
}

extern "C" int fb_correct_currents_curlfree_comoving(const void *rho_prev, const void *rho_next,
        void *Jp, void *Jm, void *Jz, long rs, const double *kz, const double *kr,
        const double *inv_k2, const void *j_corr_coef, const void *T_eb, const void *T_cc,
        int Nz, int Nr, void *stream)
{
    hipLaunchKernelGGL(k_correct_currents_comoving, dim3(grid_for(Nz, Nr)), dim3(256), 0,
                       (hipStream_t)stream, (const cplx *)rho_prev, (const cplx *)rho_next, (cplx *)Jp,
                       (cplx *)Jm, (cplx *)Jz, rs, kz, kr, inv_k2, (const cplx *)j_corr_coef,
                       (const cplx *)T_eb, (const cplx *)T_cc, Nz, Nr);
    FB_CHECK_LAUNCH("fb_correct_currents_curlfree_comoving");
}

extern "C" int fb_correct_currents_crossdeposition_standard(const void *rho_prev,
        const void *rho_next, const void *rho_next_z, const void *rho_next_xy,
        void *Jp, void *Jm, void *Jz, long rs, const double *kz, const double *kr,
        double inv_dt, int Nz, int Nr, void *stream)
{
    hipLaunchKernelGGL(k_correct_currents_cross<false>, dim3(grid_for(Nz, Nr)), dim3(256), 0,
                       (hipStream_t)stream, (const cplx *)rho_prev, (const cplx *)rho_next,
                       (const cplx *)rho_next_z, (const cplx *)rho_next_xy, (cplx *)Jp, (cplx *)Jm,
                       (cplx *)Jz, rs, kz, kr, inv_dt, (const cplx *)nullptr, (const cplx *)nullptr,
                       (const cplx *)nullptr, Nz, Nr);
    FB_CHECK_LAUNCH("fb_correct_currents_crossdeposition_standard");
}

extern "C" int fb_correct_currents_crossdeposition_comoving(const void *rho_prev,
        const void *rho_next, const void *rho_next_z, const void *rho_next_xy,
        void *Jp, void *Jm, void *Jz, long rs, const double *kz, const double *kr,
        const void *j_corr_coef, const void *T_eb, const void *T_cc, int Nz, int Nr, void *stream)
{
    hipLaunchKernelGGL(k_correct_currents_cross<true>, dim3(grid_for(Nz, Nr)), dim3(256), 0,
                       (hipStream_t)stream, (const cplx *)rho_prev, (const cplx *)rho_next,
                       (const cplx *)rho_next_z, (const cplx *)rho_next_xy, (cplx *)Jp, (cplx *)Jm,
                       (cplx *)Jz, rs, kz, kr, 0., (const cplx *)j_corr_coef, (const cplx *)T_eb,
                       (const cplx *)T_cc, Nz, Nr);
    FB_CHECK_LAUNCH("fb_correct_currents_crossdeposition_comoving");
}

extern "C" int fb_push_eb_comoving(void *Ep, void *Em, void *Ez, void *Bp, void *Bm, void *Bz,
        const void *Jp, const void *Jm, const void *Jz, const void *rho_prev,
        const void *rho_next, long rs, const void *rho_prev_coef, const void *rho_next_coef,
        const void *j_coef, const double *C, const double *S_w, const void *T_eb, const void *T_cc,
        const void *T_rho, const double *kr, const double *kz, double dt, double V,
        int use_true_rho, double c, double epsilon_0, double mu_0, int Nz, int Nr, void *stream)
{
    (void)dt;
    hipLaunchKernelGGL(k_push_eb_comoving, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream,
                       (cplx *)Ep, (cplx *)Em, (cplx *)Ez, (cplx *)Bp, (cplx *)Bm, (cplx *)Bz,
                       (const cplx *)Jp, (const cplx *)Jm, (const cplx *)Jz, (const cplx *)rho_prev,
                       (const cplx *)rho_next, rs, (const cplx *)rho_prev_coef,
                       (const cplx *)rho_next_coef, (const cplx *)j_coef, C, S_w, (const cplx *)T_eb,
                       (const cplx *)T_cc, (const cplx *)T_rho, kr, kz, V, use_true_rho, c * c,
                       epsilon_0, mu_0, Nz, Nr);
    FB_CHECK_LAUNCH("fb_push_eb_comoving");
}

extern "C" int fb_push_rho(void *rho_prev, void *rho_next, long rs, int Nz, int Nr, void *stream)
{
    hipLaunchKernelGGL(k_push_rho, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream,
                       (cplx *)rho_prev, (cplx *)rho_next, rs, Nz, Nr);
    FB_CHECK_LAUNCH("fb_push_rho");
}

extern "C" int fb_psatd_step_standard_shift(int Nm, void *const *fields, long rs,
        const double *const *tables, double dt, int correct_currents, int use_true_rho,
        double c, double epsilon_0, double mu_0, int Nz, int Nr, const void *shift, int n_move,
        void *stream)
{
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error("fb_psatd_step_standard", "Nm out of range"); return -1; }
    if (n_move != 0 && correct_currents == 2) {
        set_error("fb_psatd_step_standard_shift", "the window shift belongs to the push, not to the correction-only call");
        return -1;
    }
    PsatdModes M;
    for (int i = 0; i < 11 * FB_MAX_MODES; i++) M.f[i] = i < 11 * Nm ? (cplx *)fields[i] : nullptr;
    for (int i = 0; i < 8 * FB_MAX_MODES; i++) M.t[i] = i < 8 * Nm ? tables[i] : nullptr;
    dim3 grid(stream_grid((long)Nz * Nr, 256, 256 * 4), Nm);
    // (240 B of operands per cell and mode: beyond ~3/4 of the Infinity Cache they are read non-temporally)
    const bool big = 240. * (double)Nz * (double)Nr * (double)Nm > 192. * 1048576.;
    if (big)
        hipLaunchKernelGGL(k_psatd_step<true>, grid, dim3(256), 0, (hipStream_t)stream, M, rs, dt, 1. / dt,
                           correct_currents, use_true_rho, c * c, epsilon_0, mu_0, Nz, Nr,
                           (const cplx *)shift, n_move);
    else
        hipLaunchKernelGGL(k_psatd_step<false>, grid, dim3(256), 0, (hipStream_t)stream, M, rs, dt, 1. / dt,
                           correct_currents, use_true_rho, c * c, epsilon_0, mu_0, Nz, Nr,
                           (const cplx *)shift, n_move);
    FB_CHECK_LAUNCH("fb_psatd_step_standard");
}

extern "C" int fb_psatd_step_standard(int Nm, void *const *fields, long rs,
        const double *const *tables, double dt, int correct_currents, int use_true_rho,
        double c, double epsilon_0, double mu_0, int Nz, int Nr, void *stream)
{
    return fb_psatd_step_standard_shift(Nm, fields, rs, tables, dt, correct_currents, use_true_rho, c,
                                        epsilon_0, mu_0, Nz, Nr, nullptr, 0, stream);
}

extern "C" int fb_shift_spect(int nf, void *const *ptrs, long rs, const void *shift, int n_move,
                              int Nz, int Nr, void *stream)
{
    Ptrs48 P;
    if (!fill48(P, ptrs, nf, "fb_shift_spect")) return -1;
    if (nf == 0 || n_move == 0) return 0;
    hipLaunchKernelGGL(k_shift_spect, dim3(grid_for(Nz, Nr)), dim3(256), 0, (hipStream_t)stream, nf, P,
                       rs, (const cplx *)shift, n_move, Nz, Nr);
    FB_CHECK_LAUNCH("fb_shift_spect");
}

// ---- guard-cell buffers of the z-domain decomposition ------------------------------------
// boundaries/cuda_methods.py:12-195 (copy_*_to_gpu_buffer), :197-372 (replace_*_from_gpu_buffer),
// :374-484 (add_*_from_gpu_buffer): one launch moves the `nrows` slab rows next to BOTH z ends
// between the slab and two contiguous message buffers.  A field group (all modes and
// components) is `ncontig` adjacent complex values of every z row of the z-major slab.
// MODE 0: slab -> buffers, 1: buffers replace slab, 2: buffers add to slab.
template <int MODE>
__global__ __launch_bounds__(256) void k_guard_buffers(cplx *__restrict__ slab, long rs, long ncontig,
        int z_left, int z_right, int nrows, cplx *__restrict__ buf_l, cplx *__restrict__ buf_r)
{
    const long per_side = (long)nrows * ncontig;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * per_side; i += stride) {
        const bool right = i >= per_side;
        cplx *buf = right ? buf_r : buf_l;
        if (buf == nullptr) continue;
        const long j = right ? i - per_side : i;
        const long row = j / ncontig, col = j - row * ncontig;
        cplx *g = slab + ((right ? z_right : z_left) + row) * rs + col;
        if (MODE == 0) st(buf + j, ld(g));
        else if (MODE == 1) st(g, ld(buf + j));
        else st(g, cadd(ld(g), ld(buf + j)));
    }
}

// boundaries/cuda_methods.py:486-640 (cuda_damp_EB_left / _right): rows [0, nd_l) of the group
// are multiplied by damp_l[iz], rows [Nz - nd_r, Nz) by damp_r[iz - (Nz - nd_r)].
__global__ __launch_bounds__(256) void k_damp_rows(cplx *__restrict__ slab, long rs, long ncontig,
        const double *__restrict__ damp_l, int nd_l, const double *__restrict__ damp_r, int nd_r,
        int Nz)
{
    const long total = (long)(nd_l + nd_r) * ncontig;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long row = i / ncontig, col = i - row * ncontig;
        const bool right = row >= nd_l;
        const long iz = right ? (Nz - nd_r) + (row - nd_l) : row;
        const double d = right ? damp_r[row - nd_l] : damp_l[row];
        cplx *g = slab + iz * rs + col;
        st(g, rmul(d, ld(g)));
    }
}

extern "C" int fb_guard_buffers(int mode, void *slab, long row_stride, long ncontig, int z_left,
        int z_right, int nrows, void *buf_left, void *buf_right, void *stream)
{
    if (mode < 0 || mode > 2) { set_error("fb_guard_buffers", "mode must be 0, 1 or 2"); return -1; }
    if (nrows <= 0 || ncontig <= 0 || (buf_left == nullptr && buf_right == nullptr)) return 0;
    const dim3 grid(stream_grid(2L * nrows * ncontig)), block(256);
#define FB_GUARD_LAUNCH(M)                                                                      \
    hipLaunchKernelGGL(k_guard_buffers<M>, grid, block, 0, (hipStream_t)stream, (cplx *)slab,      \
                       row_stride, ncontig, z_left, z_right, nrows, (cplx *)buf_left, (cplx *)buf_right)
    if (mode == 0) FB_GUARD_LAUNCH(0);
    else if (mode == 1) FB_GUARD_LAUNCH(1);
    else FB_GUARD_LAUNCH(2);
#undef FB_GUARD_LAUNCH
    FB_CHECK_LAUNCH("fb_guard_buffers");
}

extern "C" int fb_damp_rows(void *slab, long row_stride, long ncontig, const double *damp_left,
        int nd_left, const double *damp_right, int nd_right, int Nz, void *stream)
{
    if (damp_left == nullptr) nd_left = 0;
    if (damp_right == nullptr) nd_right = 0;
    if (nd_left + nd_right <= 0 || ncontig <= 0) return 0;
    hipLaunchKernelGGL(k_damp_rows, dim3(stream_grid((long)(nd_left + nd_right) * ncontig)), dim3(256),
                       0, (hipStream_t)stream, (cplx *)slab, row_stride, ncontig, damp_left, nd_left,
                       damp_right, nd_right, Nz);
    FB_CHECK_LAUNCH("fb_damp_rows");
}
