// Particle kernels for gfx950: position/momentum push, periodic shift, field gather.
// One lane = one (or two, 16-byte vectorised) macroparticle(s); structure-of-arrays
// float64 streams, coalesced; grid-stride over a capped grid (256 CUs x 8 blocks).
#include "fb_common.h"

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

// ------------------------------------------------------------------ push_p
// Vay pusher, fbpic/particles/push/inline_functions.py:11-48.  112 B / particle.
__device__ __forceinline__ void vay(double &ux, double &uy, double &uz, double &ig,
        double Ex, double Ey, double Ez, double Bx, double By, double Bz,
        double econst, double bconst)
{
    double taux = bconst * Bx, tauy = bconst * By, tauz = bconst * Bz;
    double tau2 = taux * taux + tauy * tauy + tauz * tauz;
    double uxp = ux + econst * Ex + ig * (uy * tauz - uz * tauy);
    double uyp = uy + econst * Ey + ig * (uz * taux - ux * tauz);
    double uzp = uz + econst * Ez + ig * (ux * tauy - uy * taux);
    double sigma = 1 + uxp * uxp + uyp * uyp + uzp * uzp - tau2;
    double utau = uxp * taux + uyp * tauy + uzp * tauz;
    double igf = sqrt(2. / (sigma + sqrt(sigma * sigma + 4 * (tau2 + utau * utau))));
    double tx = igf * taux, ty = igf * tauy, tz = igf * tauz;
    double ut = igf * utau;
    double s = 1. / (1 + tau2 * (igf * igf));
    ux = s * (uxp + tx * ut + uyp * tz - uzp * ty);
    uy = s * (uyp + ty * ut + uzp * tx - uxp * tz);
    uz = s * (uzp + tz * ut + uxp * ty - uyp * tx);
    ig = igf;
}

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
// Field gather for any number of azimuthal modes in ONE pass over the particles
// (the reference needs Nm launches + an erase for Nm != 2).  Semantics:
// fbpic/particles/gathering/threading_methods.py:25-201 (linear), :207-367 (cubic),
// inline_functions.py:9-187; exptheta_m by recurrence; modes are summed before the
// (r,t)->(x,y) rotation as in the Nm==2 kernels.
struct GatherGrids { const cplx *g[6 * FB_MAX_MODES]; };

__device__ __forceinline__ double2 ldc(const cplx *p) { return *(const double2 *)p; }

template <int SHAPE>
__global__ __launch_bounds__(256) void k_gather(int Nm, long n,
        const double *__restrict__ x, const double *__restrict__ y,
        const double *__restrict__ z, double rmax_gather,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        GatherGrids G, long rs,
        double *__restrict__ Ex, double *__restrict__ Ey, double *__restrict__ Ez,
        double *__restrict__ Bx, double *__restrict__ By, double *__restrict__ Bz)
{
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double xj = x[i], yj = y[i], zj = z[i];
        double rj = sqrt(xj * xj + yj * yj);
        double cs, sn;
        if (rj != 0.) { double invr = 1. / rj; cs = xj * invr; sn = yj * invr; }
        else { cs = 1.; sn = 0.; }
        double r_cell = invdr * (rj - rmin) - 0.5;
        double z_cell = invdz * (zj - zmin) - 0.5;
        double F[6] = {0., 0., 0., 0., 0., 0.};   // Er,Et,Ez,Br,Bt,Bz summed over modes
        if (rj < rmax_gather) {
            if constexpr (SHAPE == FB_SHAPE_LINEAR) {
                int irl = (int)floor(r_cell), iru = irl + 1;
                int izl = (int)floor(z_cell), izu = izl + 1;
                double Srl = iru - r_cell, Sru = r_cell - irl;
                double Szl = izu - z_cell, Szu = z_cell - izl;
                double Srg = 0.;
                if (irl < 0) { Srg = Srl; Srl = 0.; irl = 0; }
                if (irl > Nr - 1) irl = Nr - 1;
                if (iru > Nr - 1) iru = Nr - 1;
                if (izl < 0) izl += Nz;
                if (izu < 0) izu += Nz;
                if (izl > Nz - 1) izl -= Nz;
                if (izu > Nz - 1) izu -= Nz;
                const double S_ll = Szl * Srl, S_lu = Szl * Sru, S_ul = Szu * Srl,
                             S_uu = Szu * Sru, S_lg = Szl * Srg, S_ug = Szu * Srg;
                const bool guard = (irl == 0 && iru == 0);
                const long o_ll = (long)izl * rs + irl, o_lu = (long)izl * rs + iru,
                           o_ul = (long)izu * rs + irl, o_uu = (long)izu * rs + iru,
                           o_lg = (long)izl * rs, o_ug = (long)izu * rs;
                double er = 1., ei = 0.;   // exptheta_m = (cos - i sin)^m
                for (int m = 0; m < Nm; m++) {
                    const double flip = m1pow(m);
                    const double factor = (m == 0) ? 1. : 2.;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const cplx *g = G.g[6 * m + k];
                        double2 a = ldc(g + o_ll), b = ldc(g + o_lu), cc = ldc(g + o_ul),
                                d = ldc(g + o_uu);
                        double fr = 0., fi = 0.;
                        fr += S_ll * a.x; fi += S_ll * a.y;
                        fr += S_lu * b.x; fi += S_lu * b.y;
                        fr += S_ul * cc.x; fi += S_ul * cc.y;
                        fr += S_uu * d.x; fi += S_uu * d.y;
                        if (guard) {
                            // r,t components: -(-1)^m ; z component: +(-1)^m
                            const double sg = (k % 3 == 2) ? flip : -flip;
                            double2 gl = ldc(g + o_lg), gu = ldc(g + o_ug);
                            fr += sg * S_lg * gl.x; fi += sg * S_lg * gl.y;
                            fr += sg * S_ug * gu.x; fi += sg * S_ug * gu.y;
                        }
                        F[k] += factor * (fr * er - fi * ei);
                    }
                    // next mode: (er + i ei) *= (cos - i sin)
                    double nr_ = er * cs - ei * (-sn);
                    double ni_ = er * (-sn) + ei * cs;
                    er = nr_; ei = ni_;
                }
            } else {
                double Sr[4], Sz[4];
                const long ir_lowest = (long)floor(r_cell) - 1;
                const long iz_lowest = (long)floor(z_cell) - 1;
                {
                    double l = r_cell - ir_lowest;
                    double a = l - 2., b = l - 1., cc = 2. - l, d = 1. - l;
                    Sr[0] = -1. / 6. * (a * (a * a));
                    Sr[1] = 1. / 6. * (3. * (b * (b * b)) - 6. * (b * b) + 4.);
                    Sr[2] = 1. / 6. * (3. * (cc * (cc * cc)) - 6. * (cc * cc) + 4.);
                    Sr[3] = -1. / 6. * (d * (d * d));
                    l = z_cell - iz_lowest;
                    a = l - 2.; b = l - 1.; cc = 2. - l; d = 1. - l;
                    Sz[0] = -1. / 6. * (a * (a * a));
                    Sz[1] = 1. / 6. * (3. * (b * (b * b)) - 6. * (b * b) + 4.);
                    Sz[2] = 1. / 6. * (3. * (cc * (cc * cc)) - 6. * (cc * cc) + 4.);
                    Sz[3] = -1. / 6. * (d * (d * d));
                }
                long irs[4], izs[4];
                bool below[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    long ir = ir_lowest + j;
                    below[j] = ir < 0;
                    if (ir < 0) ir = -ir - 1;
                    else if (ir > Nr - 1) ir = Nr - 1;
                    irs[j] = ir;
                    long iz = iz_lowest + j;
                    if (iz < 0) iz += Nz;
                    else if (iz > Nz - 1) iz -= Nz;
                    izs[j] = iz * rs;
                }
                double er = 1., ei = 0.;
                for (int m = 0; m < Nm; m++) {
                    const double flip = m1pow(m);
                    const double factor = (m == 0) ? 1. : 2.;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const cplx *g = G.g[6 * m + k];
                        const double sg = (k % 3 == 2) ? flip : -flip;
                        double fr = 0., fi = 0.;
#pragma unroll
                        for (int jr = 0; jr < 4; jr++) {
                            double sr = Sr[jr];
                            if (below[jr]) sr *= sg;
#pragma unroll
                            for (int jz = 0; jz < 4; jz++) {
                                double2 v = ldc(g + izs[jz] + irs[jr]);
                                double s = Sz[jz] * sr;
                                fr += s * v.x; fi += s * v.y;
                            }
                        }
                        F[k] += factor * (fr * er - fi * ei);
                    }
                    double nr_ = er * cs - ei * (-sn);
                    double ni_ = er * (-sn) + ei * cs;
                    er = nr_; ei = ni_;
                }
            }
        }
        Ex[i] = cs * F[0] - sn * F[1];
        Ey[i] = sn * F[0] + cs * F[1];
        Ez[i] = F[2];
        Bx[i] = cs * F[3] - sn * F[4];
        By[i] = sn * F[3] + cs * F[4];
        Bz[i] = F[5];
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

extern "C" int fb_gather(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, double rmax_gather, double invdz, double zmin, int Nz, double invdr,
        double rmin, int Nr, const void *const *grids, long row_stride,
        double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz, void *stream)
{
    if (n <= 0) return 0;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error("fb_gather", "Nm out of range"); return -1; }
    GatherGrids G;
    for (int i = 0; i < 6 * Nm; i++) G.g[i] = (const cplx *)grids[i];
    for (int i = 6 * Nm; i < 6 * FB_MAX_MODES; i++) G.g[i] = nullptr;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(stream_grid(n, 256, 256 * 16)), block(256);
    if (shape == FB_SHAPE_LINEAR)
        hipLaunchKernelGGL(k_gather<FB_SHAPE_LINEAR>, grid, block, 0, s, Nm, n, x, y, z,
                           rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride,
                           Ex, Ey, Ez, Bx, By, Bz);
    else if (shape == FB_SHAPE_CUBIC)
        hipLaunchKernelGGL(k_gather<FB_SHAPE_CUBIC>, grid, block, 0, s, Nm, n, x, y, z,
                           rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride,
                           Ex, Ey, Ez, Bx, By, Bz);
    else { set_error("fb_gather", "unknown shape"); return -1; }
    FB_CHECK_LAUNCH("fb_gather");
}
