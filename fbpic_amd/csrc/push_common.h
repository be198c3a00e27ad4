// Pieces shared by particles.hip and cycle.hip: the Vay momentum push and the gather's grid table.
#pragma once
#include "fb_common.h"

namespace fb {

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

// grids of the gather: for mode m, g[6m .. 6m+5] = Er, Et, Ez, Br, Bt, Bz
struct GatherGrids { const cplx *g[6 * FB_MAX_MODES]; };

__device__ __forceinline__ double2 ldc(const cplx *p) { return *(const double2 *)p; }

}  // namespace fb
