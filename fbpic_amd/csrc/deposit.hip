// Charge / current deposition for gfx950.
//
// Design (MI355X-first, not the reference's one-thread-per-cell CUDA layout):
//   * particles are cell-sorted (sort.hip); a workgroup owns a TZ x TR tile of cells and
//     therefore a few contiguous particle ranges (one per z row of the tile) that it
//     streams with coalesced SoA loads, one lane per macroparticle;
//   * the tile's stencil footprint ((TZ+S-1) x (TR+S) grid nodes, all components and
//     modes) is privatised in LDS and accumulated with ds_add_f64; the deposition guard
//     cells of the reference (below the axis / beyond rmax / periodic z,
//     fbpic/fields/numba_methods.py:409-461) are folded when the tile is flushed to HBM
//     with global_atomic_add_f64;
//   * any contribution that falls outside the LDS tile (possible only if the sort is
//     stale) goes straight to HBM with the same folding, so the result never depends on
//     the sort being exact -- only the speed does.
// Numerics: shape factors, Ruyten correction, axis flips and the mode recurrence restate
// fbpic/particles/deposition/particle_shapes.py:17-80 and threading_methods.py:27-650.
// Summation order differs from any CPU run (as the reference's own GPU path does):
// parity is to 1e-13 * max|F| (tests/test_cpu_gpu_deposition.py:96).
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

// NCOMP = 1 (rho) or 3 (Jr,Jt,Jz); NM = modes handled by this launch (m0 .. m0+NM-1)
template <int SHAPE, int NCOMP, int NM>
__global__ __launch_bounds__(256) void k_deposit(long n,
        const double *__restrict__ x, const double *__restrict__ y,
        const double *__restrict__ z, const double *__restrict__ w, double q,
        const double *__restrict__ ux, const double *__restrict__ uy,
        const double *__restrict__ uz, const double *__restrict__ inv_gamma, double c_light,
        double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        DepGrids G, long rs, int m0,
        const int *__restrict__ prefix_sum,
        const double *__restrict__ beta0, const double *__restrict__ betah,
        int TZ, int TR, int n_rtiles)
{
    constexpr int S = ShapeTraits<SHAPE>::S, H = ShapeTraits<SHAPE>::H;
    constexpr int NK = NCOMP * NM * 2;                 // doubles per node
    extern __shared__ double lds[];
    const int tz_id = blockIdx.x / n_rtiles, tr_id = blockIdx.x % n_rtiles;
    const int z0 = tz_id * TZ, z1 = min(Nz, z0 + TZ);
    const int r0 = tr_id * TR, r1 = min(Nr + 1, r0 + TR);
    const int NTZ = TZ + S - 1, NTR = TR + S;          // node footprint (+1 col: r clamp)
    const int plane = NTZ * NTR;
    const int nz_org = z0 - H, nr_org = r0 - H;        // node (row, col) of tile origin
    for (int i = threadIdx.x; i < NK * plane; i += blockDim.x) lds[i] = 0.;
    __syncthreads();

    for (int izu = z0; izu < z1; izu++) {
        const long c0 = (long)izu * (Nr + 1) + r0, c1 = (long)izu * (Nr + 1) + r1;
        const long p_beg = (c0 > 0) ? prefix_sum[c0 - 1] : 0;
        const long p_end = prefix_sum[c1 - 1];
        for (long ip = p_beg + threadIdx.x; ip < p_end; ip += blockDim.x) {
            const double xj = x[ip], yj = y[ip], zj = z[ip];
            const double wj = q * w[ip];
            const double rj = sqrt(xj * xj + yj * yj);
            double cs, sn;
            if (rj != 0.) { double invr = 1. / rj; cs = xj * invr; sn = yj * invr; }
            else { cs = 1.; sn = 0.; }
            // amplitudes for mode m0 ... (threading_methods.py:119-121, 261-267)
            double are[NCOMP], aim[NCOMP];
            if constexpr (NCOMP == 1) {
                are[0] = wj; aim[0] = 0.;
            } else {
                const double ig = inv_gamma[ip];
                are[0] = wj * c_light * ig * (cs * ux[ip] + sn * uy[ip]); aim[0] = 0.;
                are[1] = wj * c_light * ig * (cs * uy[ip] - sn * ux[ip]); aim[1] = 0.;
                are[2] = wj * c_light * ig * uz[ip]; aim[2] = 0.;
            }
            for (int m = 0; m < m0; m++) {
#pragma unroll
                for (int k = 0; k < NCOMP; k++) {
                    double re = cs * are[k] - sn * aim[k], im = cs * aim[k] + sn * are[k];
                    are[k] = re; aim[k] = im;
                }
            }
            const double r_cell = invdr * (rj - rmin) - 0.5;
            const double z_cell = invdz * (zj - zmin) - 0.5;
            const int icr = (int)ceil(r_cell), icz = (int)ceil(z_cell);
            int ir_low, iz_low;                        // lowest node (unfolded)
            if constexpr (SHAPE == FB_SHAPE_LINEAR) { ir_low = min(icr - 1, Nr); iz_low = icz - 1; }
            else { ir_low = min(icr, Nr) - 2; iz_low = icz - 2; }
            const int ir_ruy = min(icr, Nr);
            const int ir_shape = icr - H;              // particle_shapes: ir of index 0
            double Sz[S], Sr0[S], Srh[S];
            shape_z<SHAPE>(z_cell, Sz);
            if (m0 == 0) shape_r<SHAPE>(r_cell, beta0[ir_ruy], Sr0);
            if (m0 + NM > 1) shape_r<SHAPE>(r_cell, betah[ir_ruy], Srh);
            // tile-local row of the lowest node; z is periodic inside every kernel
            int lz = iz_low - nz_org;
            if (lz < 0) lz += Nz; else if (lz >= Nz) lz -= Nz;
            const int lr = ir_low - nr_org;
            const bool in_tile = (lz >= 0) && (lz + S <= NTZ) && (lr >= 0) && (lr + S <= NTR);
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
                const int m = m0 + mm;
                const double flip = m1pow(m);
                const double *Srm = (m == 0) ? Sr0 : Srh;
#pragma unroll
                for (int k = 0; k < NCOMP; k++) {
                    // rho, Jz: (-1)^m ; Jr, Jt: -(-1)^m (threading_methods.py:143-146, 289-302)
                    const double fl = (NCOMP == 1 || k == 2) ? flip : -flip;
#pragma unroll
                    for (int jr = 0; jr < S; jr++) {
                        double sr = Srm[jr];
                        if (jr + ir_shape < 0) sr *= fl;
#pragma unroll
                        for (int jz = 0; jz < S; jz++) {
                            const double Sw = Sz[jz] * sr;
                            const double vr = Sw * are[k], vi = Sw * aim[k];
                            if (in_tile) {
                                double *t = lds + (long)((k * NM + mm) * 2) * plane +
                                            (lz + jz) * NTR + (lr + jr);
                                atomicAdd(t, vr);
                                atomicAdd(t + plane, vi);
                            } else {
                                int gz = iz_low + jz, gr = ir_low + jr;
                                fold_node(gz, gr, Nz, Nr);
                                cplx *g = G.g[k + NCOMP * m] + (long)gz * rs + gr;
                                atomicAdd(&g->re, vr);
                                atomicAdd(&g->im, vi);
                            }
                        }
                    }
                    // next mode amplitude
                    double re = cs * are[k] - sn * aim[k], im = cs * aim[k] + sn * are[k];
                    are[k] = re; aim[k] = im;
                }
            }
        }
    }
    __syncthreads();
    // flush the privatised tile, folding the deposition guard nodes
    for (int i = threadIdx.x; i < NK * plane; i += blockDim.x) {
        const double v = lds[i];
        if (v == 0.) continue;
        const int kk = i / plane, rem = i - kk * plane;
        const int tz = rem / NTR, tr = rem - tz * NTR;
        int gz = nz_org + tz, gr = nr_org + tr;
        fold_node(gz, gr, Nz, Nr);
        const int comp_mode = kk >> 1;                 // k*NM + mm
        const int k = comp_mode / NM, mm = comp_mode - k * NM;
        double *g = (double *)(G.g[k + NCOMP * (m0 + mm)] + (long)gz * rs + gr) + (kk & 1);
        atomicAdd(g, v);
    }
}

struct TilePlan { int TZ, TR, n_ztiles, n_rtiles; size_t lds_bytes; };

static TilePlan plan_tiles(int S, int NK, int Nz, int Nr)
{
    // Target <= 48 KiB of LDS per workgroup (3 workgroups / CU) and >= ~1000 workgroups
    // at the headline size so all 256 CUs stay busy.
    TilePlan p;
    p.TR = 32; p.TZ = 4;
    if (p.TR > Nr + 1) p.TR = Nr + 1;
    if (p.TZ > Nz) p.TZ = Nz;
    auto bytes = [&](int tz, int tr) { return (size_t)NK * (tz + S - 1) * (tr + S) * 8; };
    while (bytes(p.TZ, p.TR) > 48 * 1024 && p.TZ > 1) p.TZ--;
    while (bytes(p.TZ, p.TR) > 48 * 1024 && p.TR > 4) p.TR /= 2;
    p.n_ztiles = (Nz + p.TZ - 1) / p.TZ;
    p.n_rtiles = (Nr + 1 + p.TR - 1) / p.TR;
    p.lds_bytes = bytes(p.TZ, p.TR);
    return p;
}

template <int SHAPE, int NCOMP, int NM>
static int launch_one(long n, const double *x, const double *y, const double *z, const double *w,
        double q, const double *ux, const double *uy, const double *uz, const double *ig,
        double c, double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
        const DepGrids &G, long rs, int m0, const int *prefix, const double *b0,
        const double *bh, hipStream_t s)
{
    constexpr int S = ShapeTraits<SHAPE>::S;
    TilePlan p = plan_tiles(S, NCOMP * NM * 2, Nz, Nr);
    auto kern = k_deposit<SHAPE, NCOMP, NM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return check(e, "fb_deposit(attr)");
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.n_ztiles * p.n_rtiles), dim3(256), p.lds_bytes, s, n, x, y, z,
                       w, q, ux, uy, uz, ig, c, invdz, zmin, Nz, invdr, rmin, Nr, G, rs, m0,
                       prefix, b0, bh, p.TZ, p.TR, p.n_rtiles);
    return check(hipGetLastError(), "fb_deposit");
}

template <int SHAPE, int NCOMP>
static int launch_modes(int Nm, long n, const double *x, const double *y, const double *z,
        const double *w, double q, const double *ux, const double *uy, const double *uz,
        const double *ig, double c, double invdz, double zmin, int Nz, double invdr, double rmin,
        int Nr, const DepGrids &G, long rs, const int *prefix, const double *b0, const double *bh,
        hipStream_t s)
{
    int m0 = 0;
    while (m0 < Nm) {
        int left = Nm - m0, r;
#define ARGS n, x, y, z, w, q, ux, uy, uz, ig, c, invdz, zmin, Nz, invdr, rmin, Nr, G, rs, m0, prefix, b0, bh, s
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
        const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh, void *stream)
{
    if (n <= 0) return 0;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error("fb_deposit_rho", "Nm out of range"); return -1; }
    DepGrids G;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) G.g[i] = i < Nm ? (cplx *)rho[i] : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (shape == FB_SHAPE_LINEAR)
        return launch_modes<FB_SHAPE_LINEAR, 1>(Nm, n, x, y, z, w, q, nullptr, nullptr, nullptr,
                nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, prefix_sum,
                ruyten_m0, ruyten_mh, s);
    if (shape == FB_SHAPE_CUBIC)
        return launch_modes<FB_SHAPE_CUBIC, 1>(Nm, n, x, y, z, w, q, nullptr, nullptr, nullptr,
                nullptr, 0., invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, prefix_sum,
                ruyten_m0, ruyten_mh, s);
    set_error("fb_deposit_rho", "unknown shape");
    return -1;
}

extern "C" int fb_deposit_J(int shape, int Nm, long n, const double *x, const double *y,
        const double *z, const double *w, double q, const double *ux, const double *uy,
        const double *uz, const double *inv_gamma, double c, double invdz, double zmin, int Nz,
        double invdr, double rmin, int Nr, void *const *J, long row_stride,
        const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh, void *stream)
{
    if (n <= 0) return 0;
    if (Nm < 1 || Nm > FB_MAX_MODES) { set_error("fb_deposit_J", "Nm out of range"); return -1; }
    DepGrids G;
    for (int i = 0; i < 3 * FB_MAX_MODES; i++) G.g[i] = i < 3 * Nm ? (cplx *)J[i] : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (shape == FB_SHAPE_LINEAR)
        return launch_modes<FB_SHAPE_LINEAR, 3>(Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c,
                invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, prefix_sum, ruyten_m0,
                ruyten_mh, s);
    if (shape == FB_SHAPE_CUBIC)
        return launch_modes<FB_SHAPE_CUBIC, 3>(Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c,
                invdz, zmin, Nz, invdr, rmin, Nr, G, row_stride, prefix_sum, ruyten_m0,
                ruyten_mh, s);
    set_error("fb_deposit_J", "unknown shape");
    return -1;
}
