#!/usr/bin/env python3
"""Golden-vector capture from the REAL reference (build container only).

TEST INFRASTRUCTURE - not part of the product.  Run as

    cd /tmp && PYTHONPATH=/root/repo/oracle/shim:/root/reference \
        python3 -W ignore /root/repo/oracle/capture_golden.py [names...]

The pass-through `numba`/`pyfftw`/`h5py` stand-ins in oracle/shim let the
reference's plain-Python CPU kernels run interpreted (1 thread).  The outputs
are DATA (seeded inputs + the reference's outputs) written as small .npz files
under tests/golden/.  Neither the reference nor any transformed copy of it is
stored.  Nothing on the GPU box may import this file (it needs /root/reference).
"""
import os
import sys
import numpy as np
from scipy.constants import c, e, m_e, epsilon_0

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
OUT = os.path.normpath(OUT)
os.makedirs(OUT, exist_ok=True)


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote', path, '%.1f kB' % (os.path.getsize(path) / 1e3))


# ---------------------------------------------------------------- particles
def make_positions(rng, n, zmin, zmax, rmax, dz, dr):
    """Random particles incl. hand-placed edge cases (axis, r>rmax, z edges)."""
    z = rng.uniform(zmin - dz, zmax + dz, n)
    r = rng.uniform(0., 1.1 * rmax, n)
    th = rng.uniform(0, 2 * np.pi, n)
    x = r * np.cos(th)
    y = r * np.sin(th)
    # edge cases
    x[0] = 0.; y[0] = 0.                       # exactly on axis
    x[1] = 0.1 * dr; y[1] = 0.                 # first half cell
    x[2] = 0.; y[2] = -0.3 * dr
    x[3] = rmax - 0.25 * dr; y[3] = 0.         # last half cell
    x[4] = 0.; y[4] = rmax + 0.3 * dr          # beyond rmax
    x[5] = -(rmax + 1.7 * dr); y[5] = 0.
    z[6] = zmin + 0.2 * dz                     # first half cell in z
    z[7] = zmax - 0.2 * dz                     # last half cell in z
    z[8] = zmin - 0.7 * dz
    z[9] = zmax + 0.7 * dz
    x[10] = 0.49 * dr; y[10] = 0.; z[10] = zmin + 0.49 * dz
    x[11] = 1.51 * dr; y[11] = 0.; z[11] = zmax - 0.51 * dz
    return x, y, z


def cap_push():
    from fbpic.particles.push.numba_methods import push_p_numba, push_x_numba
    rng = np.random.default_rng(101)
    n = 4096
    dt = 6.67e-16
    scale = np.repeat(np.array([1e-3, 1., 100.]), [1366, 1365, 1365])
    ux = rng.normal(size=n) * scale
    uy = rng.normal(size=n) * scale
    uz = rng.normal(size=n) * scale
    ig = 1. / np.sqrt(1 + ux**2 + uy**2 + uz**2)
    E = rng.normal(size=(3, n)) * 1e11
    B = rng.normal(size=(3, n)) * 1e3
    x = rng.normal(size=n) * 1e-5
    y = rng.normal(size=n) * 1e-5
    z = rng.normal(size=n) * 1e-5
    inp = dict(ux=ux.copy(), uy=uy.copy(), uz=uz.copy(), inv_gamma=ig.copy(),
               Ex=E[0], Ey=E[1], Ez=E[2], Bx=B[0], By=B[1], Bz=B[2],
               x=x.copy(), y=y.copy(), z=z.copy())
    out = {}
    for tag, q, m in [('e', -e, m_e), ('p', e, 1836.15267 * m_e)]:
        a = [inp[k].copy() for k in ('ux', 'uy', 'uz', 'inv_gamma')]
        push_p_numba(a[0], a[1], a[2], a[3], E[0], E[1], E[2], B[0], B[1], B[2],
                     q, m, n, dt)
        for k, v in zip(('ux', 'uy', 'uz', 'inv_gamma'), a):
            out['pp_%s_%s' % (tag, k)] = v
        if tag == 'e':
            xx, yy, zz = x.copy(), y.copy(), z.copy()
            push_x_numba(xx, yy, zz, a[0], a[1], a[2], a[3], n, 0.5 * dt, 1., 1., 1.)
            out['px_x'], out['px_y'], out['px_z'] = xx, yy, zz
            xx, yy, zz = x.copy(), y.copy(), z.copy()
            push_x_numba(xx, yy, zz, a[0], a[1], a[2], a[3], n, dt, -1., -1., 1.)
            out['pxm_x'], out['pxm_y'], out['pxm_z'] = xx, yy, zz
    save('push', dt=dt, m_p=1836.15267 * m_e, **{'in_' + k: v for k, v in inp.items()}, **out)


def cap_gather():
    from fbpic.particles.gathering.threading_methods import \
        gather_field_numba_linear, gather_field_numba_cubic
    from fbpic.particles.gathering.threading_methods_one_mode import \
        erase_eb_numba, gather_field_numba_linear_one_mode, \
        gather_field_numba_cubic_one_mode
    rng = np.random.default_rng(202)
    Nz, Nr, Nm = 32, 16, 4
    dz, dr = 0.25e-6, 0.5e-6
    zmin = -2e-6
    zmax = zmin + Nz * dz
    rmax = Nr * dr
    n = 1536
    x, y, z = make_positions(rng, n, zmin, zmax, rmax, dz, dr)
    grids = (rng.normal(size=(Nm, 6, Nz, Nr)) + 1j * rng.normal(size=(Nm, 6, Nz, Nr)))
    rmax_gather = rmax
    res = {}
    chunk = np.array([0, n], dtype=np.int64)
    for shape in ('linear', 'cubic'):
        F = [np.zeros(n) for _ in range(6)]
        g = grids
        args = (x, y, z, rmax_gather, 1. / dz, zmin, Nz, 1. / dr, 0., Nr,
                g[0, 0], g[0, 1], g[0, 2], g[1, 0], g[1, 1], g[1, 2],
                g[0, 3], g[0, 4], g[0, 5], g[1, 3], g[1, 4], g[1, 5],
                F[0], F[1], F[2], F[3], F[4], F[5])
        if shape == 'linear':
            gather_field_numba_linear(*args)
        else:
            gather_field_numba_cubic(*args, 1, chunk)
        res['%s_nm2' % shape] = np.array(F)
        for nm in (1, 3, 4):
            F = [np.full(n, 7.) for _ in range(6)]
            erase_eb_numba(*F, n)
            for m in range(nm):
                a = (x, y, z, rmax_gather, 1. / dz, zmin, Nz, 1. / dr, 0., Nr,
                     g[m, 0], g[m, 1], g[m, 2], g[m, 3], g[m, 4], g[m, 5], m,
                     F[0], F[1], F[2], F[3], F[4], F[5])
                if shape == 'linear':
                    gather_field_numba_linear_one_mode(*a)
                else:
                    gather_field_numba_cubic_one_mode(*a, 1, chunk)
            res['%s_nm%d_onemode' % (shape, nm)] = np.array(F)
    save('gather', x=x, y=y, z=z, grids=grids, Nz=Nz, Nr=Nr, dz=dz, dr=dr, zmin=zmin,
         rmax_gather=rmax_gather, **res)


def cap_deposit():
    """Deposition + the reference's own target cell for every particle."""
    from fbpic.particles.deposition.threading_methods import \
        deposit_rho_numba_linear, deposit_rho_numba_cubic, \
        deposit_J_numba_linear, deposit_J_numba_cubic
    from fbpic.fields.numba_methods import sum_reduce_2d_array
    from fbpic.fields.interpolation_grid import InterpolationGrid
    rng = np.random.default_rng(303)
    Nz, Nr = 24, 12
    dz, dr = 0.25e-6, 0.5e-6
    zmin = 1e-6
    zmax = zmin + Nz * dz
    rmax = Nr * dr
    n = 1200
    x, y, z = make_positions(rng, n, zmin, zmax, rmax, dz, dr)
    # cubic deposition writes rows ceil(z_cell)..ceil(z_cell)+3 of a (Nz+4) array:
    # particles must sit within (zmin - dz/2, zmax + dz/2] -> clip the far ones
    z = np.clip(z, zmin - 0.49 * dz, zmax + 0.49 * dz)
    w = rng.uniform(0.5, 1.5, n) * 1e5
    ux = rng.normal(size=n); uy = rng.normal(size=n); uz = rng.normal(size=n)
    ig = 1. / np.sqrt(1 + ux**2 + uy**2 + uz**2)
    q = -e
    chunk = np.array([0, n], dtype=np.int64)
    res = {}
    grid = [InterpolationGrid(Nz, Nr, m, zmin, zmax, rmax) for m in range(2)]
    grid_nor = [InterpolationGrid(Nz, Nr, m, zmin, zmax, rmax, use_ruyten_shapes=False,
                                  use_modified_volume=False) for m in range(2)]
    res['invvol_m0'] = grid[0].invvol; res['invvol_m1'] = grid[1].invvol
    res['ruy_lin_m0'] = grid[0].ruyten_linear_coef; res['ruy_lin_m1'] = grid[1].ruyten_linear_coef
    res['ruy_cub_m0'] = grid[0].ruyten_cubic_coef; res['ruy_cub_m1'] = grid[1].ruyten_cubic_coef
    res['invvol_std_m0'] = grid_nor[0].invvol
    for shape in ('linear', 'cubic'):
        for ruy in (True, False):
            gr = grid if ruy else grid_nor
            for Nm in (1, 2, 4):
                b0 = getattr(gr[0], 'ruyten_%s_coef' % shape)
                b1 = getattr(gr[1 if Nm > 1 else 0], 'ruyten_%s_coef' % shape)
                glob = np.zeros((1, Nm, Nz + 4, Nr + 4), dtype=np.complex128)
                fr = deposit_rho_numba_linear if shape == 'linear' else deposit_rho_numba_cubic
                fr(x, y, z, w, q, 1. / dz, zmin, Nz, 1. / dr, 0., Nr, glob, Nm, 1, chunk, b0, b1)
                red = np.zeros((Nm, Nz, Nr), dtype=np.complex128)
                for m in range(Nm):
                    sum_reduce_2d_array(glob, red[m], m)
                tag = '%s_r%d_nm%d' % (shape, int(ruy), Nm)
                res['rho_' + tag] = red
                if Nm == 2 and ruy:
                    res['rho_glob_' + tag] = glob[0]
                gJ = [np.zeros((1, Nm, Nz + 4, Nr + 4), dtype=np.complex128) for _ in range(3)]
                fj = deposit_J_numba_linear if shape == 'linear' else deposit_J_numba_cubic
                fj(x, y, z, w, q, ux, uy, uz, ig, 1. / dz, zmin, Nz, 1. / dr, 0., Nr,
                   gJ[0], gJ[1], gJ[2], Nm, 1, chunk, b0, b1)
                redJ = np.zeros((3, Nm, Nz, Nr), dtype=np.complex128)
                for k in range(3):
                    for m in range(Nm):
                        sum_reduce_2d_array(gJ[k], redJ[k, m], m)
                res['J_' + tag] = redJ
    # The reference's own lowest deposition cell for each particle (linear):
    # deposit one particle at a time with beta=0 and find the touched corner.
    zero = np.zeros(Nr + 1)
    izc = np.zeros(n, dtype=np.int64); irc = np.zeros(n, dtype=np.int64)
    one = np.array([0, 1], dtype=np.int64)
    for i in range(n):
        glob = np.zeros((1, 1, Nz + 4, Nr + 4), dtype=np.complex128)
        # unit weight so every touched node is visible; use |.| of S*S
        deposit_rho_numba_linear(x[i:i+1], y[i:i+1], z[i:i+1], np.ones(1), 1.,
                                 1. / dz, zmin, Nz, 1. / dr, 0., Nr, glob, 1, 1, one, zero, zero)
        nzr = np.argwhere(glob[0, 0] != 0)
        # upper node = ceil(cell); padded index = ceil+2 ; lowest touched = ceil+1
        # (a particle exactly on a node touches a single row/col: S_lower == 0)
        izc[i] = nzr[:, 0].max() - 2
        irc[i] = nzr[:, 1].max() - 2
    res['iz_upper_unwrapped'] = izc
    res['ir_upper_clamped_plus'] = irc   # = min(ceil(r_cell), Nr+1): column index of upper node
    save('deposit', x=x, y=y, z=z, w=w, ux=ux, uy=uy, uz=uz, inv_gamma=ig, q=q,
         Nz=Nz, Nr=Nr, dz=dz, dr=dr, zmin=zmin, **res)


def cap_grid_setup():
    from fbpic.fields.spectral_transform.hankel import DHT
    from fbpic.fields.interpolation_grid import InterpolationGrid
    from fbpic.fields.utility_methods import get_modified_k, get_stencil_reach
    from fbpic.fields import Fields
    res = {}
    for Nr in (16, 32):
        rmax = Nr * 0.5e-6
        for m in range(4):
            for p in (m - 1, m, m + 1):
                d = DHT(p, m, Nr, 4, rmax)
                tag = 'Nr%d_m%d_p%d' % (Nr, m, p - m + 1)
                res['M_' + tag] = d.M
                res['invM_' + tag] = d.invM
                res['nu_' + tag] = d.nu
    for Nr in (16, 128):
        g0 = InterpolationGrid(8, Nr, 0, 0., 8 * 0.2e-6, Nr * 0.2e-6)
        g1 = InterpolationGrid(8, Nr, 1, 0., 8 * 0.2e-6, Nr * 0.2e-6)
        res['invvol_Nr%d_m0' % Nr] = g0.invvol
        res['invvol_Nr%d_m1' % Nr] = g1.invvol
        res['ruyl_Nr%d_m0' % Nr] = g0.ruyten_linear_coef
        res['ruyl_Nr%d_m1' % Nr] = g1.ruyten_linear_coef
        res['ruyc_Nr%d_m0' % Nr] = g0.ruyten_cubic_coef
        res['ruyc_Nr%d_m1' % Nr] = g1.ruyten_cubic_coef
    kz = 2 * np.pi * np.fft.fftfreq(64, 0.1e-6)
    for n_order in (8, 16, 32):
        res['kzmod_%d' % n_order] = get_modified_k(kz, n_order, 0.1e-6)
        res['reach_%d' % n_order] = get_stencil_reach(1024, 0.2e-6, 0.2e-6, n_order, None, False)
    Nz, Nr, Nm = 32, 16, 3
    dt = 0.25e-6 / c
    for n_order in (-1, 16):
        f = Fields(Nz, Nz * 0.25e-6, Nr, Nr * 0.5e-6, Nm, dt, n_order=n_order, zmin=0.,
                   current_correction='curl-free')
        for m in range(Nm):
            t = 'o%d_m%d' % (n_order, m)
            res['C_' + t] = f.psatd[m].C
            res['S_w_' + t] = f.psatd[m].S_w
            res['j_coef_' + t] = f.psatd[m].j_coef
            res['rho_prev_coef_' + t] = f.psatd[m].rho_prev_coef
            res['rho_next_coef_' + t] = f.psatd[m].rho_next_coef
            res['kz_' + t] = f.spect[m].kz[:, 0]
            res['kr_' + t] = f.spect[m].kr[0, :]
            res['inv_k2_' + t] = f.spect[m].inv_k2
            res['filter_z_' + t] = f.spect[m].filter_array_z
            res['filter_r_' + t] = f.spect[m].filter_array_r
    save('grid_setup', **res)


def cap_spectral():
    from fbpic.fields import Fields
    from fbpic.fields.numba_methods import numba_push_eb_standard, \
        numba_correct_currents_curlfree_standard, numba_filter_scalar, numba_filter_vector
    rng = np.random.default_rng(404)
    Nz, Nr, Nm = 32, 16, 3
    dz, dr = 0.25e-6, 0.5e-6
    dt = dz / c
    f = Fields(Nz, Nz * dz, Nr, Nr * dr, Nm, dt, n_order=-1, zmin=0., current_correction='curl-free')
    res = {}

    def rc():
        return rng.normal(size=(Nz, Nr)) + 1j * rng.normal(size=(Nz, Nr))
    for m in range(Nm):
        tr = f.trans[m]
        a = rc(); b = rc(); t_ = rc()
        res['in_scal_m%d' % m] = a
        res['in_r_m%d' % m] = b
        res['in_t_m%d' % m] = t_
        o = np.zeros((Nz, Nr), complex)
        tr.interp2spect_scal(a, o); res['i2s_scal_m%d' % m] = o.copy()
        tr.spect2interp_scal(a, o); res['s2i_scal_m%d' % m] = o.copy()
        op = np.zeros((Nz, Nr), complex); om = np.zeros((Nz, Nr), complex)
        tr.interp2spect_vect(b, t_, op, om)
        res['i2s_p_m%d' % m] = op.copy(); res['i2s_m_m%d' % m] = om.copy()
        tr.spect2interp_vect(b, t_, op, om)
        res['s2i_r_m%d' % m] = op.copy(); res['s2i_t_m%d' % m] = om.copy()
        tr.fft.transform(a, o); res['fft_m%d' % m] = o.copy()
        tr.fft.inverse_transform(a, o); res['ifft_m%d' % m] = o.copy()
        # spectral kernels
        sp = f.spect[m]; ps = f.psatd[m]
        names = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']
        scale = dict(E=1e9, B=3., J=1e12, r=1e4)
        arrs = {k: rc() * scale[k[0]] for k in names}
        for k in names:
            res['sp_in_%s_m%d' % (k, m)] = arrs[k].copy()
        # correct currents
        cc = {k: arrs[k].copy() for k in names}
        numba_correct_currents_curlfree_standard(
            cc['rho_prev'], cc['rho_next'], cc['Jp'], cc['Jm'], cc['Jz'],
            sp.kz, sp.kr, sp.inv_k2, 1. / dt, Nz, Nr)
        for k in ('Jp', 'Jm', 'Jz'):
            res['cc_%s_m%d' % (k, m)] = cc[k]
        for utr in (False, True):
            pe = {k: arrs[k].copy() for k in names}
            numba_push_eb_standard(
                pe['Ep'], pe['Em'], pe['Ez'], pe['Bp'], pe['Bm'], pe['Bz'],
                pe['Jp'], pe['Jm'], pe['Jz'], pe['rho_prev'], pe['rho_next'],
                ps.rho_prev_coef, ps.rho_next_coef, ps.j_coef, ps.C, ps.S_w,
                sp.kr, sp.kz, ps.dt, utr, Nz, Nr)
            for k in names[:6]:
                res['pe%d_%s_m%d' % (int(utr), k, m)] = pe[k]
        fl = {k: arrs[k].copy() for k in names}
        numba_filter_vector(fl['Jp'], fl['Jm'], fl['Jz'], Nz, Nr,
                            sp.filter_array_z, sp.filter_array_r)
        numba_filter_scalar(fl['rho_next'], Nz, Nr, sp.filter_array_z, sp.filter_array_r)
        for k in ('Jp', 'Jm', 'Jz', 'rho_next'):
            res['fl_%s_m%d' % (k, m)] = fl[k]
    save('spectral', Nz=Nz, Nr=Nr, Nm=Nm, dz=dz, dr=dr, dt=dt, **res)


# ---------------------------------------------------------------- whole cycle
INTERP = ['Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz', 'Jr', 'Jt', 'Jz', 'rho']
SPECT = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']
PTCL = ['x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w', 'Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz']


def snapshot(sim, tag, res, spect=True):
    Nm = sim.fld.Nm
    res[tag + '_interp'] = np.array([[getattr(sim.fld.interp[m], k) for k in INTERP]
                                     for m in range(Nm)])
    if spect:
        res[tag + '_spect'] = np.array([[getattr(sim.fld.spect[m], k) for k in SPECT]
                                        for m in range(Nm)])
    for isp, sp in enumerate(sim.ptcl):
        arr = np.array([getattr(sp, k) for k in PTCL])
        if tag == 's0':
            # Before the first gather the reference's Ex .. Bz are np.empty (uninitialised
            # memory, overwritten by that gather and never an input): stored as zeros so that
            # the fixture regenerates bit for bit
            arr[8:14] = 0.
        res['%s_ptcl%d' % (tag, isp)] = arr


def cap_cycle():
    import importlib
    sys.path.insert(0, '/root/reference/tests')
    T = importlib.import_module('test_periodic_plasma_wave')
    from fbpic.main import Simulation
    for name, Nz, Nr, Nm, shape, n_order, ppc in [
            ('cycle_lin_16x8_nm2', 16, 8, 2, 'linear', -1, (2, 2, 4)),
            ('cycle_cub_16x8_nm2', 16, 8, 2, 'cubic', -1, (2, 2, 4)),
            ('cycle_lin_32x16_nm3', 32, 16, 3, 'linear', 8, (1, 2, 8)),
            ('cycle_cub_32x16_nm2_ions', 32, 16, 2, 'cubic', -1, (1, 2, 4))]:
        dz = 0.2e-6
        zmax = Nz * dz
        rmax = Nr * 0.3125e-6
        dt = dz / c
        np.random.seed(7)
        ions = name.endswith('ions')
        sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, 0., zmax, 0., 0.9 * rmax,
                         ppc[0], ppc[1], ppc[2], 2.e24, n_order=n_order,
                         particle_shape=shape, verbose_level=0, initialize_ions=ions,
                         n_guard=(None if n_order == -1 else 8))
        k0 = 2 * np.pi / zmax
        wp = np.sqrt(2.e24 * e**2 / (m_e * epsilon_0))
        T.impart_momenta(sim.ptcl[0], [0.01, 0.01, 0.01], k0, 0.4 * rmax, wp)
        res = dict(Nz=Nz, Nr=Nr, Nm=Nm, zmax=zmax, rmax=rmax, dt=dt, n_order=n_order,
                   shape=shape, n_species=len(sim.ptcl),
                   q=np.array([s.q for s in sim.ptcl]), m=np.array([s.m for s in sim.ptcl]))
        snapshot(sim, 's0', res, spect=False)
        done = 0
        for upto in (1, 2, 5):
            sim.step(upto - done, show_progress=False, use_true_rho=ions)
            done = upto
            snapshot(sim, 's%d' % upto, res)
        res['use_true_rho'] = ions
        save(name, **res)


def cap_bunch():
    """Counterpart of tests/test_cpu_gpu_deposition.py (arrays from memory)."""
    from fbpic.main import Simulation
    from fbpic.lpa_utils.bunch import add_elec_bunch_gaussian
    Nz, zmax, zmin, Nr, rmax, Nm = 100, 30.e-6, -10.e-6, 50, 20.e-6, 2
    dt = (zmax - zmin) / Nz / c
    for shape in ('linear', 'cubic'):
        sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, particle_shape=shape,
                         verbose_level=0)
        sim.ptcl = []
        np.random.seed(0)
        add_elec_bunch_gaussian(sim, 20.e-6, 10.e-6, 10.e-6, 10, 0., 10.e-12, 2000)
        res = dict(Nz=Nz, Nr=Nr, Nm=Nm, zmin=zmin, zmax=zmax, rmax=rmax, dt=dt,
                   q=np.array([s.q for s in sim.ptcl]), m=np.array([s.m for s in sim.ptcl]))
        snapshot(sim, 's0', res, spect=False)
        for it in (1, 2, 3):
            sim.step(1, show_progress=False)
            Nm_ = sim.fld.Nm
            res['s%d_JrJtJzrho' % it] = np.array(
                [[getattr(sim.fld.interp[m], k) for k in ('Jr', 'Jt', 'Jz', 'rho')]
                 for m in range(Nm_)])
            res['s%d_ptcl0' % it] = np.array([getattr(sim.ptcl[0], k) for k in PTCL[:8]])
        save('bunch_' + shape, **res)


def cap_lwfa():
    """"Next" rows: open z boundary + damping, moving window, continuous
    injection, Gaussian laser initialised on the grid -- a miniature of
    docs/source/example_input/lwfa_script.py."""
    from fbpic.main import Simulation
    from fbpic.lpa_utils.laser import add_laser_pulse, GaussianLaser
    for shape in ('linear', 'cubic'):
        Nz, Nr, Nm = 96, 24, 2
        zmax, zmin, rmax = 12.e-6, -12.e-6, 12.e-6
        dt = (zmax - zmin) / Nz / c
        np.random.seed(11)
        sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin,
                         p_zmin=2.e-6, p_zmax=1., p_rmin=0., p_rmax=10.e-6, p_nz=1, p_nr=2, p_nt=4,
                         n_e=4.e24, n_order=-1, particle_shape=shape, verbose_level=0,
                         boundaries={'z': 'open', 'r': 'reflective'}, n_guard=16,
                         n_damp={'z': 16, 'r': 8}, exchange_period=4)
        prof = GaussianLaser(a0=1.5, waist=4.e-6, tau=8.e-15, z0=0.e-6, zf=4.e-6,
                             lambda0=0.8e-6, theta_pol=0.3, cep_phase=0.4)
        add_laser_pulse(sim, prof)
        sim.set_moving_window(v=c)
        res = dict(Nz=Nz, Nr=Nr, Nm=Nm, zmin=zmin, zmax=zmax, rmax=rmax, dt=dt, shape=shape,
                   Nz_local=sim.fld.Nz, n_guard=sim.comm.n_guard, n_inject=sim.comm.n_inject,
                   nz_damp=sim.comm.nz_damp)
        def snap(tag, ptcl=True):
            res[tag + '_interp'] = np.array([[getattr(sim.fld.interp[m], k) for k in INTERP]
                                             for m in range(Nm)])
            res[tag + '_zmin'] = sim.fld.interp[0].zmin
            if ptcl:
                res[tag + '_ptcl0'] = np.array([getattr(sim.ptcl[0], k) for k in PTCL[:8]])
        snap('s0')
        done = 0
        for upto in (6, 14):
            sim.step(upto - done, show_progress=False)
            done = upto
            snap('s%d' % upto)
        save('lwfa_' + shape, **res)


def cap_galilean():
    """Galilean / comoving-current PSATD: coefficient tables, the two
    spectral kernels, and short whole-cycle trajectories of a drifting periodic plasma."""
    from fbpic.fields import Fields
    from fbpic.fields.numba_methods import numba_push_eb_comoving, \
        numba_correct_currents_curlfree_comoving
    from fbpic.main import Simulation
    rng = np.random.default_rng(505)
    Nz, Nr, Nm = 32, 16, 2
    dz, dr = 0.25e-6, 0.5e-6
    dt = dz / c
    names = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']
    res = dict(Nz=Nz, Nr=Nr, Nm=Nm, dz=dz, dr=dr, dt=dt)
    for tag, V, gal in (('gal', 0.9 * c, True), ('com', -0.5 * c, False), ('gal0', 0., True)):
        f = Fields(Nz, Nz * dz, Nr, Nr * dr, Nm, dt, n_order=-1, zmin=0.,
                   current_correction='curl-free', v_comoving=V, use_galilean=gal)
        res['%s_V' % tag] = V
        for m in range(Nm):
            sp, ps = f.spect[m], f.psatd[m]
            for k in ('C', 'S_w', 'j_coef', 'rho_prev_coef', 'rho_next_coef', 'T_eb', 'T_cc',
                      'T_rho', 'j_corr_coef'):
                res['%s_%s_m%d' % (tag, k, m)] = np.asarray(getattr(ps, k))
            for k in ('kz', 'kr', 'inv_k2'):
                res['%s_%s_m%d' % (tag, k, m)] = np.asarray(getattr(sp, k))
            scale = dict(E=1e9, B=3., J=1e12, r=1e4)
            arrs = {k: (rng.normal(size=(Nz, Nr)) + 1j * rng.normal(size=(Nz, Nr))) * scale[k[0]]
                    for k in names}
            for k in names:
                res['%s_in_%s_m%d' % (tag, k, m)] = arrs[k].copy()
            cc = {k: arrs[k].copy() for k in names}
            numba_correct_currents_curlfree_comoving(
                cc['rho_prev'], cc['rho_next'], cc['Jp'], cc['Jm'], cc['Jz'], sp.kz, sp.kr,
                sp.inv_k2, ps.j_corr_coef, ps.T_eb, ps.T_cc, 1. / dt, Nz, Nr)
            for k in ('Jp', 'Jm', 'Jz'):
                res['%s_cc_%s_m%d' % (tag, k, m)] = cc[k]
            for utr in (False, True):
                pe = {k: arrs[k].copy() for k in names}
                numba_push_eb_comoving(
                    pe['Ep'], pe['Em'], pe['Ez'], pe['Bp'], pe['Bm'], pe['Bz'], pe['Jp'], pe['Jm'],
                    pe['Jz'], pe['rho_prev'], pe['rho_next'], ps.rho_prev_coef, ps.rho_next_coef,
                    ps.j_coef, ps.C, ps.S_w, ps.T_eb, ps.T_cc, ps.T_rho, sp.kr, sp.kz, ps.dt,
                    ps.V, utr, Nz, Nr)
                for k in names[:6]:
                    res['%s_pe%d_%s_m%d' % (tag, int(utr), k, m)] = pe[k]
    save('galilean_kernels', **res)
    # whole cycle: periodic box, plasma drifting at uz_m, grid following it (Galilean) or
    # currents assumed comoving
    # (comoving currents without a Galilean grid: the plasma crosses ~1 cell of the periodic
    # box per step, which the reference's cubic deposition only tolerates up to half a cell
    # beyond the box -> linear shape there, cubic with the Galilean grid)
    for name, shape, gal, n_order in (('cycle_galilean_cub_16x8', 'cubic', True, -1),
                                      ('cycle_comoving_lin_16x8', 'linear', False, -1),
                                      ('cycle_galilean_lin_32x8_o8', 'linear', True, 8)):
        Nz = 32 if n_order > 0 else 16
        Nr, Nm = 8, 2
        dz = 0.2e-6
        zmax, rmax = Nz * dz, Nr * 0.3125e-6
        dt = dz / c
        gamma_b = 3.
        beta_b = -np.sqrt(1. - 1. / gamma_b**2)
        np.random.seed(9)
        sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, 0., zmax, 0., 0.9 * rmax, 2, 2, 4, 2.e24,
                         n_order=n_order, particle_shape=shape, verbose_level=0,
                         v_comoving=beta_b * c, use_galilean=gal, initialize_ions=False,
                         n_guard=(None if n_order == -1 else 8))
        sp0 = sim.ptcl[0]
        rng2 = np.random.default_rng(2)
        sp0.uz[:] = gamma_b * beta_b + 0.01 * rng2.normal(size=sp0.Ntot)
        sp0.ux[:] = 0.01 * rng2.normal(size=sp0.Ntot)
        sp0.uy[:] = 0.01 * rng2.normal(size=sp0.Ntot)
        sp0.inv_gamma[:] = 1. / np.sqrt(1. + sp0.ux**2 + sp0.uy**2 + sp0.uz**2)
        res = dict(Nz=Nz, Nr=Nr, Nm=Nm, zmax=zmax, rmax=rmax, dt=dt, n_order=n_order, shape=shape,
                   v_comoving=beta_b * c, use_galilean=gal, n_species=1,
                   q=np.array([s.q for s in sim.ptcl]), m=np.array([s.m for s in sim.ptcl]))
        snapshot(sim, 's0', res, spect=False)
        res['s0_zmin'] = sim.fld.interp[0].zmin
        done = 0
        for upto in (1, 2, 5):
            sim.step(upto - done, show_progress=False)
            done = upto
            snapshot(sim, 's%d' % upto, res)
            res['s%d_zmin' % upto] = sim.fld.interp[0].zmin
        save(name, **res)


def cap_crossdep():
    """Cross-deposition current correction: the two spectral kernels on
    random inputs, and whole-cycle trajectories with current_correction='cross-deposition'
    (standard PSATD plasma wave, linear + cubic; Galilean drifting plasma, cubic)."""
    import importlib
    from fbpic.fields import Fields
    from fbpic.fields.numba_methods import numba_correct_currents_crossdeposition_standard, \
        numba_correct_currents_crossdeposition_comoving
    from fbpic.main import Simulation
    rng = np.random.default_rng(606)
    Nz, Nr, Nm = 32, 16, 2
    dz, dr = 0.25e-6, 0.5e-6
    dt = dz / c
    names = ['rho_prev', 'rho_next', 'rho_next_z', 'rho_next_xy', 'Jp', 'Jm', 'Jz']
    res = dict(Nz=Nz, Nr=Nr, Nm=Nm, dz=dz, dr=dr, dt=dt)
    for tag, V, gal in (('std', None, False), ('gal', 0.9 * c, True), ('com', -0.5 * c, False)):
        f = Fields(Nz, Nz * dz, Nr, Nr * dr, Nm, dt, n_order=-1, zmin=0.,
                   current_correction='cross-deposition', v_comoving=V, use_galilean=gal)
        res['%s_V' % tag] = np.nan if V is None else V
        for m in range(Nm):
            sp, ps = f.spect[m], f.psatd[m]
            for k in ('kz', 'kr'):
                res['%s_%s_m%d' % (tag, k, m)] = np.asarray(getattr(sp, k))
            if V is not None:
                for k in ('T_eb', 'T_cc', 'j_corr_coef'):
                    res['%s_%s_m%d' % (tag, k, m)] = np.asarray(getattr(ps, k))
            scale = dict(J=1e12, r=1e4)
            arrs = {k: (rng.normal(size=(Nz, Nr)) + 1j * rng.normal(size=(Nz, Nr))) * scale[k[0]]
                    for k in names}
            for k in names:
                res['%s_in_%s_m%d' % (tag, k, m)] = arrs[k].copy()
            cc = {k: arrs[k].copy() for k in names}
            if V is None:
                numba_correct_currents_crossdeposition_standard(
                    cc['rho_prev'], cc['rho_next'], cc['rho_next_z'], cc['rho_next_xy'],
                    cc['Jp'], cc['Jm'], cc['Jz'], sp.kz, sp.kr, 1. / dt, Nz, Nr)
            else:
                numba_correct_currents_crossdeposition_comoving(
                    cc['rho_prev'], cc['rho_next'], cc['rho_next_z'], cc['rho_next_xy'],
                    cc['Jp'], cc['Jm'], cc['Jz'], sp.kz, sp.kr, ps.j_corr_coef, ps.T_eb,
                    ps.T_cc, 1. / dt, Nz, Nr)
            for k in ('Jp', 'Jm', 'Jz'):
                res['%s_cc_%s_m%d' % (tag, k, m)] = cc[k]
    save('crossdep_kernels', **res)

    sys.path.insert(0, '/root/reference/tests')
    T = importlib.import_module('test_periodic_plasma_wave')
    for name, shape, V in (('cycle_cross_lin_16x8', 'linear', None),
                           ('cycle_cross_cub_16x8', 'cubic', None),
                           ('cycle_cross_galilean_cub_16x8', 'cubic', 'gal')):
        Nz, Nr, Nm = 16, 8, 2
        dz = 0.2e-6
        zmax, rmax = Nz * dz, Nr * 0.3125e-6
        dt = dz / c
        gamma_b = 3.
        beta_b = -np.sqrt(1. - 1. / gamma_b**2)
        np.random.seed(13)
        sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, 0., zmax, 0., 0.9 * rmax, 2, 2, 4, 2.e24,
                         n_order=-1, particle_shape=shape, verbose_level=0,
                         current_correction='cross-deposition',
                         v_comoving=(beta_b * c if V else None), use_galilean=bool(V))
        sp0 = sim.ptcl[0]
        if V:
            rng2 = np.random.default_rng(3)
            sp0.uz[:] = gamma_b * beta_b + 0.01 * rng2.normal(size=sp0.Ntot)
            sp0.ux[:] = 0.01 * rng2.normal(size=sp0.Ntot)
            sp0.uy[:] = 0.01 * rng2.normal(size=sp0.Ntot)
            sp0.inv_gamma[:] = 1. / np.sqrt(1. + sp0.ux**2 + sp0.uy**2 + sp0.uz**2)
        else:
            k0 = 2 * np.pi / zmax
            wp = np.sqrt(2.e24 * e**2 / (m_e * epsilon_0))
            T.impart_momenta(sp0, [0.01, 0.01, 0.01], k0, 0.4 * rmax, wp)
        res = dict(Nz=Nz, Nr=Nr, Nm=Nm, zmax=zmax, rmax=rmax, dt=dt, n_order=-1, shape=shape,
                   v_comoving=(beta_b * c if V else np.nan), use_galilean=bool(V), n_species=1,
                   q=np.array([s.q for s in sim.ptcl]), m=np.array([s.m for s in sim.ptcl]))
        snapshot(sim, 's0', res, spect=False)
        res['s0_zmin'] = sim.fld.interp[0].zmin
        done = 0
        for upto in (1, 2, 5):
            sim.step(upto - done, show_progress=False)
            done = upto
            snapshot(sim, 's%d' % upto, res)
            res['s%d_zmin' % upto] = sim.fld.interp[0].zmin
        save(name, **res)


LASER_CASES = [
    ('lg', (0, 1), dict()),
    ('lg', (1, 2), dict(theta0=0.3, theta_pol=0.7)),
    ('lg', (2, 0), dict(cep_phase=0.4)),
    ('lg', (0, 3), dict(propagation_direction=-1)),
    ('donut', (0, -1), dict()),
    ('donut', (1, 2), dict(theta_pol=0.7)),
    ('donut', (2, 0), dict(cep_phase=0.4)),
    ('donut', (0, -3), dict(propagation_direction=-1)),
    ('gauss', (), dict(theta_pol=0.2, cep_phase=0.1)),
]


def cap_laser_profiles():
    """E_field of the reference's Gaussian / Laguerre-Gauss / donut-like profiles (and of a sum
    of two) at random points (fbpic/lpa_utils/laser/laser_profiles.py:179-585)."""
    from fbpic.lpa_utils.laser import GaussianLaser, LaguerreGaussLaser, DonutLikeLaguerreGaussLaser
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=(2, 2000)) * 5e-6
    z = rng.uniform(-20e-6, 40e-6, 2000)
    t = 3e-15
    res = dict(x=x, y=y, z=z, t=t)
    cls = dict(lg=LaguerreGaussLaser, donut=DonutLikeLaguerreGaussLaser, gauss=GaussianLaser)
    for i, (kind, pm, kw) in enumerate(LASER_CASES):
        prof = cls[kind](*pm, 1.3, 4e-6, 8e-15, 1e-6, zf=12e-6, **kw)
        res['case%d' % i] = np.array(prof.E_field(x, y, z, t))
    s = LaguerreGaussLaser(0, 1, 0.5, 4e-6, 8e-15, 0., zf=5e-6, theta_pol=0., theta0=0.) \
        + LaguerreGaussLaser(0, 1, 0.5, 4e-6, 8e-15, 0., zf=5e-6, theta_pol=np.pi / 2,
                             theta0=np.pi / 2)
    res['sum'] = np.array(s.E_field(x, y, z, t))
    save('laser_profiles', **res)


def cap_uniform_rho():
    """Counterpart of tests/test_uniform_rho_deposition.py: only the assertion values."""
    # The assertions are analytic (rho = -n e inside the plasma); no fixture needed.


def c3_rows(Nz_local, iz_slab):
    """z rows of the local grid stored by the C3 fixture: 20 around the left edge of the plasma, 28
    spread over the grid (guard and damping cells at both ends included)."""
    near = np.arange(iz_slab - 10, iz_slab + 10)
    far = np.linspace(0, Nz_local - 1, 28).astype(int)
    return np.unique(np.clip(np.concatenate([near, far]), 0, Nz_local - 1))


def c3_particle_sample(P, every=300):
    """Order-independent reduction of the (8, N) particle arrays of the C3 fixture: every 300th particle
    of the lexicographic order (w, x, y, z) - unperturbed lattice particles tie exactly in (w, x, y) and
    differ in z by half a cell, perturbed ones by physical amounts - plus sum and sum of squares of every
    attribute over ALL particles."""
    o = np.lexsort((P[2], P[1], P[0], P[7]))
    return P[:, o[::every]], P.sum(axis=1), (P**2).sum(axis=1)


def c3_reduce(res, full, P, g0_zmin, dz, z_slab, Nz_local):
    iz_slab = int(round((z_slab - g0_zmin) / dz))
    rows = c3_rows(Nz_local, iz_slab)
    res['sf_rows'] = rows
    res['sf_interp_rows'] = full[:, :, rows, :]
    res['sf_interp_sum'] = full.sum(axis=(2, 3))
    res['sf_interp_sum2'] = (np.abs(full)**2).sum(axis=(2, 3))
    res['sf_interp_max'] = np.abs(full).max(axis=(2, 3))
    res['sf_zmin'] = g0_zmin
    res['sf_ntot'] = P.shape[1]
    res['sf_ptcl_sample'], res['sf_ptcl_sum'], res['sf_ptcl_sum2'] = c3_particle_sample(P)


def cap_c3_full_grid():
    """BASELINE configs[2] on its OWN grid (4096 x 256, Nm = 2, open z, moving window at c, a0 = 4
    Gaussian pulse: docs/source/example_input/lwfa_script.py).  The plasma is loaded as a slab of two
    cells inside the pulse (7360 macroparticles, 2 x 2 x 4 per cell); the reference's continuous
    injection then fills the window to the right of it at the first particle exchange (uniform
    density: 5.95 M macroparticles), so the two steps run the laser through a window full of plasma.
    The interpreted reference needs ~20 min for this; stored: 48 z rows of every grid (sf_rows), sum /
    sum of squares / maximum of every grid over all cells, every 300th particle of the (w, x, y, z)
    order and sum / sum of squares of every particle attribute."""
    import time
    from fbpic.main import Simulation
    from fbpic.lpa_utils.laser import add_laser_pulse, GaussianLaser
    zmin, zmax, rmax = -10.e-6, 30.e-6, 20.e-6
    Nz, Nr, Nm = 4096, 256, 2
    dz = (zmax - zmin) / Nz
    dt = dz / c
    z_slab = 15.e-6
    np.random.seed(0)
    t0 = time.time()
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, p_zmin=z_slab, p_zmax=z_slab + 2 * dz,
                     p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24,
                     n_order=-1, particle_shape='linear', verbose_level=0,
                     boundaries={'z': 'open', 'r': 'reflective'})
    add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
    sim.set_moving_window(v=c)
    print('built', time.time() - t0, 'Ntot', sim.ptcl[0].Ntot, 'local Nz', sim.fld.Nz, flush=True)
    nstep = 2
    res = dict(Nz=Nz, Nr=Nr, Nm=Nm, zmin=zmin, zmax=zmax, rmax=rmax, dt=dt, z_slab=z_slab, nstep=nstep,
               Nz_local=sim.fld.Nz, n_guard=sim.comm.n_guard, n_inject=sim.comm.n_inject,
               nz_damp=sim.comm.nz_damp)
    res['s0_ptcl0'] = np.array([getattr(sim.ptcl[0], k) for k in PTCL[:8]])
    for it in range(nstep):
        t0 = time.time()
        sim.step(1, show_progress=False)
        print('step', it, time.time() - t0, 'Ntot', sim.ptcl[0].Ntot, flush=True)
    g0 = sim.fld.interp[0]
    full = np.array([[getattr(sim.fld.interp[m], k) for k in INTERP] for m in range(Nm)])
    P = np.array([getattr(sim.ptcl[0], k) for k in PTCL[:8]])
    c3_reduce(res, full, P, g0.zmin, dz, z_slab, sim.fld.Nz)
    save('c3_full_grid', **res)


ALL = dict(push=cap_push, gather=cap_gather, deposit=cap_deposit, grid_setup=cap_grid_setup,
           spectral=cap_spectral, cycle=cap_cycle, bunch=cap_bunch, lwfa=cap_lwfa,
           galilean=cap_galilean, crossdep=cap_crossdep,
           laser_profiles=cap_laser_profiles, c3_full_grid=cap_c3_full_grid)

if __name__ == '__main__':
    names = sys.argv[1:] or list(ALL)
    for nme in names:
        print('==', nme)
        ALL[nme]()
