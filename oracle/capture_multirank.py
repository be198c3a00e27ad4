#!/usr/bin/env python3
"""Golden trajectories of the REAL reference running DOMAIN-DECOMPOSED (build container only).

TEST INFRASTRUCTURE - not part of the product.  Run as

    cd /tmp && PYTHONPATH=/root/repo/oracle/shim_mpi:/root/repo/oracle/shim:/root/reference \
        python3 -W ignore /root/repo/oracle/capture_multirank.py [names...]

mpi4py is absent from this image, so the reference's ranks run as THREADS of this process on
top of oracle/shim_mpi/mpi4py (a queue-based stand-in for the handful of MPI calls on the
PIC-cycle path: Isend / Irecv / Wait / bcast).  Every rank executes the reference's own
`Simulation.step` (CPU path: guard exchange `boundary_communicator.py:556-707`, particle
hand-over `particle_buffer_handling.py:17-172`, moving window, damping) on its own slab.
Outputs: seeded inputs + per-rank reference outputs (all local grids incl. guard cells, all
particle arrays) as tests/golden/mr_*.npz.  Nothing on the GPU box imports this file.

The Simulation objects are built one after the other in the main thread (rank by rank, each
after its own np.random.seed, as separate MPI processes would); only `step` runs threaded.
During `step` just the last rank draws random numbers (continuous injection), and it is
built last, so the global NumPy generator is in the state its own process would have.
"""
import os
import sys
import threading
import traceback
import numpy as np
from scipy.constants import c, e, m_e

OUT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
INTERP = ['Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz', 'Jr', 'Jt', 'Jz', 'rho']
PTCL = ['x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w']


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote', path, '%.1f kB' % (os.path.getsize(path) / 1e3))


def run_ranks(sims, nsteps, **kw):
    """sim.step(nsteps) on every rank, one thread per rank."""
    from mpi4py import MPI
    errs = []

    def work(r):
        MPI.set_rank(r)
        try:
            sims[r].step(nsteps, show_progress=False, **kw)
        except Exception:   # pragma: no cover
            errs.append((r, traceback.format_exc()))
    th = [threading.Thread(target=work, args=(r,)) for r in range(len(sims))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise RuntimeError('rank %d failed:\n%s' % errs[0])


def snap(sims, tag, res, ptcl=True, nfields=10):
    for r, sim in enumerate(sims):
        Nm = sim.fld.Nm
        res['%s_r%d_interp' % (tag, r)] = np.array([[getattr(sim.fld.interp[m], k) for k in INTERP[:nfields]]
                                                    for m in range(Nm)])
        res['%s_r%d_zmin' % (tag, r)] = sim.fld.interp[0].zmin
        for isp, s in enumerate(sim.ptcl):
            if ptcl:
                res['%s_r%d_ptcl%d' % (tag, r, isp)] = np.array([getattr(s, k) for k in PTCL])
            else:
                res['%s_r%d_n%d' % (tag, r, isp)] = s.Ntot


def global_plasma(Nz, Nr, dz, ppc, seed, u_th):
    """Global uniform electron plasma (lattice of the reference + seeded thermal momenta),
    generated once; every rank takes the particles of its physical z range."""
    from fbpic.main import Simulation
    from mpi4py import MPI
    MPI.set_world(1)
    MPI.set_rank(0)
    np.random.seed(seed)
    sim = Simulation(Nz, Nz * dz, Nr, Nr * dz, 2, dz / c, 0., Nz * dz, 0., Nr * dz, ppc[0], ppc[1],
                     ppc[2], 2.e24, n_order=8, verbose_level=0, use_cuda=False)
    s = sim.ptcl[0]
    rng = np.random.default_rng(seed + 1)
    P = np.array([getattr(s, k) for k in PTCL])
    P[3] = rng.normal(0., u_th, s.Ntot)
    P[4] = rng.normal(0., u_th, s.Ntot)
    P[5] = rng.normal(0., u_th, s.Ntot)
    P[6] = 1. / np.sqrt(1 + P[3]**2 + P[4]**2 + P[5]**2)
    return P


def cap_periodic(name, nranks, shape, Nz, Nr, n_order, n_guard, ppc, correct, nsteps):
    """z-periodic uniform thermal plasma on `nranks` slabs, curl-free correction on/off."""
    from fbpic.main import Simulation
    from mpi4py import MPI
    dz = 0.2e-6
    P = global_plasma(Nz, Nr, dz, ppc, seed=5, u_th=0.2)
    MPI.set_world(nranks)
    sims = []
    for r in range(nranks):
        MPI.set_rank(r)
        sim = Simulation(Nz, Nz * dz, Nr, Nr * dz, 2, dz / c, n_order=n_order, n_guard=n_guard,
                         particle_shape=shape, verbose_level=0, use_cuda=False)
        zlo, zhi = sim.comm.get_zmin_zmax(local=True, with_damp=False, with_guard=False, rank=r)
        sel = (P[2] >= zlo) & (P[2] < zhi)
        sp = sim.add_new_species(q=-e, m=m_e)
        for j, k in enumerate(PTCL):
            setattr(sp, k, P[j, sel].copy())
        sp.Ntot = int(sel.sum())
        for k in ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz'):
            setattr(sp, k, np.zeros(sp.Ntot))
        sp.cell_idx = np.empty(sp.Ntot, dtype=np.int32)
        sp.sorted_idx = np.empty(sp.Ntot, dtype=np.intp)
        sp.sorting_buffer = np.empty(sp.Ntot, dtype=np.float64)
        sims.append(sim)
    res = dict(Nz=Nz, Nr=Nr, Nm=2, dz=dz, n_order=n_order, n_guard=sims[0].comm.n_guard,
               exchange_period=sims[0].comm.exchange_period, nranks=nranks, shape=shape,
               correct=correct, P=P, nsteps=np.array(nsteps))
    done = 0
    for upto in nsteps:
        run_ranks(sims, upto - done, correct_currents=correct)
        done = upto
        snap(sims, 's%d' % upto, res, ptcl=(upto == nsteps[-1]))
    save(name, **res)


LWFA_2R = dict(Nz=128, Nr=16, zmin=-16.e-6, zmax=16.e-6, rmax=12.e-6, p_zmin=-4.e-6, p_rmax=10.e-6,
               n_order=16, n_guard=16, nz_damp=16, z0=2.e-6, zf=6.e-6)
# 8 slabs of 64 cells (the same cell size), plasma over the seven left-most slab boundaries: every
# pair of neighbours hands particles over (exchange_period 3) while the window moves, the last
# rank injects; n_order 8, curl-free current correction on every rank before the J exchange
LWFA_8R = dict(Nz=512, Nr=16, zmin=-112.e-6, zmax=16.e-6, rmax=12.e-6, p_zmin=-100.e-6, p_rmax=10.e-6,
               n_order=8, n_guard=8, nz_damp=16, z0=2.e-6, zf=6.e-6)


def lwfa_sim(shape, G=LWFA_2R):
    from fbpic.main import Simulation
    Nz, Nr, Nm = G['Nz'], G['Nr'], 2
    zmax, zmin, rmax = G['zmax'], G['zmin'], G['rmax']
    dt = (zmax - zmin) / Nz / c
    np.random.seed(11)
    return Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin,
                      p_zmin=G['p_zmin'], p_zmax=1., p_rmin=0., p_rmax=G['p_rmax'], p_nz=1, p_nr=2, p_nt=4,
                      n_e=4.e24, n_order=G['n_order'], particle_shape=shape, verbose_level=0,
                      boundaries={'z': 'open', 'r': 'reflective'}, n_guard=G['n_guard'],
                      n_damp={'z': G['nz_damp'], 'r': 8}, exchange_period=3, use_cuda=False)


def cap_lwfa(name, nranks, shape, nsteps, G=LWFA_2R):
    """Laser-wakefield miniature (open z, damping, moving window, continuous injection,
    Gaussian laser; docs/source/example_input/lwfa_script.py) on `nranks` slabs: the C4
    code path.  The plasma starts left of the slab boundary, so plasma particles are handed
    from slab to slab while the window moves.  The ranks build their Simulation one after
    the other (each after its own np.random.seed, as separate processes would; the last rank
    last), then add the laser together (add_laser_pulse gathers / scatters the global grid)."""
    from fbpic.lpa_utils.laser import add_laser_pulse, GaussianLaser
    from mpi4py import MPI
    MPI.set_world(nranks)
    sims = [None] * nranks
    errs = []
    turn = [threading.Semaphore(0) for _ in range(nranks + 1)]
    turn[0].release()

    def build(r):
        MPI.set_rank(r)
        try:
            turn[r].acquire()
            sims[r] = lwfa_sim(shape, G)
            turn[r + 1].release()
            MPI.COMM_WORLD.barrier()
            prof = GaussianLaser(a0=1.5, waist=4.e-6, tau=8.e-15, z0=G['z0'], zf=G['zf'],
                                 lambda0=0.8e-6, theta_pol=0.3, cep_phase=0.4)
            add_laser_pulse(sims[r], prof)
            sims[r].set_moving_window(v=c)
        except Exception:   # pragma: no cover
            errs.append((r, traceback.format_exc()))
            turn[r + 1].release()
    th = [threading.Thread(target=build, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise RuntimeError('rank %d failed in setup:\n%s' % errs[0])
    s0 = sims[0]
    res = dict(Nz=G['Nz'], Nr=G['Nr'], Nm=2, zmin=G['zmin'], zmax=G['zmax'], rmax=G['rmax'], dt=s0.dt, shape=shape,
               p_zmin=G['p_zmin'], p_rmax=G['p_rmax'], n_order=G['n_order'], z0=G['z0'], zf=G['zf'],
               nranks=nranks, n_guard=s0.comm.n_guard, n_inject=s0.comm.n_inject,
               nz_damp=s0.comm.nz_damp, nsteps=np.array(nsteps),
               Nz_local=np.array([s.fld.Nz for s in sims]))
    snap(sims, 's0', res, nfields=6)        # laser fields on the decomposed grid
    done = 0
    for upto in nsteps:
        run_ranks(sims, upto - done)
        done = upto
        snap(sims, 's%d' % upto, res, ptcl=(upto == nsteps[-1]))
    save(name, **res)


# ---- BASELINE config C4 on its own grid: 4096 x 256, Nm = 2, n_order = 32, n_guard = 64, 8 slabs
C4 = dict(Nz=4096, Nr=256, zmin=-10.e-6, zmax=30.e-6, rmax=20.e-6, n_order=32, n_guard=64, nranks=8,
          nslab=2)          # plasma slabs of `nslab` cells, one cell to the right of every slab boundary


def c4_dens_func(G):
    """Plasma only in thin slabs just right of the 7 inner slab boundaries (the window moves one cell per
    step: every slab crosses its boundary within four steps and is handed to the left neighbour at the first
    particle exchange; the slab of boundary 5 sits at the centre of the laser pulse).  Everywhere else the
    density is 0, so the continuous injection adds nothing and the interpreted reference stays affordable."""
    dz = (G['zmax'] - G['zmin']) / G['Nz']
    per = G['Nz'] // G['nranks']
    edges = [G['zmin'] + r * per * dz for r in range(1, G['nranks'])]

    def dens(z, r):
        n = np.zeros_like(z)
        for b in edges:
            n = np.where((z >= b + dz) & (z < b + (1 + G['nslab']) * dz), 1., n)
        return n
    return dens


def c4_rows(Nz_local, ng):
    """z rows of a rank's local grid kept by the C4 fixture: two in each guard region, two inside."""
    return np.array([ng // 2, ng + 1, ng + 3, Nz_local // 2, Nz_local - ng - 2, Nz_local - ng // 2 - 1])


def c4_sim(G):
    from fbpic.main import Simulation
    dt = (G['zmax'] - G['zmin']) / G['Nz'] / c
    np.random.seed(0)
    return Simulation(G['Nz'], G['zmax'], G['Nr'], G['rmax'], 2, dt, zmin=G['zmin'],
                      p_zmin=G['zmin'], p_zmax=G['zmax'], p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4,
                      n_e=4.e24, dens_func=c4_dens_func(G), n_order=G['n_order'], n_guard=G['n_guard'],
                      particle_shape='linear', verbose_level=0, boundaries={'z': 'open', 'r': 'reflective'},
                      use_cuda=False)


def cap_c4_full_grid(name='c4_full_grid', G=C4):
    """The laser-wakefield window of BASELINE config 4 at its own size on 8 slabs, run by the REAL reference
    (every rank its own Simulation.step: guard exchanges of E, B and of the corrected J, particle hand-over at
    the first exchange, moving window, damping at the two open ends, a0 = 4 pulse), exchange_period + 2
    steps.  Stored per rank: 6 z rows of every grid, sum / sum of squares / maximum of every grid over
    the whole local grid, every third particle of the (w, x, y, z) order and the moments of every particle
    attribute."""
    import time
    from fbpic.lpa_utils.laser import add_laser_pulse, GaussianLaser
    from mpi4py import MPI
    nranks = G['nranks']
    MPI.set_world(nranks)
    sims = [None] * nranks
    errs = []
    turn = [threading.Semaphore(0) for _ in range(nranks + 1)]
    turn[0].release()

    def build(r):
        MPI.set_rank(r)
        try:
            turn[r].acquire()
            sims[r] = c4_sim(G)
            turn[r + 1].release()
            MPI.COMM_WORLD.barrier()
            add_laser_pulse(sims[r], GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
            sims[r].set_moving_window(v=c)
        except Exception:   # pragma: no cover
            errs.append((r, traceback.format_exc()))
            turn[r + 1].release()
    t0 = time.time()
    th = [threading.Thread(target=build, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise RuntimeError('rank %d failed in setup:\n%s' % errs[0])
    s0 = sims[0]
    nstep = int(s0.comm.exchange_period) + 2
    print('built', time.time() - t0, 'exchange_period', s0.comm.exchange_period, 'local Nz',
          [s.fld.Nz for s in sims], 'Ntot', [s.ptcl[0].Ntot for s in sims], flush=True)
    res = dict(Nz=G['Nz'], Nr=G['Nr'], Nm=2, zmin=G['zmin'], zmax=G['zmax'], rmax=G['rmax'], dt=s0.dt,
               n_order=G['n_order'], n_guard=s0.comm.n_guard, nz_damp=s0.comm.nz_damp, n_inject=s0.comm.n_inject,
               nranks=nranks, nslab=G['nslab'], nstep=nstep, exchange_period=s0.comm.exchange_period,
               Nz_local=np.array([s.fld.Nz for s in sims]),
               n0=np.array([s.ptcl[0].Ntot for s in sims]))
    for it in range(nstep):
        t0 = time.time()
        run_ranks(sims, 1)
        print('step', it, time.time() - t0, 'Ntot', [s.ptcl[0].Ntot for s in sims], flush=True)
    for r, sim in enumerate(sims):
        full = np.array([[getattr(sim.fld.interp[m], k) for k in INTERP] for m in range(2)])
        rows = c4_rows(sim.fld.Nz, sim.comm.n_guard)
        res['r%d_rows' % r] = rows
        res['r%d_interp_rows' % r] = full[:, :, rows, :]
        res['r%d_interp_sum' % r] = full.sum(axis=(2, 3))
        res['r%d_interp_sum2' % r] = (np.abs(full)**2).sum(axis=(2, 3))
        res['r%d_interp_max' % r] = np.abs(full).max(axis=(2, 3))
        res['r%d_zmin' % r] = sim.fld.interp[0].zmin
        P = np.array([getattr(sim.ptcl[0], k) for k in PTCL])
        res['r%d_ntot' % r] = P.shape[1]
        o = np.lexsort((P[2], P[1], P[0], P[7]))
        res['r%d_ptcl_sample' % r] = P[:, o[::3]]
        res['r%d_ptcl_sum' % r] = P.sum(axis=1)
        res['r%d_ptcl_sum2' % r] = (P**2).sum(axis=1)
    save(name, **res)


CASES = {
    'c4_full_grid': cap_c4_full_grid,
    # 48 physical + 2 x 12 guard cells per rank; exchange_period = int((12/2 - 3) / 2) = 1
    'mr_periodic_lin_2r': lambda: cap_periodic('mr_periodic_lin_2r', 2, 'linear', 96, 8, 8, 12,
                                               (1, 2, 4), True, (1, 5)),
    'mr_periodic_cub_2r': lambda: cap_periodic('mr_periodic_cub_2r', 2, 'cubic', 96, 8, 8, 12,
                                               (1, 2, 4), True, (1, 5)),
    'mr_periodic_lin_2r_nocorr': lambda: cap_periodic('mr_periodic_lin_2r_nocorr', 2, 'linear', 96,
                                                      8, 8, 12, (1, 2, 4), False, (1, 5)),
    # 32 physical + 2 x 10 guard cells per rank (n_order 4): every rank has two distinct neighbours
    'mr_periodic_lin_4r': lambda: cap_periodic('mr_periodic_lin_4r', 4, 'linear', 128, 8, 4, 10,
                                               (1, 2, 4), True, (1, 4)),
    'mr_lwfa_lin_2r': lambda: cap_lwfa('mr_lwfa_lin_2r', 2, 'linear', (10,)),
    'mr_lwfa_lin_8r': lambda: cap_lwfa('mr_lwfa_lin_8r', 8, 'linear', (7,), LWFA_8R),
}

if __name__ == '__main__':
    names = sys.argv[1:] or list(CASES)
    for n in names:
        CASES[n]()
