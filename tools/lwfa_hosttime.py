#!/usr/bin/env python3
"""Host issue time vs wall time of the C3 (laser-wakefield, moving window) step + host profile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch
from scipy.constants import c
from fbpic_amd.main import Simulation, GpuMemoryManager
from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
zmin, zmax, rmax = -10.e-6, 30.e-6, 20.e-6
Nz, Nr, Nm = 4096, 256, 2
dt = (zmax - zmin) / Nz / c
ramp_start, ramp_length = 5.e-6, 10.e-6


def dens_func(z, r):
    n = np.ones_like(z)
    n = np.where(z < ramp_start + ramp_length, (z - ramp_start) / ramp_length, n)
    return np.where(z < ramp_start, 0., n)


np.random.seed(0)
sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, p_zmin=ramp_start, p_zmax=500.e-6, p_rmin=0.,
                 p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24, dens_func=dens_func, n_order=-1,
                 particle_shape='linear', boundaries={'z': 'open', 'r': 'reflective'}, n_damp={'z': 64, 'r': 32})
add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
sim.set_moving_window(v=c)
with GpuMemoryManager(sim):
    sim.step(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); sim.step(40); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('host issue %.3f ms/step, wall %.3f ms/step, exchange_period %d' % (1e3 * (t1 - t0) / 40, 1e3 * (t2 - t0) / 40, sim.comm.exchange_period))
    ts = []
    for i in range(34):
        t0 = time.perf_counter(); sim.step(1); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print('per-call ms:', ' '.join('%.2f' % v for v in ts))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); sim.step(40); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
