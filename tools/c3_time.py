#!/usr/bin/env python3
"""where the wall time of the C3 bench goes: step(20) / finish_outputs, per-step with a sync"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from scipy.constants import c
import bench
from fbpic_amd.main import Simulation, GpuMemoryManager
from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
zmin, zmax, rmax = -10.e-6, 30.e-6, 20.e-6
Nz, Nr, Nm = 4096, 256, 2
dt = (zmax - zmin) / Nz / c
def dens_func(z, r):
    n = np.ones_like(z); n = np.where(z < 15e-6, (z - 5e-6) / 10e-6, n); return np.where(z < 5e-6, 0., n)
np.random.seed(0)
sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, p_zmin=5e-6, p_zmax=500.e-6, p_rmin=0., p_rmax=18.e-6,
                 p_nz=2, p_nr=2, p_nt=4, n_e=4.e24, dens_func=dens_func, n_order=-1, particle_shape='linear',
                 boundaries={'z': 'open', 'r': 'reflective'})
add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
sim.set_moving_window(v=c)
with GpuMemoryManager(sim):
    sim.step(16); torch.cuda.synchronize()
    t0 = time.perf_counter(); sim.step(20); torch.cuda.synchronize(); t1 = time.perf_counter()
    bench.finish_outputs(sim); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('step(20): %.2f ms/step; finish_outputs: %.1f ms' % ((t1 - t0) * 50, (t2 - t1) * 1e3))
    for k in range(6):
        t0 = time.perf_counter(); sim.step(1); torch.cuda.synchronize(); print('  step(1): %.2f ms' % ((time.perf_counter() - t0) * 1e3))
    sim.step(2); torch.cuda.synchronize()
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    bench.finish_outputs(sim); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumtime').print_stats(18)
