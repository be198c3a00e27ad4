#!/usr/bin/env python3
"""Timing of the LWFA configuration of BASELINE.json (configs[2]): docs example input scaled
to Nz x Nr = 4096 x 256, Nm = 2, 16 ppc, open z boundary, moving window, continuous injection,
Gaussian laser a0 = 4.  Prints ms/step and particle-updates/s over a steady window, and the
per-entry-point device times.  (Parity of this code path: tests/test_gpu_lwfa.py.)
usage: python tools/lwfa_bench.py [--Nz 4096 --Nr 256 --steps 100 --warmup 20]"""
import argparse
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--Nz', type=int, default=4096)
    ap.add_argument('--Nr', type=int, default=256)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--shape', default='linear')
    ap.add_argument('--nz-damp', type=int, default=64, help='damping cells in z (160: local length 4608 = 9 * 2^9, LDS FFT)')
    ap.add_argument('--filled', action='store_true', help='plasma profile starts inside the initial box (> 8 M particles from step 0)')
    a = ap.parse_args()
    import numpy as np
    import torch
    from scipy.constants import c
    from fbpic_amd import _capi
    from fbpic_amd.main import Simulation, GpuMemoryManager
    from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
    # docs/source/example_input/lwfa_script.py box (zmin=-10 um, zmax=30 um, rmax=20 um) with
    # the longitudinal / radial resolution raised to the BASELINE grid
    zmin, zmax, rmax = -10.e-6, 30.e-6, 20.e-6
    Nz, Nr, Nm = a.Nz, a.Nr, 2
    dt = (zmax - zmin) / Nz / c
    ramp_start, ramp_length = (5.e-6, 10.e-6) if a.filled else (30.e-6, 40.e-6)

    def dens_func(z, r):
        n = np.ones_like(z)
        n = np.where(z < ramp_start + ramp_length, (z - ramp_start) / ramp_length, n)
        return np.where(z < ramp_start, 0., n)
    np.random.seed(0)
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, p_zmin=ramp_start, p_zmax=500.e-6,
                     p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24,
                     dens_func=dens_func, n_order=-1, particle_shape=a.shape,
                     boundaries={'z': 'open', 'r': 'reflective'}, n_damp={'z': a.nz_damp, 'r': 32})
    add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
    sim.set_moving_window(v=c)
    t_total = 0.
    with GpuMemoryManager(sim):
        sim.step(a.warmup)
        torch.cuda.synchronize()
        n0 = sum(s.Ntot for s in sim.ptcl)
        t0 = time.perf_counter()
        sim.step(a.steps)
        torch.cuda.synchronize()
        t_total = time.perf_counter() - t0
        n1 = sum(s.Ntot for s in sim.ptcl)
        _capi.enable_timing()
        sim.step(8)
        kern = _capi.collect_timing()
    npart = 0.5 * (n0 + n1)
    out = {'workload': 'LWFA %dx%d Nm=2 16 ppc open-z moving window' % (Nz, Nr), 'Nz_local': sim.fld.Nz,
           'particles_start': n0, 'particles_end': n1, 'ms_per_step': 1e3 * t_total / a.steps,
           'particle_updates_per_s': npart * a.steps / t_total,
           'kernels_ms_per_step': {k: round(sum(r[0] for r in v) / 8, 4) for k, v in kern.items()}}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
