#!/usr/bin/env python3
"""Steady-state step time of the C2 workload (or --Nz/--Nr/--Nm/--shape/--ppc), with the
per-entry-point device times: python tools/stepbench.py [--steps 30]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--Nz', type=int, default=1024); ap.add_argument('--Nr', type=int, default=128)
ap.add_argument('--Nm', type=int, default=2); ap.add_argument('--shape', default='linear')
ap.add_argument('--ppc', default='2,4,4')
a = ap.parse_args()
import torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(a.Nz, a.Nr, a.Nm, tuple(int(v) for v in a.ppc.split(',')), a.shape, seed=0)
with GpuMemoryManager(sim):
    sim.step(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); sim.step(a.steps); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    _capi.enable_timing(); sim.step(10); k = _capi.collect_timing()
n = sim.ptcl[0].Ntot
print('%.4f ms/step  %.3e updates/s' % (ms, n / ms * 1e3))
for name, recs in sorted(k.items(), key=lambda kv: -sum(r[0] for r in kv[1])):
    print('  %-34s %3d x %8.1f us' % (name, len(recs), 1e3 * sum(r[0] for r in recs) / len(recs)))
