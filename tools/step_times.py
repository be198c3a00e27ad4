#!/usr/bin/env python3
"""Wall time of every step (synchronised) next to the sum of the entry-point device times:
python tools/step_times.py [--steps 30 --Nz .. --Nr .. --Nm .. --shape .. --ppc ..]"""
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
    sim.step(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); sim.step(a.steps); torch.cuda.synchronize(); t1 = time.perf_counter()
    print('%d steps in one call: %.3f ms/step' % (a.steps, 1e3 * (t1 - t0) / a.steps))
    _capi.enable_timing(); t0 = time.perf_counter(); sim.step(10); torch.cuda.synchronize(); t1 = time.perf_counter()
    k = _capi.collect_timing()
    tot = sum(r[0] for v in k.values() for r in v) / 10
    print('10 timed steps: %.3f ms/step wall, %.3f ms/step in entry points' % (1e3 * (t1 - t0) / 10, tot))
    for name, recs in sorted(k.items(), key=lambda kv: -sum(r[0] for r in kv[1])):
        print('  %-34s %3d x %8.1f us = %8.1f us/step' % (name, len(recs), 1e3 * sum(r[0] for r in recs) / len(recs), 1e2 * sum(r[0] for r in recs)))
