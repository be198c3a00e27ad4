#!/usr/bin/env python3
"""Per-launch device times of one entry point over consecutive steps:
python tools/launch_times.py fb_deposit_J [--Nz ... stepbench args]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('entry')
ap.add_argument('--steps', type=int, default=24)
ap.add_argument('--Nz', type=int, default=1024); ap.add_argument('--Nr', type=int, default=128)
ap.add_argument('--Nm', type=int, default=2); ap.add_argument('--shape', default='linear')
ap.add_argument('--ppc', default='2,4,4')
a = ap.parse_args()
import torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(a.Nz, a.Nr, a.Nm, tuple(int(v) for v in a.ppc.split(',')), a.shape, seed=0)
with GpuMemoryManager(sim):
    _capi.enable_timing(); sim.step(a.steps); k = _capi.collect_timing()
for name in a.entry.split(','):
    print(name, ' '.join('%.0f' % (1e3 * r[0]) for r in k.get(name, [])))
