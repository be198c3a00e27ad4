#!/usr/bin/env python3
"""Device time of the particle launches step by step from the start of a C2 run (why the first
~35 steps are slower than the steady state)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
with GpuMemoryManager(sim):
    sim.step(2)
    rows = []
    for it in range(64):
        _capi.enable_timing()
        sim.step(1)
        k = _capi.collect_timing()
        one = sum(r[0] for r in k.get('fb_gather_push_deposit_J_rho', []))
        two = sum(r[0] for n in ('fb_gather_push_rank_next_home', 'fb_gather_push_rank_next', 'fb_push_x_sort_deposit_J_rho') for r in k.get(n, []))
        grid = sum(r[0] for n, v in k.items() if n not in ('fb_gather_push_deposit_J_rho', 'fb_gather_push_rank_next_home', 'fb_gather_push_rank_next', 'fb_push_x_sort_deposit_J_rho') for r in v)
        rows.append((it + 3, one, two, grid, sim.ptcl[0].cycle_last_stray_fraction))
for r in rows:
    print('%3d  one-pass %.3f  two-pass %.3f  rest %.3f  strays %s' % r)
