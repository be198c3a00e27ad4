#!/usr/bin/env python3
"""The C2 line gets faster with the number of steps already run (warm-up 5 / 25 / 60: 0.417 / 0.40 / 0.381 ms per step).
Device or host?  Device time of every launch of the first 100 steps after the bench's own start-up (entry-point events),
averaged per window of 10 steps, next to the wall time of the same windows.
usage: early_drift.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
with GpuMemoryManager(sim):
    sim.step(5); torch.cuda.synchronize()
    # (a) wall time of windows of 10 steps, one call each, no event recording
    walls = []
    for w in range(10):
        t0 = time.perf_counter(); sim.step(10); torch.cuda.synchronize(); walls.append(1e2 * (time.perf_counter() - t0))
    print('wall ms/step per window of 10 steps (steps 5 ..):', ' '.join('%.3f' % v for v in walls))
sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
with GpuMemoryManager(sim):
    sim.step(5); torch.cuda.synchronize()
    for w in range(10):
        _capi.enable_timing()
        t0 = time.perf_counter(); sim.step(10); th = time.perf_counter() - t0
        k = _capi.collect_timing()
        tw = time.perf_counter() - t0
        tot = sum(r[0] for v in k.values() for r in v) / 10
        op = k.get('fb_gather_push_deposit_J_rho', [])
        print('window %d: wall %.3f host-issue %.3f device %.3f ms/step | one-pass %d x %.1f us, spect %.1f us, zfft %.1f + %.1f us, strays %s'
              % (w, 1e2 * tw, 1e2 * th, tot, len(op), 1e3 * sum(r[0] for r in op) / max(len(op), 1),
                 1e3 * sum(r[0] for r in k.get('fb_spect_cycle_standard', [(0,)])) / max(len(k.get('fb_spect_cycle_standard', [1])), 1),
                 1e3 * sum(r[0] for r in k.get('fb_zfft_from_records_consume', [(0,)])) / max(len(k.get('fb_zfft_from_records_consume', [1])), 1),
                 1e3 * sum(r[0] for r in k.get('fb_zfft_pm_to_rt', [(0,)])) / max(len(k.get('fb_zfft_pm_to_rt', [1])), 1),
                 sim.ptcl[0].cycle_last_stray_fraction), flush=True)
