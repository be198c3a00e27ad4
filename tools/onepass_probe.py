#!/usr/bin/env python3
"""Step time of the C2 workload through the one-pass particle cycle for a list of re-sort
periods, next to the two-pass sequence; per-entry-point device times of each.
usage: python tools/onepass_probe.py [--periods 1,4,8,16] [--steps 48] [--config C2|C5lin]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('--periods', default='0,1,4,8,16,32')
ap.add_argument('--steps', type=int, default=48)
ap.add_argument('--Nz', type=int, default=1024); ap.add_argument('--Nr', type=int, default=128)
ap.add_argument('--Nm', type=int, default=2); ap.add_argument('--ppc', default='2,4,4')
ap.add_argument('--uth', type=float, default=0.01)
ap.add_argument('--limit', type=float, default=1.0, help='stray limit (1.0 = period only)')
a = ap.parse_args()
import torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
print('WPE', os.environ.get('FBPIC_AMD_CYCLE_WPE', '0'), 'LIB', os.environ.get('FBPIC_AMD_LIB', 'default'))
for per in [int(v) for v in a.periods.split(',')]:
    sim = helpers.uniform_plasma_sim(a.Nz, a.Nr, a.Nm, tuple(int(v) for v in a.ppc.split(',')), 'linear',
                                     seed=0, u_th=a.uth)
    sim.one_pass_cycle = per > 0
    s = sim.ptcl[0]
    s.cycle_sort_period = max(per, 1)
    s.cycle_stray_limit = a.limit
    with GpuMemoryManager(sim):
        sim.step(24); torch.cuda.synchronize()          # lattice thermalised
        t0 = time.perf_counter(); sim.step(a.steps); torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        _capi.enable_timing(); sim.step(16); k = _capi.collect_timing()
    tot = sum(sum(r[0] for r in recs) for recs in k.values()) / 16
    print('period %2d: %.4f ms/step  %.3e updates/s   (device %.4f ms; sorts %d passes %d stray %s)' % (
        per, ms, s.Ntot / ms * 1e3, tot, s.cycle_sorts, s.cycle_passes, s.cycle_stray_fraction))
    for name, recs in sorted(k.items(), key=lambda kv: -sum(r[0] for r in kv[1]))[:7]:
        print('     %-34s %3d x %8.1f us' % (name, len(recs), 1e3 * sum(r[0] for r in recs) / len(recs)))
    sys.stdout.flush()
