#!/usr/bin/env python3
"""Host issue time vs device time of Simulation.step: python tools/hosttime.py [--decomposed]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
ap = argparse.ArgumentParser(); ap.add_argument('--decomposed', action='store_true'); ap.add_argument('--steps', type=int, default=28)
a = ap.parse_args()
import torch, helpers
from fbpic_amd.main import GpuMemoryManager
if a.decomposed:
    import loopback_multirank as lb
    from fbpic_amd.boundaries import boundary_communicator as bc
    lb.install_loopback(bc, torch)
world = 2 if a.decomposed else 1
sim = helpers.uniform_plasma_sim(1024 * world, 128, 2, (2, 4, 4), 'linear', seed=0, n_order=(32 if a.decomposed else -1), n_guard=(64 if a.decomposed else None))
with GpuMemoryManager(sim):
    sim.step(6); torch.cuda.synchronize()
    # iterations 6.. : start right after an exchange (iteration 14 would be the next one)
    sim.step(9); torch.cuda.synchronize()      # now at iteration 15
    t0 = time.perf_counter(); sim.step(12); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('%s: host issue %.1f us/step, total %.1f us/step' % ('decomposed' if a.decomposed else 'single', 1e6 * (t1 - t0) / 12, 1e6 * (t2 - t0) / 12))
import cProfile, pstats
with GpuMemoryManager(sim):
    pr = cProfile.Profile(); pr.enable(); sim.step(12); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
