#!/usr/bin/env python3
"""ms/step of successive blocks of 5 steps of the C2 workload, as bench.py runs it (calls of
step(5) + the output hand-back): shows whether the first calls of a process are slower."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, helpers, bench
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
with GpuMemoryManager(sim):
    if len(sys.argv) > 1:
        bench.measured_ceilings(torch)
    out = []
    for b in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.step(5); bench.finish_outputs(sim); torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / 5)
print(' '.join('%.3f' % v for v in out))
