#!/usr/bin/env python3
"""`for _ in range(n): sim.step(1)` against `sim.step(n)` at C2 (state carried across calls)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, helpers
from fbpic_amd.main import GpuMemoryManager
n = 20
for carry in (True, False):
    sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
    sim.carry_state_between_calls = carry
    with GpuMemoryManager(sim):
        sim.step(5); torch.cuda.synchronize()
        res = []
        for rep in range(3):
            t0 = time.perf_counter(); sim.step(n); torch.cuda.synchronize()
            a = 1e3 * (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                sim.step(1)
            torch.cuda.synchronize()
            b = 1e3 * (time.perf_counter() - t0) / n
            res.append((a, b))
    print('carry=%s: step(%d) %.4f ms/step, %d x step(1) %.4f ms/step (ratio %.3f)'
          % (carry, n, min(r[0] for r in res), n, min(r[1] for r in res),
             min(r[1] for r in res) / min(r[0] for r in res)))
