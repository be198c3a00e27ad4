#!/usr/bin/env python3
"""Where a wave of the one-pass kernel spends its time: shader clocks between the FB_MARK points of the
chunk loop, summed over all waves, from a -DFB_CYCLE_TRACE build (tools/variant.sh cycle_trace cycle.hip
-DFB_CYCLE_TRACE).  C2, steady state, a few steps; printed per 64 particles."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
path = os.path.join(ROOT, 'fbpic_amd', 'csrc', 'variants', 'libfbpic_amd_cycle_trace.so')
os.environ['FBPIC_AMD_LIB'] = path
import numpy as np, torch, helpers
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
lib = ctypes.CDLL(path)
buf = (ctypes.c_ulonglong * 16)()
names = ['loop end -> TOP (loads issued)', 'TOP -> EVAL (weights)', 'EVAL -> FRONT (stencil sums)', 'FRONT -> VAY (front of next chunk)',
         'VAY -> JSTAGE (rotate, Vay push, positions)', 'JSTAGE -> RSTAGE (wait vmcnt, stores, stage J)', 'RSTAGE -> SCATTER (stage rho, masks)',
         'SCATTER -> REDUCE (strays)', 'REDUCE -> END (run reductions)']
with GpuMemoryManager(sim):
    sim.step(40)
    torch.cuda.synchronize()
    lib.fb_debug_cycle_trace(buf, 1)
    p0 = sim.ptcl[0].cycle_passes
    sim.step(12)
    torch.cuda.synchronize()
    lib.fb_debug_cycle_trace(buf, 0)
    passes = sim.ptcl[0].cycle_passes - p0
t = np.array(buf[:9], dtype=np.float64) / (passes * 65536.)
print('%d one-pass launches; shader clocks per 64 particles and wave (sum %.0f):' % (passes, t.sum()))
for nm, v in zip(names, t):
    print('  %-48s %8.0f  %5.1f %%' % (nm, v, 100. * v / t.sum()))
