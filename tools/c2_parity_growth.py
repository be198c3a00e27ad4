#!/usr/bin/env python3
"""How the deviation between the HIP path and the CPU oracle grows with the step count at the
benchmarked size (C2: 1024 x 128, Nm = 2, 32 ppc), field group by field group - and the same for two
ORACLE runs that differ only in their thread count (thread-private deposition grids: another summation
order of the same arithmetic): what a rounding-level difference of the deposition becomes after n
steps of a thermal plasma.  usage: c2_parity_growth.py [nsteps] [Nz Nr]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch, helpers
from oracle import oracle as orc
from fbpic_amd.main import GpuMemoryManager
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
Nz, Nr = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 128)
sim = helpers.uniform_plasma_sim(Nz, Nr, 2, (2, 4, 4), 'linear', seed=0)
oa = helpers.oracle_from_sim(orc, sim, nthreads=16)
ob = helpers.oracle_from_sim(orc, sim, nthreads=3)
INTERP = helpers.INTERP
def dev(get, ref):
    out = {}
    for g in 'EBJr':
        keys = [k for k in INTERP if k[0] == g]
        scale = max(np.abs(ref.interp[m][k]).max() for m in range(2) for k in keys)
        out[g] = max(np.abs(get(m, k) - ref.interp[m][k]).max() for m in range(2) for k in keys) / max(scale, 1e-300)
    return out
with GpuMemoryManager(sim):
    for it in range(1, nsteps + 1):
        sim.step(1)
        oa.step(1)
        ob.step(1)
        hip = dev(lambda m, k: getattr(sim.fld.interp[m], k).cpu().numpy(), oa)
        thr = dev(lambda m, k: ob.interp[m][k], oa)
        s = sim.ptcl[0]
        print('step %d  HIP vs oracle(16 thr): ' % it + '  '.join('%s %.2e' % (g, hip[g]) for g in 'EBJr')
              + '   | oracle(3 thr) vs oracle(16 thr): ' + '  '.join('%s %.2e' % (g, thr[g]) for g in 'EBJr'), flush=True)
