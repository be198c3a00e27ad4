#!/usr/bin/env python3
"""C2 through the one-pass sequence for a few dozen steps (profiling target of tools/sq_probe.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, helpers
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
with GpuMemoryManager(sim):
    sim.step(24)
    torch.cuda.synchronize()
    sim.step(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
    torch.cuda.synchronize()
