#!/usr/bin/env python3
"""Phase stamps (shader clock, wave 0 of every workgroup) of the fused spectral launch, from a -DSC_TRACE
build of spectral_cycle.hip (tools/variant.sh sc_trace spectral_cycle.hip -DSC_TRACE): runs C2 for a few
steps on the default library with the variant's kernel and prints the mean duration of every phase."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
path = os.path.join(ROOT, 'fbpic_amd', 'csrc', 'variants', 'libfbpic_amd_sc_trace.so')
os.environ['FBPIC_AMD_LIB'] = path
import numpy as np, torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
with GpuMemoryManager(sim):
    sim.step(12)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(path)
    n = 256 * 8
    buf = (ctypes.c_ulonglong * n)()
    lib.fb_debug_sc_trace(buf, n)
    t = np.array(buf[:], dtype=np.float64).reshape(256, 8)
names = ['sources -> panels', 'forward products', 'cell update', 'barrier', 'inverse Ep|Bp', 'inverse Em|Bm', 'inverse Ez|Bz']
d = np.diff(t[:, :8], axis=1)
print('phase durations in shader clocks (mean / min / max over 256 workgroups), last launch:')
for i, nm in enumerate(names):
    print('  %-20s %8.0f %8.0f %8.0f' % (nm, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
print('  %-20s %8.0f' % ('total', (t[:, 7] - t[:, 0]).mean()), ' start spread', t[:, 0].max() - t[:, 0].min(), ' end spread', t[:, 7].max() - t[:, 7].min(),
      ' kernel span', t[:, 7].max() - t[:, 0].min())
