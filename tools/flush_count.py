#!/usr/bin/env python3
"""Runs (= flushes) counted by the J deposition kernel as the lattice thermalises:
python tools/flush_count.py [--Nz 512 --Nr 128 --Nm 4 --shape cubic --ppc 2,2,16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('--Nz', type=int, default=512); ap.add_argument('--Nr', type=int, default=128)
ap.add_argument('--Nm', type=int, default=4); ap.add_argument('--shape', default='cubic')
ap.add_argument('--ppc', default='2,2,16')
a = ap.parse_args()
import torch, helpers
from scipy.constants import c
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
from fbpic_amd.particles import particles as P
sim = helpers.uniform_plasma_sim(a.Nz, a.Nr, a.Nm, tuple(int(v) for v in a.ppc.split(',')), a.shape, seed=0)
s = sim.ptcl[0]
real_lib = _capi.lib()
real = real_lib.fb_deposit_J
log = []


def counted(*args):
    args = list(args)
    s._nflush.zero_()
    args[-2] = _capi.ptr(s._nflush)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = real(*args); e1.record(); torch.cuda.synchronize()
    log.append((int(s._nflush.sum().item()), e0.elapsed_time(e1)))
    return rc


class Lib:
    def __getattr__(self, k):
        return counted if k == 'fb_deposit_J' else getattr(real_lib, k)


orig = _capi.lib
with GpuMemoryManager(sim):
    P._capi.lib = lambda: Lib()
    try:
        sim.step(30)
    finally:
        P._capi.lib = orig
n = s.Ntot
print('particles', n, 'chunks', n // 64)
for i, (r, ms) in enumerate(log):
    print('J deposit %2d: runs %8d = %.2f per 64 particles, %.3f ms' % (i, r, r / (n / 64.), ms))
