#!/usr/bin/env python3
"""Probe: fb_hankel device time vs job count and size (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbpic_amd import _capi

PAD = 8 if '--nopad' not in sys.argv else 0


def run(Nz, Nr, njobs, reps=40):
    lib = _capi.lib()
    # padded z-major slabs as in Fields._alloc_slab (row stride = nfields*Nr + 8)
    rs = njobs * Nr + PAD
    a = torch.randn(Nz * rs, dtype=torch.complex128, device='cuda').as_strided((Nz, njobs, Nr), (rs, Nr, 1))
    b = torch.zeros(Nz * rs, dtype=torch.complex128, device='cuda').as_strided((Nz, njobs, Nr), (rs, Nr, 1))
    mats = [torch.randn((Nr, Nr), dtype=torch.float64, device='cuda') for _ in range(njobs)]
    ins = _capi.ptr_array([a[:, j, :] for j in range(njobs)])
    outs = _capi.ptr_array([b[:, j, :] for j in range(njobs)])
    mp = _capi.ptr_array(mats)
    def call():
        _capi.check(lib.fb_hankel(njobs, ins, rs, outs, rs, mp, 1.0, Nz, Nr, _capi.stream()), 'hk')
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 4.0 * Nz * Nr * Nr * njobs / (us * 1e-6) / 1e12
    print('Nz=%d Nr=%d jobs=%2d : %7.1f us  %5.1f TFLOP/s (%.0f%% of 78.6)' % (Nz, Nr, njobs, us, tf, 100 * tf / 78.6), flush=True)

if '--only' in sys.argv:          # --only Nz,Nr,njobs
    Nz, Nr, nj = (int(v) for v in sys.argv[sys.argv.index('--only') + 1].split(','))
    run(Nz, Nr, nj, reps=10)
else:
    for (Nz, Nr) in ((1024, 128), (2048, 512), (4096, 256), (256, 64)):
        for nj in (2, 6, 12, 24):
            run(Nz, Nr, nj)
