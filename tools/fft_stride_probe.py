#!/usr/bin/env python3
"""Probe: rocFFT z-transform time vs row stride of the z-major slab (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbpic_amd.fields.spectral_transform.fourier import fft_exec

def run(Nz, Nr, nf, NFsrc, NFdst, pad_src, pad_dst, direction, reps=30):
    rs_s, rs_d = NFsrc * Nr + pad_src, NFdst * Nr + pad_dst
    a = torch.randn(Nz * rs_s, dtype=torch.complex128, device='cuda')
    b = torch.zeros(Nz * rs_d, dtype=torch.complex128, device='cuda')
    src = torch.as_strided(a, (Nz, nf * Nr), (rs_s, 1))
    dst = torch.as_strided(b, (Nz, nf * Nr), (rs_d, 1))
    for _ in range(3):
        fft_exec(src, dst, direction)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fft_exec(src, dst, direction)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gbs = 2 * 16 * Nz * nf * Nr / (us * 1e-6) / 1e9
    print('Nz=%d Nr=%d nf=%2d strides %6d->%6d dir %+d : %7.1f us  %7.0f GB/s' % (Nz, Nr, nf, rs_s, rs_d, direction, us, gbs), flush=True)

from fbpic_amd.fields.spectral_transform import fourier
pads = (0, 8) if '--pads' not in sys.argv else (0, 8, 16, 24, 40, 72, 136)
for use_zfft in (False, True):
    fourier.USE_ZFFT = use_zfft
    print('--- hand-written kernel' if use_zfft else '--- rocFFT')
    for (Nz, Nr, Nm) in ((256, 64, 2), (512, 128, 2), (1024, 128, 2), (1152, 128, 2), (2048, 512, 4), (4096, 256, 2)):
        NFi, NFs, NFx = 10 * Nm, 11 * Nm, 6 * Nm
        for pad in pads:
            run(Nz, Nr, 3 * Nm, NFi, NFx, pad, pad, -1)
            run(Nz, Nr, 6 * Nm, NFx, NFi, pad, pad, +1)
            run(Nz, Nr, Nm, NFi, NFx, pad, pad, -1)
