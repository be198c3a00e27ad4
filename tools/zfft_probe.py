#!/usr/bin/env python3
"""Probe: hand-written z-FFT on the slab shapes of the C2 step (single and decomposed domain)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbpic_amd.fields.spectral_transform.fourier import fft_exec


def run(Nz, Nr, nf, NFsrc, NFdst, direction, reps=50):
    rs_s, rs_d = NFsrc * Nr + 8, NFdst * Nr + 8
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
    print('Nz=%d Nr=%d nf=%2d dir %+d : %7.1f us  %6.0f GB/s' % (Nz, Nr, nf, direction, us, 2 * 16 * Nz * nf * Nr / (us * 1e-6) / 1e9), flush=True)


for Nz in (1024, 1152, 2048):
    Nr, Nm = (128, 2) if Nz < 2048 else (512, 4)
    for nf in (3 * Nm, 6 * Nm):
        for d in (-1, +1):
            run(Nz, Nr, nf, 10 * Nm, 6 * Nm, d)
