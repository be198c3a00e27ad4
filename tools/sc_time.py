#!/usr/bin/env python3
"""device time of fb_spect_cycle_standard alone at 1024 x 128, Nm = 2 (random data)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.constants import c, epsilon_0, mu_0
from fbpic_amd import _capi as hip
t = hip.torch(); hip.require_device()
Nz, Nr, Nm = 1024, 128, 2
rng = np.random.default_rng(0)
dev = lambda a: hip.to_device(np.ascontiguousarray(a))
src = dev(rng.normal(size=(Nz, 4 * Nm, Nr)) + 1j * rng.normal(size=(Nz, 4 * Nm, Nr)))
spect = dev(rng.normal(size=(Nz, 11 * Nm, Nr)) + 0j)
out = t.zeros((Nz, 6 * Nm, Nr), dtype=t.complex128, device='cuda')
mf = [dev(rng.normal(size=(Nr, Nr))) for _ in range(3 * Nm)]; mi = [dev(rng.normal(size=(Nr, Nr))) for _ in range(3 * Nm)]
iv = [dev(rng.uniform(0.5, 2, Nr)) for _ in range(Nm)]
fz = [dev(rng.uniform(0, 1, Nz)) for _ in range(Nm)]; fr = [dev(rng.uniform(0, 1, Nr)) for _ in range(Nm)]
tables = [dev(rng.normal(size=(Nz, Nr)) * 1e-3) for _ in range(8 * Nm)]
sv = [src[:, j, :] for j in range(4 * Nm)]; ov = [out[:, j, :] for j in range(6 * Nm)]
fields = [spect[:, 11 * m + i, :] for m in range(Nm) for i in range(11)]
srcs, outs = [], []
for m in range(Nm):
    srcs += sv[3 * m:3 * m + 3] + [sv[3 * Nm + m]]; outs += ov[3 * m:3 * m + 3] + ov[3 * Nm + 3 * m:3 * Nm + 3 * m + 3]
pa = hip.ptr_array
def call():
    hip.check(hip.lib().fb_spect_cycle_standard(Nm, pa(srcs), src.stride(0), pa(iv), pa(mf), pa(mi), pa(fz), pa(fr), pa(fields),
              spect.stride(0), pa(tables), 1e-16, 1, 0, c, epsilon_0, mu_0, pa(outs), out.stride(0), Nz, Nr, hip.stream()), 'sc')
for _ in range(3): call()
t.cuda.synchronize()
e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): call()
e1.record(); t.cuda.synchronize()
print(os.environ.get('FBPIC_AMD_LIB', 'default').split('/')[-1], '%.1f us per launch' % (e0.elapsed_time(e1) / 50 * 1e3))
