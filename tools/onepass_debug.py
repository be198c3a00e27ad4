#!/usr/bin/env python3
"""debug driver: one launch of fb_gather_push_deposit_J_rho on random particles with toggles
usage: onepass_debug.py n ustd store(0/1) stats(0/1) sorted(0/1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from scipy.constants import c, e, m_e
from fbpic_amd import _capi as hip
n, ustd, store, use_stats, do_sort = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
t = hip.torch(); hip.require_device(); p = hip.ptr
rng = np.random.default_rng(9)
Nz, Nr, Nm = 36, 20, 2
dzc = 0.2e-6
geom = (1. / dzc, 0., Nz, 1. / dzc, 0., Nr)
r = rng.uniform(0, float(sys.argv[6]) * Nr * dzc, n); th = rng.uniform(0, 2 * np.pi, n)
x, y, z = r * np.cos(th), r * np.sin(th), rng.uniform(0., Nz * dzc, n)
ux, uy, uz = (rng.normal(size=n) * ustd for _ in range(3))
ig = 1. / np.sqrt(1. + ux**2 + uy**2 + uz**2); w = rng.uniform(0.5, 1.5, n)
dev = lambda a: hip.to_device(np.ascontiguousarray(a))
src = [dev(a) for a in (x, y, z, ux, uy, uz, w, ig)]
ncell = Nz * (Nr + 1)
home = t.zeros(n, dtype=t.int32, device='cuda')
if do_sort:
    dst = [t.empty_like(a) for a in src]
    pre = t.empty(ncell, dtype=t.int32, device='cuda')
    nb = int(hip.lib().fb_bin_sort_workspace_bytes(n, ncell)); ws = t.empty(nb, dtype=t.uint8, device='cuda')
    hip.check(hip.lib().fb_bin_sort_particles(n, ncell, p(src[0]), p(src[1]), p(src[2]), *geom, 8, hip.ptr_array(src),
                                              hip.ptr_array(dst), p(home), None, p(pre), p(ws), nb, hip.stream()), 'sort')
    a = dst
else:
    a = src
t.cuda.synchronize(); print('sorted ok', flush=True)
views = [dev((rng.normal(size=(Nz, Nr)) + 1j * rng.normal(size=(Nz, Nr))) * 1e9) for _ in range(6 * Nm)]
ruy = dev(np.zeros(Nr + 1))
F = [t.zeros(n, dtype=t.float64, device='cuda') for _ in range(6)]
stats = t.zeros(1024, dtype=t.int64, device='cuda')
rec = t.zeros((Nz, Nr, 4 * Nm), dtype=t.complex128, device='cuda')
jv = [rec[:, :, 4 * mm + k] for mm in range(Nm) for k in range(3)]; rv = [rec[:, :, 4 * mm + 3] for mm in range(Nm)]
dt = dzc / c
hip.check(hip.lib().fb_gather_push_deposit_J_rho(
    1, Nm, n, p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(a[4]), p(a[5]), p(a[7]), p(a[6]), p(home),
    Nr * dzc, *geom, hip.ptr_array(views), Nr, *[p(f) if store else None for f in F], -e, m_e, c, dt, 0.5 * dt, 0., Nz * dzc,
    hip.ptr_array(jv), jv[0].stride(0), jv[0].stride(1), hip.ptr_array(rv), rv[0].stride(0), rv[0].stride(1),
    p(ruy), p(ruy), p(stats) if use_stats else None, hip.stream()), 'one pass')
t.cuda.synchronize()
print('OK', sys.argv[1:], 'strays', int(stats.sum()), 'rec', float(rec.abs().max()), flush=True)
