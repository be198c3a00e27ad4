#!/usr/bin/env python3
"""Why does the 1024-row z-FFT take 19 us inside a step and 8 us back to back?  Device time of
fb_zfft_from_records_consume (8 fields gathered from the deposition records) and of fb_zfft_pm_to_rt-sized plain
transforms at 1024 x 128, Nm = 2, in four contexts:
  warm      the same launch repeated back to back
  data      after a 1 GiB fill (L2 / MALL hold other data; small-code kernel: the instruction caches keep the FFT)
  code      after the one-pass particle kernel ran on OTHER buffers ... not available stand-alone: instead
  step      after a real one-pass particle launch of a C2 simulation (data and code cold, as in a step)
  step+touch after the particle launch AND a light kernel that reads the records once (data warm in L2, code cold)
usage: zfft_cold.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager

sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
lib = _capi.lib()
big = None


def timed(fn, pre=None, reps=12):
    ts = []
    for _ in range(reps):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0], ts[-1]


with GpuMemoryManager(sim):
    sim.step(12)
    fld, comm = sim.fld, sim.comm
    Nz, Nr, Nm = fld.Nz, fld.Nr, fld.Nm
    S = fld.source_records()
    S2 = torch.zeros_like(S.as_strided((Nz * S.stride(0),), (1,))).as_strided(S.shape, S.stride())
    dst = fld.d_scratch[:, 0, :]
    dst_rs = fld.d_scratch.stride(0)
    st = _capi.stream()
    big = torch.empty(1 << 27, dtype=torch.float64, device='cuda')      # 1 GiB

    def fft(rec=S):
        _capi.check(lib.fb_zfft_from_records_consume(Nz, 4 * Nm, Nr, rec.data_ptr(), rec.stride(0), rec.shape[2],
                                                     dst.data_ptr(), dst_rs, st), 'fft')

    def fill():
        big.fill_(1.0)

    sp = [s for s in sim.ptcl if s.q != 0][0]
    wz = (fld.interp[0].zmin, fld.interp[0].zmax)
    saved = [getattr(sp, k).clone() for k in ('x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma')]

    def particles():
        # one real one-pass launch (deposits into the records S), then the state is put back
        sp.cycle_sort_period, lim = 10 ** 9, (sp.cycle_stray_limit, sp.cycle_bad_limit)
        sp.cycle_stray_limit, sp.cycle_bad_limit = 2.0, 2.0
        sp.cycle(fld, comm, 0., store_fields=False, wrap_z=wz)
        sp.cycle_stray_limit, sp.cycle_bad_limit = lim

    def touch():
        # reads every record once (a reduction over the array): the data are in L2 / MALL afterwards
        torch.view_as_real(S.as_strided((Nz * S.stride(0),), (1,))).sum()

    def inv_like():
        # the inverse transform of the step: 12 plain fields, (kz, r) -> (z, r), in place on the scratch slab
        v = fld.d_scratch[:, 0, :]
        _capi.check(lib.fb_zfft(Nz, 12 * Nr, v.data_ptr(), dst_rs, v.data_ptr(), dst_rs, +1, st), 'ifft')

    print('context            median   min   max  [us]')
    for name, fn, pre in (('fwd warm', fft, None), ('fwd after 1 GiB fill', fft, fill),
                          ('fwd after particle launch', fft, particles),
                          ('fwd after particles + touch', fft, lambda: (particles(), touch())),
                          ('fwd other records, warm code', lambda: fft(S2), lambda: fft(S)),
                          ('inv warm', inv_like, None), ('inv after 1 GiB fill', inv_like, fill),
                          ('inv after particle launch', inv_like, particles)):
        m = timed(fn, pre)
        print('%-30s %6.1f %6.1f %6.1f' % ((name,) + m), flush=True)
    for k, v in zip(('x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma'), saved):
        getattr(sp, k).copy_(v)
