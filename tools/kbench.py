#!/usr/bin/env python3
"""Per-kernel micro-benchmark on a realistic (cell-sorted, thermal) C2 state.
usage: python tools/kbench.py [--reps 20] [--only gather,deposit_J,...]
Prints mean device time per launch (HIP events) and achieved GB/s."""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--Nz', type=int, default=1024)
    ap.add_argument('--Nr', type=int, default=128)
    ap.add_argument('--Nm', type=int, default=2)
    ap.add_argument('--shape', default='linear')
    ap.add_argument('--ppc', default='2,4,4')
    a = ap.parse_args()
    import torch
    import helpers
    from fbpic_amd.main import GpuMemoryManager
    ppc = tuple(int(v) for v in a.ppc.split(','))
    sim = helpers.uniform_plasma_sim(a.Nz, a.Nr, a.Nm, ppc, a.shape, seed=0)
    s = sim.ptcl[0]
    fld = sim.fld
    n = s.Ntot
    only = set(k for k in a.only.split(',') if k)

    def timeit(name, fn, bytes_per_particle=None):
        if only and name not in only:
            return
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        extra = ''
        if bytes_per_particle:
            extra = '  %.0f GB/s (%.1f%% of 8 TB/s)' % (bytes_per_particle * n / ms / 1e6,
                                                      bytes_per_particle * n / ms / 1e6 / 80.)
        print('%-22s %8.1f us%s' % (name, ms * 1e3, extra))

    with GpuMemoryManager(sim):
        sim.step(2)
        s.sort_particles(fld)
        s.sorted = True
        timeit('gather', lambda: s.gather(fld.interp, sim.comm), 72)
        timeit('gather_push', lambda: s.gather_push(fld.interp, sim.comm, 0.), 136)
        timeit('gather_push_ns', lambda: s.gather_push(fld.interp, sim.comm, 0., store_fields=False), 112)
        timeit('push_p', lambda: s.push_p(0.), 112)
        timeit('deposit_J', lambda: s.deposit(fld, 'J'), 64)
        timeit('deposit_rho', lambda: s.deposit(fld, 'rho'), 32)
        timeit('deposit_J_rec', lambda: s.deposit(fld, 'J', records=True), 64)
        timeit('deposit_rho_rec', lambda: s.deposit(fld, 'rho', records=True), 32)
        timeit('sort', lambda: s.sort_particles(fld))
        timeit('interp2spect_J', lambda: fld.interp2spect('J'))
        timeit('spect2interp_E', lambda: fld.spect2interp('E'))
        timeit('interp2spect_rho', lambda: fld.interp2spect('rho_next'))
        timeit('push_fields', lambda: fld.push())
        timeit('correct_currents', lambda: fld.correct_currents())
        timeit('step', lambda: sim.step(1))


if __name__ == '__main__':
    main()
