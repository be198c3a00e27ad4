#!/usr/bin/env python3
"""Launch time of the one-pass kernel on a FROZEN particle state (position step 0: the strays of
the state stay what they are), r one-pass iterations after a sort, for the default library and
every tools/variant.sh build under fbpic_amd/csrc/variants/ - the simulation itself always runs on
the default library (a knock-out build's physics runs away).
usage: cycle_knock.py [Nm] [ppc as nz,nr,nt]"""
import os, sys, glob, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, helpers
from fbpic_amd import _capi
from fbpic_amd.main import GpuMemoryManager
Nm = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ppc = tuple(int(v) for v in sys.argv[2].split(',')) if len(sys.argv) > 2 else (2, 4, 4)
sim = helpers.uniform_plasma_sim(1024, 128, Nm, ppc, 'linear', seed=0)
default = _capi.lib()
libs = [('default', default)]
for path in sorted(glob.glob(os.path.join(ROOT, 'fbpic_amd', 'csrc', 'variants', '*.so'))):
    l = ctypes.CDLL(path)
    for name, (res, args) in _capi._SIGNATURES.items():
        f = getattr(l, name); f.restype = res; f.argtypes = args
    libs.append((os.path.basename(path).replace('libfbpic_amd_', '').replace('.so', ''), l))
res = {n: {} for n, _ in libs}
RS = tuple(int(v) for v in os.environ.get('KNOCK_R', '1,2,3').split(','))
REPS = int(os.environ.get('KNOCK_REPS', '10'))
with GpuMemoryManager(sim):
    sim.step(int(os.environ.get("KNOCK_AGE", "24")))
    fld, comm = sim.fld, sim.comm
    for _ in range(8):
        if all(r in res['default'] for r in RS):
            break
        r = sim.ptcl[0]._cycle_since_sort
        if r in RS and r not in res['default']:
            saved = [getattr(s, k).clone() for s in sim.ptcl for k in ('ux', 'uy', 'uz', 'inv_gamma')]
            per = [(s.cycle_sort_period, s.cycle_stray_limit, s._cycle_since_sort, s.cycle_bad_limit) for s in sim.ptcl]
            wz = (fld.interp[0].zmin, fld.interp[0].zmax)
            for name, l in libs:
                _capi._lib = l
                ms = []
                for rep in range(REPS):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for s in sim.ptcl:
                        s.cycle_sort_period, s.cycle_stray_limit, s.cycle_bad_limit = 10 ** 9, 2.0, 2.0
                    e0.record()
                    for s in sim.ptcl:
                        if s.q != 0:
                            s.cycle(fld, comm, 0., store_fields=False, wrap_z=wz)
                    e1.record(); torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                i = 0
                for s, pr in zip(sim.ptcl, per):
                    s.cycle_sort_period, s.cycle_stray_limit, s._cycle_since_sort, s.cycle_bad_limit = pr
                    for k in ('ux', 'uy', 'uz', 'inv_gamma'):
                        getattr(s, k).copy_(saved[i]); i += 1
                fld.erase_source_records()
                ms = sorted(ms[2:])
                res[name][r] = ms[len(ms) // 2]
            _capi._lib = default
        sim.step(1)
for name, _ in libs:
    print('%-14s' % name + ' | '.join('r=%d %.4f ms' % (r, m) for r, m in sorted(res[name].items())), flush=True)
