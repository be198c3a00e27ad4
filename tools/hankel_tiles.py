#!/usr/bin/env python3
"""Workgroup shapes of the Hankel GEMM (k_hankel) against each other: device time of fb_hankel (plain jobs) and
fb_hankel_pm_to_rt (dual jobs) for the tile ids of a -DFB_HANKEL_TILE_PROBE build
(bash tools/variant.sh hkprobe hankel.hip -DFB_HANKEL_TILE_PROBE), selected by FBPIC_AMD_HANKEL_TILE at the first
launch of a process - so every (case, tile) pair runs in a process of its own.
usage: hankel_tiles.py                 the whole scan (spawns itself)
       hankel_tiles.py --one plain|dual Nz,Nr,njobs   one measurement with the tile of the environment"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, 'fbpic_amd', 'csrc', 'variants', 'libfbpic_amd_%s.so' % os.environ.get('FBPIC_AMD_HK_VARIANT', 'hkprobe'))
if os.environ.get('FBPIC_AMD_HK_VARIANT') == 'default':
    VARIANT = os.path.join(ROOT, 'fbpic_amd', 'csrc', 'libfbpic_amd.so')


def one(kind, Nz, Nr, njobs, reps=20):
    import ctypes, torch
    from fbpic_amd import _capi
    lib = ctypes.CDLL(VARIANT)
    for name, (res, args) in _capi._SIGNATURES.items():
        f = getattr(lib, name); f.restype = res; f.argtypes = args
    PAD = 8
    nf = njobs * (2 if kind == 'dual' else 1)
    rs = nf * Nr + PAD

    def slab(fill):
        t = (torch.randn if fill else torch.zeros)(Nz * rs, dtype=torch.complex128, device='cuda')
        return t.as_strided((Nz, nf, Nr), (rs, Nr, 1))
    a, b = slab(True), slab(False)
    mats = [torch.randn((Nr, Nr), dtype=torch.float64, device='cuda') for _ in range(nf)]
    st = _capi.stream()
    if kind == 'plain':
        ins, outs, mp = (_capi.ptr_array([a[:, j, :] for j in range(nf)]), _capi.ptr_array([b[:, j, :] for j in range(nf)]),
                         _capi.ptr_array(mats))

        def call():
            _capi.check(lib.fb_hankel(njobs, ins, rs, outs, rs, mp, 1.0, Nz, Nr, st), 'hk')
        flop = 4.0 * Nz * Nr * Nr * njobs
    else:
        ins = _capi.ptr_array([a[:, 2 * j, :] for j in range(njobs)])
        ins2 = _capi.ptr_array([a[:, 2 * j + 1, :] for j in range(njobs)])
        outs = _capi.ptr_array([b[:, 2 * j, :] for j in range(njobs)])
        outs2 = _capi.ptr_array([b[:, 2 * j + 1, :] for j in range(njobs)])
        mp = _capi.ptr_array([mats[2 * j] for j in range(njobs)])
        mp2 = _capi.ptr_array([mats[2 * j + 1] for j in range(njobs)])

        def call():
            _capi.check(lib.fb_hankel_pm_to_rt(njobs, ins, ins2, rs, outs, outs2, rs, mp, mp2, 1.0, Nz, Nr, st), 'hk')
        flop = 8.0 * Nz * Nr * Nr * njobs
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    # check against a plain product (first job, a few rows)
    if kind == 'plain':
        ref = a[:64, 0, :] @ mats[0].to(torch.complex128)
        err = float((b[:64, 0, :] - ref).abs().max() / ref.abs().max())
    else:
        p_ = a[:64, 0, :] @ mats[0].to(torch.complex128)
        m_ = a[:64, 1, :] @ mats[1].to(torch.complex128)
        err = float(max((b[:64, 0, :] - (p_ + m_)).abs().max(), (b[:64, 1, :] - 1j * (p_ - m_)).abs().max()) / p_.abs().max())
    print('%-5s Nz=%d Nr=%d jobs=%2d tile=%s lib=%s : %8.1f us  %5.1f TFLOP/s  err %.1e'
          % (kind, Nz, Nr, njobs, os.environ.get('FBPIC_AMD_HANKEL_TILE', '0'), os.environ.get('FBPIC_AMD_HK_VARIANT', 'hkprobe'), us, flop / (us * 1e-6) / 1e12, err), flush=True)


if '--one' in sys.argv:
    i = sys.argv.index('--one')
    Nz, Nr, nj = (int(v) for v in sys.argv[i + 2].split(','))
    one(sys.argv[i + 1], Nz, Nr, nj)
else:
    cases = [('plain', '4416,256,8', (0, 13, 14, 4, 23, 24, 26, 43, 44)), ('plain', '4416,256,12', (0, 13, 14, 24)),
             ('plain', '2048,512,16', (0, 13, 14, 23, 24, 26)), ('plain', '1152,128,12', (0, 6, 33, 34, 4, 23, 24)),
             ('plain', '1024,128,12', (0, 33, 34, 4, 23, 24)),
             ('dual', '4416,256,4', (0, 13, 14, 23, 24, 33, 34)), ('dual', '2048,512,8', (0, 13, 14, 23, 24, 33, 34)),
             ('dual', '1152,128,4', (0, 13, 14, 23, 24, 33, 34))]
    if '--cases' in sys.argv:      # --cases "plain:4416,256,8:0,13;dual:..."
        cases = [(c.split(':')[0], c.split(':')[1], tuple(int(v) for v in c.split(':')[2].split(',')))
                 for c in sys.argv[sys.argv.index('--cases') + 1].split(';')]
    for kind, size, tiles in cases:
        for tl in tiles:
            env = dict(os.environ, FBPIC_AMD_HANKEL_TILE=str(tl))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--one', kind, size], env=env,
                               capture_output=True, text=True, timeout=300)
            out = [l for l in r.stdout.split('\n') if l.startswith(('plain', 'dual'))]
            print(out[0] if out else 'FAILED tile %d %s %s: %s' % (tl, kind, size, r.stderr[-300:]), flush=True)
