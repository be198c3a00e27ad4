"""CPU-only: libfbpic_amd.so builds for gfx950, loads, and exports exactly the entry points
declared in include/fbpic_amd.h; the product refuses to compute without a GPU."""
import os
import re
import numpy as np
import pytest
from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'fbpic_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(fb_[a-z_A-Z0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from fbpic_amd import _capi
    lib = _capi.lib()
    decl = _declared()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(lib, name), name
    assert sorted(_capi.EXPORTS) == decl
    assert lib.fb_abi_version() == _capi.ABI_VERSION


def test_no_cpu_fallback():
    """Without a GPU every compute path raises instead of silently running on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from scipy.constants import c
    from fbpic_amd import _capi
    from fbpic_amd.main import Simulation
    sim = Simulation(16, 16e-6, 8, 8e-6, 2, 1e-6 / c, 0, 16e-6, 0, 8e-6, 2, 2, 4, 1e24)
    with pytest.raises(_capi.BackendError):
        sim.step(1)
    with pytest.raises(_capi.BackendError):
        sim.fld.interp2spect('E')
    with pytest.raises(_capi.BackendError):
        sim.ptcl[0].push_x(1e-15)
    with pytest.raises(ValueError):
        Simulation(16, 16e-6, 8, 8e-6, 2, 1e-6 / c, use_cuda=False)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'fbpic_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt.lower().replace('no cpu oracle', ''), os.path.join(dirpath, f)


def test_fft_length_dispatch_tables():
    """Host-only queries of the library: which z lengths take the LDS kernel, which the
    generic pass-per-launch FFT, and the smooth convolution length of the Bluestein fallback."""
    from fbpic_amd import _capi
    from fbpic_amd.fields.spectral_transform.fourier import _smooth_length
    lib = _capi.lib()
    lds = [n for n in range(2, 5000) if lib.fb_zfft_supported(n)]
    assert lds == [64, 128, 256, 512, 576, 1024, 1152, 2048, 2304, 4096]
    assert lib.fb_fft_generic_supported(4416) and lib.fb_fft_generic_supported(2 * 3 * 5 * 7 * 11 * 13)
    assert lib.fb_fft_generic_supported(31 * 29 * 4)
    assert not lib.fb_fft_generic_supported(4288) and not lib.fb_fft_generic_supported(37)
    for n in (1, 2, 97, 8575, 8237, 16385):
        m = _smooth_length(n)
        assert m >= n
        k = m
        for p in (2, 3, 5):
            while k % p == 0:
                k //= p
        assert k == 1
        assert all(_smooth_length(n) <= c for c in (1 << (n - 1).bit_length(),))


def test_cpu_baseline_uses_the_cores_it_is_given():
    import bench
    n = bench.available_cores()
    assert 1 <= n <= len(__import__('os').sched_getaffinity(0))


def test_bench_roofline_object_is_stable_between_tied_kernels():
    """bench.roofline(): per-entry-point table + the entry point the `roofline` object
    describes.  gather+push, the J deposition and the sort take the same time within noise;
    the object names gather+push whenever it is within 5 % of the longest, else the longest.
    The PMC traffic comes from the newest profiles/r*_pmc_* pair."""
    import bench
    a = [0, 0, 4194304] + [None] * 30          # args of a launch: a[2] = number of particles
    for g, d, s, expect in ((0.127, 0.1276, 0.101, 'fb_gather_push'),
                            (0.127, 0.120, 0.1271, 'fb_gather_push'),
                            (0.110, 0.130, 0.100, 'fb_deposit_J_rank_next')):
        kern = {'fb_gather_push': [(g, a)] * 10, 'fb_deposit_J_rank_next': [(d, a)] * 10,
                'fb_push_x_bin_sort_particles': [(s, a)] * 10, 'fb_erase': [(0.005, a)] * 10}
        roof, table = bench.roofline(kern)
        assert roof['kernel'] == expect
        assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0 and roof['unit'] == 'GB/s'
        assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-12
        assert set(table) == set(kern) and 'frac' not in table['fb_erase']
    roof, _ = bench.roofline({'fb_gather_push': [(0.127, a)] * 3})
    # 112 B per particle when E, B stay in registers (a[19] is None)
    assert abs(roof['achieved'] - 112 * 4194304 / 0.127e-3 / 1e9) < 1e-6
    # (the newest PMC pass may hold the plain gather of the reference-sequence leg: 72 B per particle = 3.0e8)
    assert roof['traffic'] is None or roof['traffic'] > 2.5e8
