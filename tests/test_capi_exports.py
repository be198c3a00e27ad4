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
    assert lib.fb_abi_version() == 1


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
