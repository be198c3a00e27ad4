import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / max |b| (the reference's own CPU<->GPU metric,
    tests/test_cpu_gpu_deposition.py:96-98)."""
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.abs(b).max()
    if den == 0:
        return np.abs(a).max()
    return np.abs(a - b).max() / den


_ACHIEVED = []
# Deviations measured on MI355X (the newest tests/golden/achieved_rNN.json: worst figure of the full
# runs of that round on several boxes, tools/make_clamp.py; run-to-run spread <= 3.1x): every
# recorded check is also held to 10x that figure (not below 5e-15, a few tens of ulp), whatever
# wider bound the test states - so that a bound can never again be six orders of magnitude wider
# than what the kernels deliver (in round 3 such a bound hid an open-boundary damping that was
# applied twice per step: 4e-8 under a 1e-9 ... 1e-8 bound).
def newest_clamp_file():
    """tests/golden/achieved_rNN.json with the largest round NUMBER (r100 sorts after r99)."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(GOLDEN, 'achieved_r*.json')):
        m = re.match(r'achieved_r(\d+)\.json$', os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return best[1] if best else None


try:
    import json as _json
    _MEASURED = {k: v for k, v in _json.load(open(newest_clamp_file())).items() if not k.startswith('__')}
except (OSError, ValueError, TypeError):
    _MEASURED = {}


_CLAMP_APPLIES = []


def _clamp_applies():
    """The measured figures are those of MI355X (gfx950) with the HIP 7.0 runtime of this image:
    on another GPU, ROCm or rocFFT build the clamp is not applied (the stated tolerance is)."""
    if not _CLAMP_APPLIES:
        ok = False
        try:
            import torch
            if torch.cuda.is_available():
                arch = getattr(torch.cuda.get_device_properties(0), 'gcnArchName', '')
                ok = arch.startswith('gfx950') and str(getattr(torch.version, 'hip', '')).startswith('7.0')
        except Exception:
            ok = False
        _CLAMP_APPLIES.append(ok)
    return _CLAMP_APPLIES[0]


def achieved(name, err, tol, what=''):
    """Assert err < tol and keep the achieved figure: the list is printed at the end of the run
    (pytest_terminal_summary) and written to gpurun_out/achieved_errors.json, so a bound that is
    wider than what the kernels deliver is visible.  name = None: the id of the running test
    (+ `what`, e.g. 'fields s5')."""
    if name is None:
        name = os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0].split('::', 1)[-1]
        name = name.replace('test_', '', 1)
        if what:
            name += ' ' + what
    ref = _MEASURED.get(name)
    stated = float(tol)
    if ref is not None and _clamp_applies():
        tol = min(stated, max(10. * ref, 5e-15))
    _ACHIEVED.append((name, float(err), float(tol)))
    if err >= tol and err < stated:
        # the CONTRACT (the tolerance the test states) holds; what failed is the
        # regression clamp = 10 x this build's own earlier measurement on MI355X
        raise AssertionError('%s: %.3e is within the stated tolerance %.1e but above the regression clamp %.3e '
                             '(10 x the figure measured in earlier rounds, tests/golden/achieved_r*.json)'
                             % (name, err, stated, tol))
    assert err < tol, (name, err, tol)


def pytest_terminal_summary(terminalreporter):
    if not _ACHIEVED:
        return
    worst = {}
    for name, err, tol in _ACHIEVED:
        if name not in worst or err > worst[name][0]:
            worst[name] = (err, tol)
    terminalreporter.write_line('achieved errors (worst case per check, bound):')
    for name in sorted(worst):
        terminalreporter.write_line('  %-76s %.2e  (< %.0e)' % (name, worst[name][0], worst[name][1]))
    try:
        import json
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'achieved_errors.json'), 'w') as f:
            json.dump({k: {'achieved': v[0], 'bound': v[1]} for k, v in sorted(worst.items())}, f, indent=1)
    except OSError:
        pass


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc
