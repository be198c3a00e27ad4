"""Pass-through stand-in for `numba`, used ONLY in the build container to run
the reference's plain-Python kernel bodies interpreted (golden-vector capture).
Test infrastructure: never imported by the product. Contains no reference code."""
import numpy as _np
__version__ = "0.59.0"
class _Cfg:
    NUMBA_NUM_THREADS = 1
config = _Cfg()
def _deco(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
njit = jit = _deco
def vectorize(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return _np.vectorize(a[0])
    return lambda f: _np.vectorize(f)
prange = range
int64 = _np.int64
float64 = _np.float64
void = None
from . import cuda  # noqa: E402
