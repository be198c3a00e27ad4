"""External (analytic) fields applied on the particles after the field gathering -- the hook
of fbpic/main.py:472-473, class of fbpic/lpa_utils/external_fields.py:14-213 (lab frame).

The user supplies `field_func(F, x, y, z, t, amplitude, length_scale) -> F'`, written with the
`math` module as the reference asks (it compiles the function with Numba for CPU and GPU).
Here the same function object is evaluated on the device arrays: its `math` / `numpy` globals
are rebound to a namespace that maps sin, exp, sqrt ... onto the torch element-wise kernels, so
the expression runs on whole particle arrays in HBM without a host round trip.  A function
that cannot run that way (data-dependent `if`, a call into a C extension ...) is evaluated on
the host with NumPy (`numpy.vectorize` when it only accepts scalars) and the result is copied
back.  A step with external fields takes the unfused gather / push_p / push_x launches
(Simulation.step).
"""
import math
import types
import numpy as np


_RENAMED = {'fabs': 'abs', 'pow': 'pow', 'atan2': 'atan2', 'hypot': 'hypot'}


class _DeviceMath(object):
    """`math.<name>` over device arrays: plain numbers go to the real `math`, anything else to
    the torch kernel of the same name."""
    pi, e, inf, nan, tau = math.pi, math.e, math.inf, math.nan, math.tau

    def __getattr__(self, name):
        import torch
        scalar_fn = getattr(math, name, None)
        array_fn = getattr(torch, _RENAMED.get(name, name), None)
        if array_fn is None:
            raise AttributeError(name)

        def call(*args):
            if scalar_fn is not None and all(isinstance(a, (int, float)) for a in args):
                return scalar_fn(*args)
            ref = next(a for a in args if isinstance(a, torch.Tensor))
            return array_fn(*[a if isinstance(a, torch.Tensor)
                              else torch.as_tensor(a, dtype=ref.dtype, device=ref.device) for a in args])
        return call


def _on_device(func):
    """The same code object with `math` / `numpy` / `np` bound to _DeviceMath."""
    if not isinstance(func, types.FunctionType):
        return None
    g = dict(func.__globals__)
    shim = _DeviceMath()
    for k, v in list(g.items()):
        if v is math or v is np:
            g[k] = shim
    out = types.FunctionType(func.__code__, g, func.__name__, func.__defaults__, func.__closure__)
    out.__kwdefaults__ = func.__kwdefaults__
    return out


class ExternalField(object):
    def __init__(self, field_func, fieldtype, amplitude, length_scale, species=None,
                 gamma_boost=None):
        if fieldtype not in ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz'):
            raise ValueError("`fieldtype` should be 'Ex', 'Ey', 'Ez', 'Bx', 'By' or 'Bz'")
        if gamma_boost is not None:
            raise NotImplementedError('boosted-frame conversion of external fields is outside '
                                      'the fbpic_amd scope')
        self.field_func = field_func
        self.fieldtypes_and_amplitudes = [(fieldtype, amplitude)]
        self.length_scale = length_scale
        self.species = species
        self._scalar_only = None
        self._device_func = None      # None: not tried yet, False: host evaluation only

    def _evaluate(self, F, x, y, z, t, amplitude):
        if not self._scalar_only:
            try:
                out = self.field_func(F, x, y, z, t, amplitude, self.length_scale)
                self._scalar_only = False
                return np.asarray(out, dtype=np.float64)
            except (TypeError, ValueError):
                # written with `math.*` (scalars only), as the reference recommends
                self._scalar_only = True
        func = np.vectorize(self.field_func, otypes=[np.float64])
        return func(F, x, y, z, t, amplitude, self.length_scale)

    def _evaluate_on_device(self, field, species, t, amplitude):
        import torch
        if self._device_func is None:
            self._device_func = _on_device(self.field_func) or False
        if self._device_func is False:
            return False
        try:
            new = self._device_func(field, species.x, species.y, species.z, t, amplitude,
                                    self.length_scale)
        except (TypeError, ValueError, RuntimeError, AttributeError):
            self._device_func = False
            return False
        field.copy_(torch.as_tensor(new, dtype=field.dtype, device=field.device).expand_as(field))
        return True

    def apply_expression(self, ptcl, t):
        """Called at each time step after the field gathering (reference :174-213)."""
        for species in ptcl:
            if (self.species is not None) and (species is not self.species):
                continue
            if species.Ntot <= 0:
                continue
            for fieldtype, amplitude in self.fieldtypes_and_amplitudes:
                field = getattr(species, fieldtype)
                if isinstance(field, np.ndarray):
                    field[:] = self._evaluate(field, species.x, species.y, species.z, t, amplitude)
                elif not self._evaluate_on_device(field, species, t, amplitude):
                    host = [a.detach().cpu().numpy() for a in (field, species.x, species.y, species.z)]
                    new = self._evaluate(*host, t, amplitude)
                    field.copy_(_like(field, new))


def _like(tensor, array):
    import torch
    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)).to(tensor.device)
