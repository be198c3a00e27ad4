"""External (analytic) fields applied on the particles after the field gathering -- the hook
of fbpic/main.py:472-473, class of fbpic/lpa_utils/external_fields.py:14-213 (lab frame).

The user supplies `field_func(F, x, y, z, t, amplitude, length_scale) -> F'`, written with the
`math` module as the reference asks (it compiles the function with Numba for CPU and GPU).
Here the expression is a host-side hook outside the device hot path: it is evaluated with
NumPy on the particle positions (element by element through `numpy.vectorize` when the
function only accepts scalars) and the result is written back to the device array.  A step
with external fields takes the unfused gather / push_p / push_x launches (Simulation.step).
"""
import numpy as np


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

    def _evaluate(self, F, x, y, z, t, amplitude):
        if not self._scalar_only:
            try:
                out = self.field_func(F, x, y, z, t, amplitude, self.length_scale)
                self._scalar_only = False
                return np.asarray(out, dtype=np.float64)
            except TypeError:
                # written with `math.*` (scalars only), as the reference recommends
                self._scalar_only = True
        func = np.vectorize(self.field_func, otypes=[np.float64])
        return func(F, x, y, z, t, amplitude, self.length_scale)

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
                else:
                    host = [a.detach().cpu().numpy() for a in (field, species.x, species.y, species.z)]
                    new = self._evaluate(*host, t, amplitude)
                    field.copy_(_like(field, new))


def _like(tensor, array):
    import torch
    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)).to(tensor.device)
