"""Binomial smoother expressed in spectral space (host-side table construction).
Restates fbpic/fields/smoothing.py:10-94."""
import numpy as np


class BinomialSmoother(object):
    def __init__(self, n_passes=1, compensator=False):
        if type(n_passes) is int:
            self.n_passes = {'z': n_passes, 'r': n_passes}
        elif type(n_passes) is dict:
            self.n_passes = n_passes
        else:
            raise ValueError('Invalid argument `n_passes`')
        if type(compensator) is bool:
            self.compensator = {'z': compensator, 'r': compensator}
        elif type(compensator) is dict:
            self.compensator = compensator
        else:
            raise ValueError('Invalid argument `compensator`')

    def _one_axis(self, k, d, axis):
        s2 = np.sin(0.5 * k * d)**2
        n = self.n_passes[axis]
        f = (1. - s2)**n
        if self.compensator[axis]:
            f *= (1. + n * s2)
        return f

    def get_filter_array(self, kz, kr, dz, dr):
        """(filter_z[len(kz)], filter_r[len(kr)]); kz must be the TRUE wavenumbers."""
        return self._one_axis(kz, dz, 'z'), self._one_axis(kr, dr, 'r')
