"""PSATD coefficient tables (standard scheme), built once on the host with NumPy using
the reference's formulas and limits at w = 0, then uploaded.
Restates fbpic/fields/psatd_coefs.py:15-177 for V is None."""
import numpy as np
from scipy.constants import c, mu_0, epsilon_0
from .. import _capi

TABLES = ('C', 'S_w', 'j_coef', 'rho_prev_coef', 'rho_next_coef')


class PsatdCoeffs(object):
    def __init__(self, kz, kr, m, dt, Nz, Nr, V=None, use_galilean=False, use_cuda=False):
        if V is not None:
            raise NotImplementedError(
                'Galilean / comoving PSATD is outside the scope of the fbpic_amd hot path')
        self.m = m
        self.dt = dt
        self.V = None
        inv_dt = 1. / dt
        w = c * np.sqrt(kz**2 + kr**2)
        at0 = (w == 0)
        inv_w = 1. / np.where(at0, 1., w)
        self.C = np.cos(w * dt)
        self.S_w = np.sin(w * dt) * inv_w
        self.S_w[at0] = dt
        self.j_coef = mu_0 * c**2 * (1. - self.C) * inv_w**2
        self.j_coef[at0] = mu_0 * c**2 * (0.5 * dt**2)
        self.rho_prev_coef = c**2 / epsilon_0 * (self.C - inv_dt * self.S_w) * inv_w**2
        self.rho_prev_coef[at0] = c**2 / epsilon_0 * (-1. / 3 * dt**2)
        self.rho_next_coef = c**2 / epsilon_0 * (1 - inv_dt * self.S_w) * inv_w**2
        self.rho_next_coef[at0] = c**2 / epsilon_0 * (1. / 6 * dt**2)
        self._dev = None

    def device_tables(self):
        """Device copies d_C, d_S_w, ... (uploaded on first use)."""
        if self._dev is None:
            self._dev = {k: _capi.to_device(getattr(self, k)) for k in TABLES}
            for k, v in self._dev.items():
                setattr(self, 'd_' + k, v)
        return self._dev
