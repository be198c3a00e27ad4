"""PSATD coefficient tables, built once on the host with NumPy using the reference's formulas
and limits at w = 0, then uploaded.  Restates fbpic/fields/psatd_coefs.py:15-177: the standard
scheme (V is None: real tables) and the Galilean / comoving-current schemes (V given: the
Theta coefficients T_eb, T_cc, T_rho, the corrected-current coefficient j_corr_coef, and
complex j_coef / rho_prev_coef / rho_next_coef)."""
import numpy as np
from scipy.constants import c, mu_0, epsilon_0
from .. import _capi

TABLES = ('C', 'S_w', 'j_coef', 'rho_prev_coef', 'rho_next_coef')
TABLES_COMOVING = ('T_eb', 'T_cc', 'T_rho', 'j_corr_coef')


class PsatdCoeffs(object):
    def __init__(self, kz, kr, m, dt, Nz, Nr, V=None, use_galilean=False, use_cuda=False):
        i = 1.j
        self.m = m
        self.dt = dt
        self.V = V
        inv_dt = 1. / dt
        w = c * np.sqrt(kz**2 + kr**2)
        at0 = (w == 0)
        inv_w = 1. / np.where(at0, 1., w)
        self.C = np.cos(w * dt)
        self.S_w = np.sin(w * dt) * inv_w
        self.S_w[at0] = dt
        if V is not None:
            # psatd_coefs.py:76-137
            T2 = np.exp(i * kz * V * dt)
            if use_galilean is False:
                T = np.exp(i * 0.5 * kz * V * dt)
            if use_galilean:
                self.T_eb = T2
                self.T_cc = np.ones_like(T2)
            else:
                self.T_cc = T
                self.T_eb = np.ones_like(T2)
            if V != 0.:
                i_kz_V = i * kz * V
                i_kz_V[kz == 0] = 1.
                self.T_rho = np.where(kz == 0., -dt, (1. - T2) / (self.T_cc * i_kz_V))
            else:
                self.T_rho = -dt * np.ones_like(kz)
            if V != 0.:
                inv_w_kzV = 1. / np.where((w**2 - kz**2 * V**2) == 0, 1., (w**2 - kz**2 * V**2))
                inv_1_T2 = 1. / np.where(T2 == 1, 1., 1 - T2)
                xi_1 = 1. / self.T_cc * inv_w_kzV * (1. - T2 * self.C + i * kz * V * T2 * self.S_w)
                xi_2 = np.where(
                    kz != 0,
                    inv_w_kzV * (1. + i * kz * V * T2 * self.S_w * inv_1_T2
                                 + kz**2 * V**2 * inv_w**2 * T2 * inv_1_T2 * (1 - self.C)),
                    1. * inv_w**2 * (1. - self.S_w * inv_dt))
                xi_3 = np.where(
                    kz != 0,
                    self.T_eb * inv_w_kzV * (self.C + i * kz * V * T2 * self.S_w * inv_1_T2
                                             + kz**2 * V**2 * inv_w**2 * inv_1_T2 * (1 - self.C)),
                    1. * inv_w**2 * (self.C - self.S_w * inv_dt))
                self.j_corr_coef = np.where(kz != 0, (-i * kz * V) * inv_1_T2, inv_dt)
            else:
                self.j_corr_coef = inv_dt * np.ones_like(kz)
        if V is None or V == 0:
            self.j_coef = mu_0 * c**2 * (1. - self.C) * inv_w**2
        else:
            self.j_coef = mu_0 * c**2 * (xi_1)
        self.j_coef[at0] = mu_0 * c**2 * (0.5 * dt**2)
        if V is None or V == 0:
            self.rho_prev_coef = c**2 / epsilon_0 * (self.C - inv_dt * self.S_w) * inv_w**2
        else:
            self.rho_prev_coef = c**2 / epsilon_0 * (xi_3)
        self.rho_prev_coef[at0] = c**2 / epsilon_0 * (-1. / 3 * dt**2)
        if V is None or V == 0:
            self.rho_next_coef = c**2 / epsilon_0 * (1 - inv_dt * self.S_w) * inv_w**2
        else:
            self.rho_next_coef = c**2 / epsilon_0 * (xi_2)
        self.rho_next_coef[at0] = c**2 / epsilon_0 * (1. / 6 * dt**2)
        self._dev = None

    def device_tables(self):
        """Device copies d_C, d_S_w, ... (uploaded on first use).  With V set, the three
        source coefficients and the four Theta tables are uploaded as complex128."""
        if self._dev is None:
            self._dev = {}
            for k in TABLES:
                a = getattr(self, k)
                if self.V is not None and k in ('j_coef', 'rho_prev_coef', 'rho_next_coef'):
                    a = np.ascontiguousarray(a, dtype=np.complex128)
                self._dev[k] = _capi.to_device(a)
            if self.V is not None:
                for k in TABLES_COMOVING:
                    self._dev[k] = _capi.to_device(
                        np.ascontiguousarray(getattr(self, k), dtype=np.complex128))
            for k, v in self._dev.items():
                setattr(self, 'd_' + k, v)
        return self._dev
