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
        self.m, self.dt, self.V = m, dt, V
        w = c * np.sqrt(kz**2 + kr**2)
        origin = (w == 0)                       # the limits w -> 0 are set explicitly below
        inv_w = 1. / np.where(origin, 1., w)
        inv_dt = 1. / dt
        self.C = np.cos(w * dt)
        self.S_w = np.sin(w * dt) * inv_w
        self.S_w[origin] = dt
        moving = (V is not None) and (V != 0)
        if V is not None:
            self._theta_tables(kz, dt, V, use_galilean)
        if moving:
            src = [(x, None) for x in self._moving_source_coefs(kz, w, inv_w, dt, V)]
        else:
            iw2 = inv_w**2
            src = [(1. - self.C, iw2), (self.C - inv_dt * self.S_w, iw2),
                   (1 - inv_dt * self.S_w, iw2)]
        # (value at w = 0) psatd_coefs.py:139-163
        limits = (0.5 * dt**2, -1. / 3 * dt**2, 1. / 6 * dt**2)
        scale = (mu_0 * c**2, c**2 / epsilon_0, c**2 / epsilon_0)
        for name, (a, b), lim, sc in zip(('j_coef', 'rho_prev_coef', 'rho_next_coef'), src,
                                         limits, scale):
            tab = sc * a if b is None else sc * a * b
            tab[origin] = sc * lim
            setattr(self, name, tab)
        self._dev = None

    def _theta_tables(self, kz, dt, V, use_galilean):
        """Phase factors of the moving frame / comoving currents (psatd_coefs.py:76-107):
        T_eb multiplies E, B (Galilean grid), T_cc the currents (comoving currents), T_rho
        integrates the charge over the step."""
        full = np.exp(1.j * kz * V * dt)
        one = np.ones_like(full)
        if use_galilean:
            self.T_eb, self.T_cc = full, one
        else:
            self.T_eb, self.T_cc = one, np.exp(1.j * 0.5 * kz * V * dt)
        if V != 0.:
            ikv = 1.j * kz * V
            ikv[kz == 0] = 1.
            self.T_rho = np.where(kz == 0., -dt, (1. - full) / (self.T_cc * ikv))
            inv_gap = 1. / np.where(full == 1, 1., 1 - full)
            self.j_corr_coef = np.where(kz != 0, (-1.j * kz * V) * inv_gap, 1. / dt)
        else:
            self.T_rho = -dt * one
            self.j_corr_coef = (1. / dt) * np.ones_like(kz)

    def _moving_source_coefs(self, kz, w, iw, dt, V):
        """Source coefficients (J, rho_prev, rho_next; before the mu_0 c^2, c^2/eps_0 factors)
        when the sources are assumed to move at V != 0 (psatd_coefs.py:109-137)."""
        full = np.exp(1.j * kz * V * dt)
        ikv = 1.j * kz * V
        kv2 = kz**2 * V**2
        idt = 1. / dt
        gap = w**2 - kv2
        inv_gap_w = 1. / np.where(gap == 0, 1., gap)
        inv_gap_t = 1. / np.where(full == 1, 1., 1 - full)
        C, S = self.C, self.S_w
        j = 1. / self.T_cc * inv_gap_w * (1. - full * C + ikv * full * S)
        nxt = np.where(kz != 0,
                       inv_gap_w * (1. + ikv * full * S * inv_gap_t
                                    + kv2 * iw**2 * full * inv_gap_t * (1 - C)),
                       1. * iw**2 * (1. - S * idt))
        prv = np.where(kz != 0,
                       self.T_eb * inv_gap_w * (C + ikv * full * S * inv_gap_t
                                                + kv2 * iw**2 * inv_gap_t * (1 - C)),
                       1. * iw**2 * (C - S * idt))
        return j, prv, nxt

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
