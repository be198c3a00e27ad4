"""Spectral grid of one azimuthal mode: wavenumber / filter tables (host, NumPy) and the
eleven spectral field arrays (thirteen with the cross-deposition correction); device kernels for current correction, PSATD push, rho
shift and filtering (csrc/fields.hip).

Tables restate fbpic/fields/spectral_grid.py:108-124; the methods replace the CUDA
launches of :219-230 (curl-free correction), :339-355 (push_eb_standard), :416-417
(push_rho), :437-454 (filter).
"""
import numpy as np
from scipy.constants import c, epsilon_0, mu_0
from .. import _capi

SPECT_FIELDS = ('Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next')
# only allocated with current_correction='cross-deposition' (spectral_grid.py:97-99)
CROSS_FIELDS = ('rho_next_z', 'rho_next_xy')
_FILTER_GROUPS = {'E': ('Ep', 'Em', 'Ez'), 'B': ('Bp', 'Bm', 'Bz'), 'J': ('Jp', 'Jm', 'Jz'),
                  'rho_prev': ('rho_prev',), 'rho_next': ('rho_next',),
                  'rho_next_z': ('rho_next_z',), 'rho_next_xy': ('rho_next_xy',)}


class SpectralGrid(object):
    def __init__(self, kz_modified, kr, m, kz_true, dz, dr, current_correction, smoother,
                 use_pml=False, use_cuda=True):
        if use_pml:
            raise NotImplementedError('PML is outside the fbpic_amd scope')
        Nz, Nr = len(kz_modified), len(kr)
        self.Nz, self.Nr, self.m = Nz, Nr, m
        self.use_pml = False
        self.use_cuda = use_cuda
        self.field_names = SPECT_FIELDS
        if current_correction == 'cross-deposition':
            self.field_names = SPECT_FIELDS + CROSS_FIELDS
        for name in self.field_names:
            setattr(self, name, np.zeros((Nz, Nr), dtype='complex'))
        # field solve uses the (finite-order) modified kz, filtering the true kz
        self.kz, self.kr = np.meshgrid(kz_modified, kr, indexing='ij')
        self.filter_array_z, self.filter_array_r = smoother.get_filter_array(kz_true, kr, dz, dr)
        origin = (self.kz == 0) & (self.kr == 0)
        self.inv_k2 = 1. / np.where(origin, 1., self.kz**2 + self.kr**2)
        self.inv_k2[origin] = 0.
        self.field_shift = np.exp(1.j * kz_true * dz)
        self._tables_up = False

    def upload_tables(self):
        if not self._tables_up:
            for k in ('kz', 'kr', 'inv_k2', 'filter_array_z', 'filter_array_r', 'field_shift'):
                setattr(self, 'd_' + k, _capi.to_device(getattr(self, k)))
            self._tables_up = True

    def correct_currents(self, dt, ps, current_correction):
        if current_correction == 'cross-deposition':
            p = _capi.ptr
            common = (p(self.rho_prev), p(self.rho_next), p(self.rho_next_z), p(self.rho_next_xy),
                      p(self.Jp), p(self.Jm), p(self.Jz), _capi.row_stride(self.Jp),
                      p(self.d_kz), p(self.d_kr))
            if ps.V is not None:        # spectral_grid.py:250-258
                t = ps.device_tables()
                rc = _capi.lib().fb_correct_currents_crossdeposition_comoving(
                    *common, p(t['j_corr_coef']), p(t['T_eb']), p(t['T_cc']), self.Nz, self.Nr,
                    _capi.stream())
                _capi.check(rc, 'fb_correct_currents_crossdeposition_comoving')
            else:                       # spectral_grid.py:231-238
                rc = _capi.lib().fb_correct_currents_crossdeposition_standard(
                    *common, 1. / dt, self.Nz, self.Nr, _capi.stream())
                _capi.check(rc, 'fb_correct_currents_crossdeposition_standard')
            return
        if current_correction != 'curl-free':
            raise ValueError('Unknown current correction: %s' % current_correction)
        if ps.V is not None:
            # Galilean / comoving-current scheme (spectral_grid.py:240-247)
            t = ps.device_tables()
            rc = _capi.lib().fb_correct_currents_curlfree_comoving(
                _capi.ptr(self.rho_prev), _capi.ptr(self.rho_next), _capi.ptr(self.Jp),
                _capi.ptr(self.Jm), _capi.ptr(self.Jz), _capi.row_stride(self.Jp),
                _capi.ptr(self.d_kz), _capi.ptr(self.d_kr), _capi.ptr(self.d_inv_k2),
                _capi.ptr(t['j_corr_coef']), _capi.ptr(t['T_eb']), _capi.ptr(t['T_cc']),
                self.Nz, self.Nr, _capi.stream())
            _capi.check(rc, 'fb_correct_currents_curlfree_comoving')
            return
        rc = _capi.lib().fb_correct_currents_curlfree_standard(
            _capi.ptr(self.rho_prev), _capi.ptr(self.rho_next), _capi.ptr(self.Jp),
            _capi.ptr(self.Jm), _capi.ptr(self.Jz), _capi.row_stride(self.Jp),
            _capi.ptr(self.d_kz), _capi.ptr(self.d_kr), _capi.ptr(self.d_inv_k2), 1. / dt,
            self.Nz, self.Nr, _capi.stream())
        _capi.check(rc, 'fb_correct_currents_curlfree_standard')

    def correct_divE(self):
        raise NotImplementedError('correct_divE is outside the fbpic_amd hot path')

    def push_eb_with(self, ps, use_true_rho=False):
        assert self.m == ps.m
        t = ps.device_tables()
        p = _capi.ptr
        if ps.V is not None:
            # Galilean / comoving-current scheme (spectral_grid.py:357-368)
            rc = _capi.lib().fb_push_eb_comoving(
                p(self.Ep), p(self.Em), p(self.Ez), p(self.Bp), p(self.Bm), p(self.Bz),
                p(self.Jp), p(self.Jm), p(self.Jz), p(self.rho_prev), p(self.rho_next),
                _capi.row_stride(self.Ep), p(t['rho_prev_coef']), p(t['rho_next_coef']),
                p(t['j_coef']), p(t['C']), p(t['S_w']), p(t['T_eb']), p(t['T_cc']), p(t['T_rho']),
                p(self.d_kr), p(self.d_kz), ps.dt, ps.V, int(bool(use_true_rho)), c, epsilon_0,
                mu_0, self.Nz, self.Nr, _capi.stream())
            _capi.check(rc, 'fb_push_eb_comoving')
            return
        rc = _capi.lib().fb_push_eb_standard(
            p(self.Ep), p(self.Em), p(self.Ez), p(self.Bp), p(self.Bm), p(self.Bz),
            p(self.Jp), p(self.Jm), p(self.Jz), p(self.rho_prev), p(self.rho_next),
            _capi.row_stride(self.Ep),
            p(t['rho_prev_coef']), p(t['rho_next_coef']), p(t['j_coef']), p(t['C']), p(t['S_w']),
            p(self.d_kr), p(self.d_kz), ps.dt, int(bool(use_true_rho)), c, epsilon_0, mu_0,
            self.Nz, self.Nr, _capi.stream())
        _capi.check(rc, 'fb_push_eb_standard')

    def push_rho(self):
        rc = _capi.lib().fb_push_rho(_capi.ptr(self.rho_prev), _capi.ptr(self.rho_next),
                                     _capi.row_stride(self.rho_prev), self.Nz, self.Nr,
                                     _capi.stream())
        _capi.check(rc, 'fb_push_rho')

    def filter(self, fieldtype):
        if fieldtype not in _FILTER_GROUPS:
            raise ValueError('Invalid string for fieldtype: %s' % fieldtype)
        arrs = [getattr(self, k) for k in _FILTER_GROUPS[fieldtype]]
        rc = _capi.lib().fb_filter(len(arrs), _capi.ptr_array(arrs), _capi.row_stride(arrs[0]),
                                   _capi.ptr(self.d_filter_array_z),
                                   _capi.ptr(self.d_filter_array_r), self.Nz, self.Nr,
                                   _capi.stream())
        _capi.check(rc, 'fb_filter')
