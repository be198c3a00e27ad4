"""Spatial (interpolation) grid of one azimuthal mode: geometry, cell volumes, Ruyten
shape coefficients (host, NumPy) and the ten field arrays.

On the host the field attributes are NumPy arrays; while the simulation data is on the
GPU they are (Nz, Nr) views into the z-major device slab owned by `Fields`.
Geometry / volumes / Ruyten coefficients restate
fbpic/fields/interpolation_grid.py:58-138 (pinned by tests/golden/grid_setup.npz);
erase / divide_by_volume replace the CUDA launches of :236-296.
"""
import numpy as np
from scipy.special import j1, jn_zeros
from .. import _capi
from .spectral_transform.hankel import hankel_matrices

INTERP_FIELDS = ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz', 'Jr', 'Jt', 'Jz', 'rho')
GROUPS = {'E': ('Er', 'Et', 'Ez'), 'B': ('Br', 'Bt', 'Bz'), 'J': ('Jr', 'Jt', 'Jz'),
          'rho': ('rho',)}


class InterpolationGrid(object):
    def __init__(self, Nz, Nr, m, zmin, zmax, rmax, use_pml=False, use_cuda=True,
                 use_ruyten_shapes=True, use_modified_volume=True):
        if use_pml:
            raise NotImplementedError('PML (open r boundary) is outside the fbpic_amd scope')
        self.Nz, self.Nr, self.m = Nz, Nr, m
        self.use_pml = False
        self.use_cuda = use_cuda
        dr = rmax / Nr
        dz = (zmax - zmin) / Nz
        self.dr, self.dz = dr, dz
        self.invdr, self.invdz = 1. / dr, 1. / dz
        self.rmin, self.rmax = 0., rmax
        self.zmin, self.zmax = zmin, zmax
        nr_vals = np.arange(Nr)
        if use_modified_volume and m == 0:
            # volumes consistent with the mode-0 Hankel quadrature
            alphas = jn_zeros(0, Nr)
            M0 = hankel_matrices(0, 0, Nr, rmax)[0]
            vol = dz * np.array([(M0[nr, :] * 2. / (alphas * j1(alphas))).sum() for nr in nr_vals])
        else:
            r = (0.5 + np.arange(Nr)) * dr
            vol = np.pi * dz * ((r + 0.5 * dr)**2 - (r - 0.5 * dr)**2)
        self.invvol = 1. / vol
        if use_ruyten_shapes:
            norm_vol = vol / (2 * np.pi * dr**2 * dz)
            lin = 6. / (nr_vals + 1) * (np.cumsum(norm_vol) - 0.5 * (nr_vals + 1.)**2 - 1. / 24)
            cub = 6. / (nr_vals + 1) * (np.cumsum(norm_vol) - 0.5 * (nr_vals + 1.)**2 - 1. / 8)
            cub[0] = 6. * (norm_vol[0] - 0.5 - 239. / (15 * 2**7))
        else:
            lin = np.zeros(Nr)
            cub = np.zeros(Nr)
        # leading 0: coefficient of particles in the first half of the first cell
        self.ruyten_linear_coef = np.concatenate((np.array([0.]), lin))
        self.ruyten_cubic_coef = np.concatenate((np.array([0.]), cub))
        for name in INTERP_FIELDS:
            setattr(self, name, np.zeros((Nz, Nr), dtype='complex'))
        self.d_invvol = None
        self.d_ruyten_linear_coef = None
        self.d_ruyten_cubic_coef = None

    def __getattr__(self, name):
        # only reached when the normal lookup fails: Jr / Jt / Jz / rho while Fields.defer_sources
        # holds them back (they are computed from the spectral fields on first use)
        if name in ('Jr', 'Jt', 'Jz', 'rho'):
            owner = self.__dict__.get('_owner')
            if owner is not None and owner._deferred_sources is not None:
                owner.materialize_sources()
                return self.__dict__[name]
        raise AttributeError(name)

    @property
    def z(self):
        return self.zmin + (0.5 + np.arange(self.Nz)) * self.dz

    @property
    def r(self):
        return self.rmin + (0.5 + np.arange(self.Nr)) * self.dr

    def upload_tables(self):
        if self.d_invvol is None:
            self.d_invvol = _capi.to_device(self.invvol)
            self.d_ruyten_linear_coef = _capi.to_device(self.ruyten_linear_coef)
            self.d_ruyten_cubic_coef = _capi.to_device(self.ruyten_cubic_coef)

    def _group(self, fieldtype):
        if fieldtype not in GROUPS:
            raise ValueError('Invalid string for fieldtype: %s' % fieldtype)
        return [getattr(self, k) for k in GROUPS[fieldtype]]

    def erase(self, fieldtype):
        arrs = self._group(fieldtype)
        rc = _capi.lib().fb_erase(len(arrs), _capi.ptr_array(arrs), _capi.row_stride(arrs[0]),
                                  self.Nz, self.Nr, _capi.stream())
        _capi.check(rc, 'fb_erase')

    def divide_by_volume(self, fieldtype):
        if fieldtype not in ('rho', 'J'):
            raise ValueError('Invalid string for fieldtype: %s' % fieldtype)
        arrs = self._group(fieldtype)
        rc = _capi.lib().fb_divide_by_volume(len(arrs), _capi.ptr_array(arrs),
                                             _capi.row_stride(arrs[0]), _capi.ptr(self.d_invvol),
                                             self.Nz, self.Nr, _capi.stream())
        _capi.check(rc, 'fb_divide_by_volume')
