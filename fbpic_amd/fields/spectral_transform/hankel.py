"""Discrete Hankel transform along r: host-side construction of the (Nr, Nr) matrices,
device-side application as an fp64-MFMA GEMM (csrc/hankel.hip).

Matrix definitions restate fbpic/fields/spectral_transform/hankel.py:74-122 (transposed
form: a transform is `F . M`); pinned by tests/golden/grid_setup.npz.  The apply
replaces the split / cuBLAS dgemm / merge sequence of hankel.py:196-236.
"""
import numpy as np
from scipy.special import jn, jn_zeros
from ... import _capi


def hankel_matrices(p, m, Nr, rmax):
    """Return (M, invM, nu, r) for the order-p transform used with azimuthal mode m."""
    if m not in (p - 1, p, p + 1):
        raise ValueError('m must be either p-1, p or p+1')
    if m != 0:
        # 0 is a root of J_m for m != 0 and carries the k_perp = 0 mode
        alphas = np.hstack((np.array([0.]), jn_zeros(m, Nr - 1)))
    else:
        alphas = jn_zeros(m, Nr)
    nu = 1. / (2 * np.pi * rmax) * alphas
    r = (rmax * 1. / Nr) * (np.arange(Nr) + 0.5)
    p_denom = p + 1 if p == m else p
    denom = np.pi * rmax**2 * jn(p_denom, alphas)**2
    num = jn(p, 2 * np.pi * r[np.newaxis, :] * nu[:, np.newaxis])
    invM = np.empty((Nr, Nr))
    if m != 0:
        invM[1:, :] = num[1:, :] / denom[1:, np.newaxis]
        if p == m - 1:
            invM[0, :] = r**(m - 1) * 1. / (np.pi * rmax**(m + 1))
        else:
            invM[0, :] = 0.
    else:
        invM[:, :] = num[:, :] / denom[:, np.newaxis]
    if m != 0 and p != m - 1:
        M = np.empty((Nr, Nr))
        M[:, 1:] = np.linalg.pinv(invM[1:, :])
        M[:, 0] = 0.
    else:
        M = np.linalg.inv(invM)
    return M, invM, nu, r


class DHT(object):
    """Same constructor / methods as the reference's DHT (hankel.py:25-243)."""

    def __init__(self, p, m, Nr, Nz, rmax, use_cuda=True):
        self.p, self.m, self.Nr, self.Nz, self.rmax = p, m, Nr, Nz, rmax
        self.use_cuda = use_cuda
        self.M, self.invM, self.nu, self.r = hankel_matrices(p, m, Nr, rmax)
        self.d_M = None
        self.d_invM = None

    def get_r(self):
        return self.r

    def get_nu(self):
        return self.nu

    def device_matrices(self):
        if self.d_M is None:
            self.d_M = _capi.to_device(self.M)
            self.d_invM = _capi.to_device(self.invM)
        return self.d_M, self.d_invM

    def _apply(self, src, dst, mat):
        Nz, Nr = src.shape
        rc = _capi.lib().fb_hankel(1, _capi.ptr_array([src]), _capi.row_stride(src),
                                   _capi.ptr_array([dst]), _capi.row_stride(dst),
                                   _capi.ptr_array([mat]), 1.0, Nz, Nr, _capi.stream())
        _capi.check(rc, 'fb_hankel')

    def transform(self, F, G):
        """G = DHT(F) for device arrays of shape (Nz, Nr) (F and G must not alias)."""
        self._apply(F, G, self.device_matrices()[0])

    def inverse_transform(self, G, F):
        self._apply(G, F, self.device_matrices()[1])
