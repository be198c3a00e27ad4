"""Host-side (NumPy) helpers for the spectral grid: finite-order modified wavenumbers
and the spatial reach of the PSATD stencil.  Setup-time only.

Restates fbpic/fields/utility_methods.py:11-185 (get_modified_k, stencil_reach,
get_stencil_reach); pinned by tests/golden/grid_setup.npz.
"""
import numpy as np
from scipy.constants import c


def get_modified_k(k, n_order, dz):
    """[k] = sum_{n=1..m} a_n sin(n k dz)/(n dz), a_n = -(m+1-n)/(m+n) a_{n-1}, a_0 = -2,
    for a centred stencil of order n_order = 2m; n_order = -1 means infinite order."""
    if n_order == -1:
        return k
    if n_order % 2 == 1 or n_order <= 0:
        raise ValueError('Invalid n_order: %d' % n_order)
    m = int(n_order / 2)
    a = np.zeros(m + 1)
    a[0] = -2.
    for n in range(1, m + 1):
        a[n] = -(m + 1 - n) * 1. / (m + n) * a[n - 1]
    n_arr = np.arange(1, m + 1)
    s = np.sin(k[:, np.newaxis] * n_arr[np.newaxis, :] * dz) / (n_arr[np.newaxis, :] * dz)
    return np.tensordot(s, a[1:], axes=(-1, -1))


def stencil_reach(kz, kperp, cdt, v_comoving, use_galilean):
    """Number of cells after which the real-space PSATD stencil (cos / sin coefficients
    at one kperp) has decayed to machine precision."""
    k = np.sqrt(kz**2 + kperp**2)
    if use_galilean is True:
        theta = np.exp(1.j * np.abs(v_comoving) * kz * cdt / c / 2)
    else:
        theta = np.ones_like(kz)
    cos_st = np.fft.ifft(theta**2 * np.cos(k * cdt))
    sin_z = np.fft.ifft(np.where(k == 0, kz, theta**2 * np.sin(k * cdt) / k * kz))
    sin_p = np.fft.ifft(np.where(k == 0, kperp, theta**2 * np.sin(k * cdt) / k * kperp))
    alpha = np.sqrt(np.abs(cos_st)**2 + np.abs(sin_z)**2 + np.abs(sin_p)**2)
    return int(np.where(np.abs(alpha)[:int(alpha.shape[0] / 2)] < 1.e-16)[0][0])


def get_stencil_reach(Nz, dz, cdt, n_order, v_comoving, use_galilean):
    """Stencil reach for an Nz-cell grid, evaluated at kperp = 0.5 as the reference does."""
    real_kz = 2 * np.pi * np.fft.fftfreq(Nz, d=dz)
    return stencil_reach(get_modified_k(real_kz, n_order, dz=dz), 0.5, cdt,
                         v_comoving, use_galilean)
