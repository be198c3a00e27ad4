"""Initial macroparticle lattice (host-side, NumPy): synthetic-input generator of every
configuration.  Restates generate_evenly_spaced / unalign_angles of
fbpic/particles/injection/continuous_injection.py:203-320 (same ordering: theta fastest,
then r, then z; same np.random call sequence so seeded runs reproduce the reference)."""
import inspect
import warnings
import numpy as np


def _dens_func_args(dens_func):
    args = inspect.getfullargspec(dens_func).args
    if args and args[0] == 'self':
        args.pop(0)
    if args not in (['x', 'y', 'z'], ['z', 'r']):
        raise ValueError("The argument `dens_func` needs to be a function of z, r\n"
                         "or a function of x, y, z.")
    return args


def unalign_angles(thetap, Npz, Npr, method='irrational'):
    """Add the same angular offset to the Nptheta particles of each (z, r) position."""
    if method == 'random':
        shift = 2 * np.pi * np.random.rand(Npz, Npr)
    elif method == 'irrational':
        shift = 2 * np.pi * (np.sqrt(3) * np.arange(Npz)[:, np.newaxis]
                             + np.sqrt(2) * np.arange(Npr)[np.newaxis, :])
        shift = np.mod(shift, 2 * np.pi)
    else:
        raise ValueError("method must be either 'random' or 'irrational' but is %s" % method)
    thetap[:, :, :] = thetap[:, :, :] + shift[:, :, np.newaxis]


def generate_evenly_spaced(Npz, zmin, zmax, Npr, rmin, rmax, Nptheta, n, dens_func,
                           ux_m, uy_m, uz_m, ux_th, uy_th, uz_th):
    """Return (Ntot, x, y, z, ux, uy, uz, inv_gamma, w) for a regular (z, r, theta) lattice
    with weights n * r dtheta dr dz, optionally modulated by dens_func."""
    if Npz * Npr * Nptheta <= 0:
        e = np.empty(0)
        return 0, e, e.copy(), e.copy(), e.copy(), e.copy(), e.copy(), e.copy(), e.copy()
    dz = (zmax - zmin) * 1. / Npz
    z_reg = zmin + dz * (np.arange(Npz) + 0.5)
    dr = (rmax - rmin) * 1. / Npr
    r_reg = rmin + dr * (np.arange(Npr) + 0.5)
    dtheta = 2 * np.pi / Nptheta
    theta_reg = dtheta * np.arange(Nptheta)
    zp, rp, thetap = np.meshgrid(z_reg, r_reg, theta_reg, copy=True, indexing='ij')
    unalign_angles(thetap, Npz, Npr, method='random')
    r = rp.flatten()
    x = r * np.cos(thetap.flatten())
    y = r * np.sin(thetap.flatten())
    z = zp.flatten()
    w = n * r * dtheta * dr * dz
    if dens_func is not None:
        args = _dens_func_args(dens_func)
        if args == ['x', 'y', 'z']:
            w *= dens_func(x=x, y=y, z=z)
        else:
            w *= dens_func(z=z, r=r)
    selected = (w > 0)
    if np.any(w < 0):
        warnings.warn('The specified particle density returned negative densities.\n'
                      'No particles were generated in areas of negative density.')
    Ntot = int(selected.sum())
    x, y, z, w = x[selected], y[selected], z[selected], w[selected]
    uz = uz_m * np.ones(Ntot) + uz_th * np.random.normal(size=Ntot)
    ux = ux_m * np.ones(Ntot) + ux_th * np.random.normal(size=Ntot)
    uy = uy_m * np.ones(Ntot) + uy_th * np.random.normal(size=Ntot)
    inv_gamma = 1. / np.sqrt(1 + ux**2 + uy**2 + uz**2)
    return Ntot, x, y, z, ux, uy, uz, inv_gamma, w
