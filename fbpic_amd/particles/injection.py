"""Initial macroparticle lattice (host-side, NumPy): synthetic-input generator of every
configuration.  Restates generate_evenly_spaced / unalign_angles of
fbpic/particles/injection/continuous_injection.py:203-320 (same ordering: theta fastest,
then r, then z; same np.random call sequence so seeded runs reproduce the reference)."""
import inspect
import warnings
import numpy as np
from scipy.constants import c


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


class ContinuousInjector(object):
    """Book-keeping of the plasma that a moving window uncovers at its right edge
    (fbpic/particles/injection/continuous_injection.py:13-197): `z_inject` follows the
    window, `z_end_plasma` is the current end of the macroparticle lattice; whenever at
    least one particle spacing fits between them, `nz_inject` new lattice planes are due."""

    def __init__(self, Npz, zmin, zmax, dz_particles, Npr, rmin, rmax, Nptheta, n, dens_func,
                 ux_m, uy_m, uz_m, ux_th, uy_th, uz_th):
        self.Npr, self.rmin, self.rmax, self.Nptheta = Npr, rmin, rmax, Nptheta
        self.n, self.dens_func = n, dens_func
        self.ux_m, self.uy_m, self.uz_m = ux_m, uy_m, uz_m
        self.ux_th, self.uy_th, self.uz_th = ux_th, uy_th, uz_th
        self.dz_particles = (zmax - zmin) / Npz if Npz != 0 else dz_particles
        self.v_end_plasma = c * uz_m / np.sqrt(1 + ux_m**2 + uy_m**2 + uz_m**2)
        self.nz_inject = None
        self.z_inject = None
        self.z_end_plasma = None

    def initialize_injection_positions(self, comm, v_moving_window, species_z, dt):
        if comm.rank != comm.size - 1 or self.z_inject is not None:
            return
        _, zmax_with_damp = comm.get_zmin_zmax(local=False, with_damp=True, with_guard=False)
        self.z_inject = zmax_with_damp + (3 - comm.n_inject) * comm.dz \
            + comm.exchange_period * dt * (v_moving_window - self.v_end_plasma)
        self.nz_inject = 0
        if len(species_z) > 0:
            self.z_end_plasma = float(species_z.max()) + 0.5 * self.dz_particles
        else:
            _, zmax_phys = comm.get_zmin_zmax(local=False, with_damp=False, with_guard=False)
            self.z_end_plasma = zmax_phys
        if self.dz_particles is None:
            raise ValueError('The simulation uses continuous injection of particles, but was '
                             'unable to calculate the spacing between particles; pass '
                             '`dz_particles` when initializing the `Particles` object.')

    def reset_injection_positions(self):
        self.nz_inject = None
        self.z_inject = None
        self.z_end_plasma = None

    def increment_injection_positions(self, v_moving_window, duration):
        self.z_inject += v_moving_window * duration
        self.z_end_plasma += self.v_end_plasma * duration
        nz_new = int((self.z_inject - self.z_end_plasma) / self.dz_particles)
        self.nz_inject += nz_new
        self.z_end_plasma += nz_new * self.dz_particles

    def generate_particles(self, time):
        """New lattice planes between z_end_plasma - nz_inject*dz and z_end_plasma."""
        if self.dens_func is not None:
            args = _dens_func_args(self.dens_func)
            if args == ['z', 'r']:
                def dens_func(z, r):
                    return self.dens_func(z - self.v_end_plasma * time, r)
            else:
                def dens_func(x, y, z):
                    return self.dens_func(x, y, z - self.v_end_plasma * time)
        else:
            dens_func = None
        zmax = self.z_end_plasma
        zmin = self.z_end_plasma - self.nz_inject * self.dz_particles
        out = generate_evenly_spaced(self.nz_inject, zmin, zmax, self.Npr, self.rmin, self.rmax,
                                     self.Nptheta, self.n, dens_func, self.ux_m, self.uy_m,
                                     self.uz_m, self.ux_th, self.uy_th, self.uz_th)
        self.nz_inject = 0
        return out
