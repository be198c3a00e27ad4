"""Laser initialisation on the grid ("direct" method) -- one-off host-side setup that uses
the device transforms, SURVEY.md 8f row 2.

`GaussianLaser` restates the paraxial Gaussian pulse of
fbpic/lpa_utils/laser/laser_profiles.py:179-294 (longitudinal_laser_profiles.py:156-181,
transverse_laser_profiles.py:135-160), `LaguerreGaussLaser` / `DonutLikeLaguerreGaussLaser`
the Laguerre-Gauss pulses of :296-585 (what the reference's tests/test_laser.py uses to
exercise the modes m = 0 and m = 2); profiles add up with `+`; `add_laser_pulse(sim, profile)` restates
fbpic/lpa_utils/laser/direct_injection.py:12-217: the transverse field is evaluated on the
global grid and decomposed in azimuthal modes, Ez follows from div E = 0 and B from
d_t B = -curl E in spectral space, and the result is ADDED to the local grids.
Every rank builds the (small) global grid itself instead of gather/scatter through rank 0.
"""
import numpy as np
from scipy.constants import c, m_e, e
from ..fields import Fields


class LaserProfile(object):
    """Base class (laser_profiles.py:20-74): profiles can be summed with `+`."""

    def __init__(self, propagation_direction, gpu_capable=False):
        assert propagation_direction in [-1, 1]
        self.propag_direction = float(propagation_direction)
        self.gpu_capable = gpu_capable

    def E_field(self, x, y, z, t):
        raise NotImplementedError

    def __add__(self, other):
        return SummedLaserProfile(self, other)


class SummedLaserProfile(LaserProfile):
    """Sum of two profiles that propagate in the same direction (laser_profiles.py:77-102)."""

    def __init__(self, profile1, profile2):
        assert profile1.propag_direction == profile2.propag_direction
        LaserProfile.__init__(self, int(profile1.propag_direction))
        self.profile1, self.profile2 = profile1, profile2

    def E_field(self, x, y, z, t):
        Ex1, Ey1 = self.profile1.E_field(x, y, z, t)
        Ex2, Ey2 = self.profile2.E_field(x, y, z, t)
        return Ex1 + Ex2, Ey1 + Ey2


class GaussianLaser(LaserProfile):
    """Linearly polarised Gaussian pulse: a0, waist, duration tau, centroid z0, focal plane zf,
    polarisation angle, wavelength, carrier-envelope phase, chirp phi2."""

    def __init__(self, a0, waist, tau, z0, zf=None, theta_pol=0., lambda0=0.8e-6, cep_phase=0.,
                 phi2_chirp=0., propagation_direction=1):
        LaserProfile.__init__(self, propagation_direction)
        k0 = 2 * np.pi / lambda0
        E0 = a0 * m_e * c ** 2 * k0 / e
        self.E0x = E0 * np.cos(theta_pol)
        self.E0y = E0 * np.sin(theta_pol)
        self.k0 = k0
        self.z0 = z0
        self.zf = z0 if zf is None else zf
        self.cep_phase = cep_phase
        self.phi2_chirp = phi2_chirp
        self.inv_ctau2 = 1. / (c * tau) ** 2
        self.inv_zr = 1. / (0.5 * k0 * waist ** 2)
        self.w0 = waist

    def E_field(self, x, y, z, t):
        pd = self.propag_direction
        # longitudinal envelope x carrier (possibly chirped)
        stretch = 1 - 2j * self.phi2_chirp * c ** 2 * self.inv_ctau2
        arg = - 1j * self.cep_phase + 1j * self.k0 * (pd * (z - self.z0) - c * t) \
            - 1. / stretch * self.inv_ctau2 * (pd * (z - self.z0) - c * t) ** 2
        longi = np.exp(arg) / stretch ** 0.5
        # transverse Gaussian with diffraction (Gouy phase, curvature)
        diffract = 1. + 1j * pd * (z - self.zf) * self.inv_zr
        trans = np.exp(- (x ** 2 + y ** 2) / (self.w0 ** 2 * diffract)) / diffract
        profile = longi * trans
        return (self.E0x * profile).real, (self.E0y * profile).real


class _LaguerreGaussPulse(LaserProfile):
    """Paraxial Laguerre-Gauss mode (p, m) under an unchirped Gaussian temporal envelope:
    u = A (sqrt(2) r / w(z))^|m| L_p^|m|(2 r^2 / w(z)^2) exp(-r^2 / (w0^2 q)) / q
        x exp(-i (2p + |m|) psi) x azimuthal factor,   q = 1 + i (z - zf) / zR, psi = arg q,
    A = sqrt(p! / (|m| + p)!) (so that the pulse energy does not depend on p and m)."""

    def __init__(self, p, m, a0, waist, tau, z0, zf, theta_pol, lambda0, cep_phase,
                 propagation_direction):
        from scipy.special import genlaguerre, factorial
        LaserProfile.__init__(self, propagation_direction)
        k0 = 2 * np.pi / lambda0
        E0 = a0 * m_e * c ** 2 * k0 / e
        self.E0x, self.E0y = E0 * np.cos(theta_pol), E0 * np.sin(theta_pol)
        self.p, self.m, self.k0, self.z0 = p, m, k0, z0
        self.zf = z0 if zf is None else zf
        self.cep_phase = cep_phase
        self.inv_ctau2 = 1. / (c * tau) ** 2
        self.inv_zr = 1. / (0.5 * k0 * waist ** 2)
        self.w0 = waist
        self.amplitude = np.sqrt(factorial(p) / factorial(abs(m) + p))
        self.laguerre = genlaguerre(p, abs(m))

    def _azimuthal(self, theta):
        raise NotImplementedError

    def E_field(self, x, y, z, t):
        pd = self.propag_direction
        xi = pd * (z - self.z0) - c * t
        longi = np.exp(-1j * self.cep_phase + 1j * self.k0 * xi - self.inv_ctau2 * xi ** 2)
        q = 1. + 1j * pd * (z - self.zf) * self.inv_zr
        r2 = x ** 2 + y ** 2
        s2 = 2 * r2 / (self.w0 * abs(q)) ** 2
        am = abs(self.m)
        trans = np.exp(-r2 / (self.w0 ** 2 * q) - 1j * (2 * self.p + am) * np.angle(q)) / q \
            * np.sqrt(s2) ** am * self.laguerre(s2) * self._azimuthal(np.angle(x + 1j * y))
        profile = self.amplitude * longi * trans
        return (self.E0x * profile).real, (self.E0y * profile).real


class LaguerreGaussLaser(_LaguerreGaussPulse):
    """Linearly polarised Laguerre-Gauss pulse whose field varies as cos[m (theta - theta0)]
    (phase independent of theta; laser_profiles.py:296-446, transverse_laser_profiles.py:
    169-310).  Needs the azimuthal modes 0 .. m+1."""

    def __init__(self, p, m, a0, waist, tau, z0, zf=None, theta_pol=0., lambda0=0.8e-6,
                 cep_phase=0., theta0=0., propagation_direction=1):
        if m < 0 or type(m) is not int:
            raise ValueError("m should be an integer positive number.")
        _LaguerreGaussPulse.__init__(self, p, m, a0, waist, tau, z0, zf, theta_pol, lambda0,
                                     cep_phase, propagation_direction)
        if m != 0:
            self.amplitude *= 2 ** .5       # <cos^2> = 1/2: same energy as the m = 0 pulse
        self.theta0 = theta0

    def _azimuthal(self, theta):
        return np.cos(self.m * (theta - self.theta0))


class DonutLikeLaguerreGaussLaser(_LaguerreGaussPulse):
    """Linearly polarised Laguerre-Gauss pulse with the cork-screw phase exp(-i m theta) and a
    theta-independent (donut) intensity (laser_profiles.py:448-585,
    transverse_laser_profiles.py:312-432).  m may be negative; needs modes 0 .. |m|+1."""

    def __init__(self, p, m, a0, waist, tau, z0, zf=None, theta_pol=0., lambda0=0.8e-6,
                 cep_phase=0., propagation_direction=1):
        _LaguerreGaussPulse.__init__(self, p, m, a0, waist, tau, z0, zf, theta_pol, lambda0,
                                     cep_phase, propagation_direction)

    def _azimuthal(self, theta):
        return np.exp(-1j * self.m * theta)


def add_laser_pulse(sim, laser_profile, gamma_boost=None, method='direct', z0_antenna=None,
                    v_antenna=0.):
    if gamma_boost is not None and gamma_boost != 1.:
        raise NotImplementedError('boosted-frame laser initialisation is outside the fbpic_amd scope')
    if method != 'direct':
        raise NotImplementedError("only method='direct' is implemented (the laser antenna is "
                                  "outside the fbpic_amd scope)")
    add_laser_direct(sim, laser_profile)


def add_laser(sim, a0, w0, ctau, z0, zf=None, lambda0=0.8e-6, cep_phase=0., phi2_chirp=0.,
              theta_pol=0., gamma_boost=None, method='direct', fw_propagating=True,
              update_spectral=True, z0_antenna=None, v_antenna=0.):
    """Legacy signature of fbpic/lpa_utils/laser/laser.py:89-229."""
    prof = GaussianLaser(a0, w0, ctau / c, z0, zf=zf, theta_pol=theta_pol, lambda0=lambda0,
                         cep_phase=cep_phase, phi2_chirp=phi2_chirp,
                         propagation_direction=(1 if fw_propagating else -1))
    add_laser_pulse(sim, prof, gamma_boost=gamma_boost, method=method)


def _modes_of_transverse_field(z, r, Nm, laser_profile, time):
    """Er, Et on (z, r, theta) and their azimuthal DFT: arrays (Nz, Nr, 2*Nm)."""
    ntheta = 2 * Nm
    theta = (2 * np.pi / ntheta) * np.arange(ntheta)
    z3, r3, th3 = np.meshgrid(z, r, theta, indexing='ij')
    cth, sth = np.cos(th3), np.sin(th3)
    Ex, Ey = laser_profile.E_field(r3 * cth, r3 * sth, z3, time)
    Er = cth * Ex + sth * Ey
    Et = - sth * Ex + cth * Ey
    return np.fft.ifft(Er, axis=-1), np.fft.ifft(Et, axis=-1)


def add_laser_direct(sim, laser_profile, boost=None):
    comm, fld = sim.comm, sim.fld
    if fld.data_is_on_gpu:
        raise RuntimeError('add_laser_pulse must be called while the fields are on the host '
                           '(before Simulation.step / outside GpuMemoryManager)')
    Nm = fld.Nm
    # the global grid including the damp (+inject) cells, without guard cells
    gNz, giz = comm.get_Nz_and_iz(local=False, with_damp=True, with_guard=False)
    gzmin, gzmax = comm.get_zmin_zmax(local=False, with_damp=True, with_guard=False)
    g = Fields(gNz, gzmax, fld.Nr, fld.rmax, Nm, fld.dt, zmin=gzmin, n_order=fld.n_order)
    Er_m, Et_m = _modes_of_transverse_field(g.interp[0].z, g.interp[0].r, Nm, laser_profile, sim.time)
    for m in range(Nm):
        g.interp[m].Er = np.ascontiguousarray(Er_m[:, :, m])
        g.interp[m].Et = np.ascontiguousarray(Et_m[:, :, m])
    _calculate_laser_fields(g, laser_profile.propag_direction)
    # add the local part (local domain with damp cells, without guard cells)
    lNz, liz = comm.get_Nz_and_iz(local=True, with_damp=True, with_guard=False, rank=comm.rank)
    _, liz_arr = comm.get_Nz_and_iz(local=True, with_damp=True, with_guard=True, rank=comm.rank)
    i_loc = liz - liz_arr
    i_glob = liz - giz
    for m in range(Nm):
        for k in ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz'):
            getattr(fld.interp[m], k)[i_loc:i_loc + lNz, :] += getattr(g.interp[m], k)[i_glob:i_glob + lNz, :]


def _calculate_laser_fields(g, propag_direction):
    """direct_injection.py:148-217: E -> spectral space (device transforms), smooth the
    transverse components along z, Ez from div E = 0, B from the dispersion relation,
    then back to the interpolation grid."""
    g.send_fields_to_gpu()
    g.interp2spect('E')
    g.receive_fields_from_gpu()
    dz = g.interp[0].dz
    kz_true = 2 * np.pi * np.fft.fftfreq(g.Nz, dz)
    filt = (1. - np.sin(0.5 * kz_true * dz)**2) * (1. + np.sin(0.5 * kz_true * dz)**2)
    for m in range(g.Nm):
        sp = g.spect[m]
        sp.Ep *= filt[:, np.newaxis]
        sp.Em *= filt[:, np.newaxis]
        inv_kz = np.where(sp.kz == 0, 0, 1. / np.where(sp.kz == 0, 1., sp.kz))
        sp.Ez[:, :] = 1.j * sp.kr * (sp.Ep - sp.Em) * inv_kz
        w = c * np.sqrt(sp.kz**2 + sp.kr**2)
        w *= np.sign(sp.kz) * propag_direction
        inv_w = np.where(w == 0, 0., 1. / np.where(w == 0, 1., w))
        sp.Bp[:, :] = -1.j * inv_w * (sp.kz * sp.Ep - 0.5j * sp.kr * sp.Ez)
        sp.Bm[:, :] = -1.j * inv_w * (-sp.kz * sp.Em - 0.5j * sp.kr * sp.Ez)
        sp.Bz[:, :] = inv_w * sp.kr * (sp.Ep + sp.Em)
    g.send_fields_to_gpu()
    g.spect2interp('E')
    g.spect2interp('B')
    g.receive_fields_from_gpu()
