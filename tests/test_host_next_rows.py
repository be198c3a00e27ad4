"""CPU-only checks of the host-side pieces of the "next" rows: Gaussian laser
profile against its closed form at focus, continuous-injection book-keeping, damping profile.
(The full open-boundary / moving-window cycle is checked on the GPU against the reference
trajectory in tests/test_gpu_lwfa.py.)"""
import numpy as np
import pytest
from scipy.constants import c, m_e, e


def test_gaussian_laser_closed_form_at_focus():
    from fbpic_amd.lpa_utils.laser import GaussianLaser
    a0, w0, tau, z0, lam, cep = 2., 5e-6, 20e-15, 3e-6, 0.8e-6, 0.3
    prof = GaussianLaser(a0, w0, tau, z0, lambda0=lam, cep_phase=cep, theta_pol=0.)
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=50) * 3e-6, rng.normal(size=50) * 3e-6
    t = 7e-15
    z = np.full(50, z0)                       # focal plane (zf = z0): no diffraction terms
    Ex, Ey = prof.E_field(x, y, z, t)
    k0 = 2 * np.pi / lam
    E0 = a0 * m_e * c**2 * k0 / e
    ref = E0 * np.exp(-(x**2 + y**2) / w0**2 - (z - z0 - c * t)**2 / (c * tau)**2) \
        * np.cos(k0 * (z - z0 - c * t) - cep)
    assert np.allclose(Ex, ref, rtol=1e-12, atol=1e-6 * E0)
    assert np.all(Ey == 0.)
    # polarisation angle splits the same profile between x and y
    Ex2, Ey2 = GaussianLaser(a0, w0, tau, z0, lambda0=lam, cep_phase=cep, theta_pol=0.7).E_field(x, y, z, t)
    assert np.allclose(Ex2, np.cos(0.7) * ref, rtol=1e-12, atol=1e-6 * E0)
    assert np.allclose(Ey2, np.sin(0.7) * ref, rtol=1e-12, atol=1e-6 * E0)


def test_continuous_injector_bookkeeping():
    from fbpic_amd.particles.injection import ContinuousInjector
    np.random.seed(3)
    inj = ContinuousInjector(Npz=10, zmin=0., zmax=10e-6, dz_particles=None, Npr=4, rmin=0.,
                             rmax=4e-6, Nptheta=4, n=1e24, dens_func=None,
                             ux_m=0., uy_m=0., uz_m=0., ux_th=0., uy_th=0., uz_th=0.)
    assert abs(inj.dz_particles - 1e-6) < 1e-20 and inj.v_end_plasma == 0.
    inj.z_inject, inj.z_end_plasma, inj.nz_inject = 10e-6, 10e-6, 0
    total = 0
    for _ in range(7):
        inj.increment_injection_positions(c, 0.8e-6 / c)      # the window advances 0.8 um
        n, x, y, z, ux, uy, uz, ig, w = inj.generate_particles(0.)
        total += n
        assert n % (4 * 4) == 0 and inj.nz_inject == 0
        if n:
            assert z.max() < inj.z_end_plasma and z.min() > inj.z_end_plasma - (n // 16 + 1) * 1e-6
    # 7 * 0.8 um = 5.6 um uncovered -> 5 lattice planes of 16 particles
    assert total == 5 * 16
    assert abs(inj.z_end_plasma - 15e-6) < 1e-15


def test_damp_profile_matches_reference_formula():
    from fbpic_amd.boundaries.boundary_communicator import BoundaryCommunicator
    comm = BoundaryCommunicator(64, 0., 64e-6, 8, 8e-6, 2, 1e-6 / c, None, False,
                                {'z': 'open', 'r': 'reflective'}, -1, 16, {'z': 16, 'r': 8},
                                1., None, 4, False)
    d = comm.left_damp
    assert d.shape == (16 + 16 + 8,)
    i = np.arange(40)
    ref = np.where(i < 24 + 8, np.sin((i - 24) * np.pi / 16.)**2, 1.)
    ref = np.where(i < 24, 0., ref)
    assert np.array_equal(d, ref)


def test_laser_profiles_vs_reference():
    """Gaussian, Laguerre-Gauss and donut-like Laguerre-Gauss pulses and the sum of two
    profiles against the reference's E_field at 2000 random points
    (tests/golden/laser_profiles.npz, oracle/capture_golden.py:cap_laser_profiles)."""
    from conftest import golden
    from fbpic_amd.lpa_utils.laser import GaussianLaser, LaguerreGaussLaser, \
        DonutLikeLaguerreGaussLaser
    g = golden('laser_profiles')
    x, y, z, t = g['x'], g['y'], g['z'], float(g['t'])
    cases = [('lg', (0, 1), dict()),
             ('lg', (1, 2), dict(theta0=0.3, theta_pol=0.7)),
             ('lg', (2, 0), dict(cep_phase=0.4)),
             ('lg', (0, 3), dict(propagation_direction=-1)),
             ('donut', (0, -1), dict()),
             ('donut', (1, 2), dict(theta_pol=0.7)),
             ('donut', (2, 0), dict(cep_phase=0.4)),
             ('donut', (0, -3), dict(propagation_direction=-1)),
             ('gauss', (), dict(theta_pol=0.2, cep_phase=0.1))]
    cls = dict(lg=LaguerreGaussLaser, donut=DonutLikeLaguerreGaussLaser, gauss=GaussianLaser)
    for i, (kind, pm, kw) in enumerate(cases):
        got = np.array(cls[kind](*pm, 1.3, 4e-6, 8e-15, 1e-6, zf=12e-6, **kw).E_field(x, y, z, t))
        ref = g['case%d' % i]
        assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (i, kind, pm)
    s = LaguerreGaussLaser(0, 1, 0.5, 4e-6, 8e-15, 0., zf=5e-6, theta_pol=0., theta0=0.) \
        + LaguerreGaussLaser(0, 1, 0.5, 4e-6, 8e-15, 0., zf=5e-6, theta_pol=np.pi / 2,
                             theta0=np.pi / 2)
    got = np.array(s.E_field(x, y, z, t))
    assert np.abs(got - g['sum']).max() <= 1e-13 * np.abs(g['sum']).max()
    with pytest.raises(ValueError):
        LaguerreGaussLaser(0, -1, 1., 4e-6, 8e-15, 0.)


def test_external_field_hook_on_host_arrays():
    """ExternalField.apply_expression (lpa_utils/external_fields.py:174-213) with a `math`-style
    scalar function and with a NumPy-aware one; species filter; argument checks."""
    import math
    from fbpic_amd.lpa_utils.external_fields import ExternalField

    class Sp:
        pass
    rng = np.random.default_rng(5)
    sp, other = Sp(), Sp()
    for s in (sp, other):
        s.Ntot = 50
        for k in ('x', 'y', 'z', 'Ex', 'By'):
            setattr(s, k, rng.normal(size=50))
    ex0, by0, z = sp.Ex.copy(), sp.By.copy(), sp.z.copy()
    o0 = other.Ex.copy()

    def f_scalar(F, x, y, z, t, amplitude, length_scale):
        return F + amplitude * math.cos(2 * np.pi * (z - c * t) / length_scale)

    def f_array(F, x, y, z, t, amplitude, length_scale):
        return F + amplitude * np.cos(2 * np.pi * (z - c * t) / length_scale)
    t = 1e-15
    ExternalField(f_scalar, 'Ex', 3., 0.8e-6, species=sp).apply_expression([sp, other], t)
    ExternalField(f_array, 'By', 2., 0.8e-6).apply_expression([sp], t)
    wave = np.cos(2 * np.pi * (z - c * t) / 0.8e-6)
    assert np.allclose(sp.Ex, ex0 + 3. * wave, rtol=0, atol=1e-14)
    assert np.allclose(sp.By, by0 + 2. * wave, rtol=0, atol=1e-14)
    assert np.array_equal(other.Ex, o0)                 # filtered out by `species`
    with pytest.raises(ValueError):
        ExternalField(f_scalar, 'Er', 1., 1.)
    with pytest.raises(NotImplementedError):
        ExternalField(f_scalar, 'Ex', 1., 1., gamma_boost=10.)
