"""Host-side (NumPy) setup tables of fbpic_amd against golden vectors from the reference
(tests/golden/grid_setup.npz): Hankel matrices, cell volumes, Ruyten coefficients,
modified kz, stencil reach, PSATD coefficients, filters, inv_k2.  CPU-only."""
import numpy as np
import pytest
from scipy.constants import c
from conftest import golden, rel_err

from fbpic_amd.fields.spectral_transform.hankel import hankel_matrices, DHT
from fbpic_amd.fields.interpolation_grid import InterpolationGrid
from fbpic_amd.fields.utility_methods import get_modified_k, get_stencil_reach
from fbpic_amd.fields import Fields


def test_hankel_matrices():
    g = golden('grid_setup')
    for Nr in (16, 32):
        rmax = Nr * 0.5e-6
        for m in range(4):
            for p in (m - 1, m, m + 1):
                M, invM, nu, r = hankel_matrices(p, m, Nr, rmax)
                tag = 'Nr%d_m%d_p%d' % (Nr, m, p - m + 1)
                assert np.array_equal(nu, g['nu_' + tag])
                assert rel_err(invM, g['invM_' + tag]) < 1e-14
                assert rel_err(M, g['M_' + tag]) < 1e-12
    d = DHT(1, 0, 32, 8, 16e-6)
    # a DHT object must be a left inverse on the band-limited space
    assert np.abs(d.invM @ d.M - np.eye(32)).max() < 1e-12 * np.abs(d.invM).max() * np.abs(d.M).max()


def test_volumes_and_ruyten():
    g = golden('grid_setup')
    for Nr in (16, 128):
        for m in (0, 1):
            gr = InterpolationGrid(8, Nr, m, 0., 8 * 0.2e-6, Nr * 0.2e-6)
            assert rel_err(gr.invvol, g['invvol_Nr%d_m%d' % (Nr, m)]) < 1e-13
            assert rel_err(gr.ruyten_linear_coef, g['ruyl_Nr%d_m%d' % (Nr, m)]) < 1e-11
            assert rel_err(gr.ruyten_cubic_coef, g['ruyc_Nr%d_m%d' % (Nr, m)]) < 1e-11


def test_modified_k_and_stencil_reach():
    g = golden('grid_setup')
    kz = 2 * np.pi * np.fft.fftfreq(64, 0.1e-6)
    for n_order in (8, 16, 32):
        assert np.array_equal(get_modified_k(kz, n_order, 0.1e-6), g['kzmod_%d' % n_order])
        assert get_stencil_reach(1024, 0.2e-6, 0.2e-6, n_order, None, False) == int(g['reach_%d' % n_order])
    # n_guard = reach + 1 : 32 / 45 / 63 for n_order 8 / 16 / 32
    assert [int(g['reach_%d' % o]) + 1 for o in (8, 16, 32)] == [32, 45, 63]


def test_psatd_and_spectral_tables():
    g = golden('grid_setup')
    Nz, Nr, Nm = 32, 16, 3
    dt = 0.25e-6 / c
    for n_order in (-1, 16):
        f = Fields(Nz, Nz * 0.25e-6, Nr, Nr * 0.5e-6, Nm, dt, n_order=n_order, zmin=0.)
        for m in range(Nm):
            t = 'o%d_m%d' % (n_order, m)
            for k in ('C', 'S_w', 'j_coef', 'rho_prev_coef', 'rho_next_coef'):
                assert rel_err(getattr(f.psatd[m], k), g[k + '_' + t]) < 1e-14, (k, t)
            assert np.array_equal(f.spect[m].kz[:, 0], g['kz_' + t])
            assert np.array_equal(f.spect[m].kr[0, :], g['kr_' + t])
            assert rel_err(f.spect[m].inv_k2, g['inv_k2_' + t]) < 1e-15
            assert np.array_equal(f.spect[m].filter_array_z, g['filter_z_' + t])
            assert np.array_equal(f.spect[m].filter_array_r, g['filter_r_' + t])


def test_slab_indexing_is_a_bijection():
    f = Fields(8, 8e-6, 4, 4e-6, 3, 1e-15)
    from fbpic_amd.fields.interpolation_grid import INTERP_FIELDS
    from fbpic_amd.fields.spectral_grid import SPECT_FIELDS
    ii = sorted(f.interp_index(k, m) for m in range(3) for k in INTERP_FIELDS)
    ss = sorted(f.spect_index(k, m) for m in range(3) for k in SPECT_FIELDS)
    assert ii == list(range(f.NFi)) and ss == list(range(f.NFs))
    # a vector group occupies the same slots in both slabs (batched FFT maps 1:1)
    for m in range(3):
        for a, b in (('Er', 'Ep'), ('Et', 'Em'), ('Ez', 'Ez'), ('Jr', 'Jp'), ('Bt', 'Bm')):
            assert f.interp_index(a, m) == f.spect_index(b, m)


def test_slab_indexing_with_cross_deposition():
    """current_correction='cross-deposition' adds rho_next_z / rho_next_xy of every mode to the
    spectral slab (spectral_grid.py:97-99); the first 11 Nm slots keep their meaning."""
    from fbpic_amd.fields.spectral_grid import SPECT_FIELDS, CROSS_FIELDS
    Nm = 2
    f = Fields(8, 8e-6, 4, 4e-6, Nm, 1e-15, current_correction='cross-deposition')
    g = Fields(8, 8e-6, 4, 4e-6, Nm, 1e-15)
    assert f.NFs == 13 * Nm and g.NFs == 11 * Nm
    names = f.spect[0].field_names
    assert names == SPECT_FIELDS + CROSS_FIELDS and g.spect[0].field_names == SPECT_FIELDS
    ss = sorted(f.spect_index(k, m) for m in range(Nm) for k in names)
    assert ss == list(range(f.NFs))
    for m in range(Nm):
        for k in SPECT_FIELDS:
            assert f.spect_index(k, m) == g.spect_index(k, m)
        assert f.spect[m].rho_next_z.shape == (8, 4) and not hasattr(g.spect[m], 'rho_next_z')
    # transform groups: both extra densities come from the interpolation-grid rho
    assert f._group('rho_next_z') == (9 * Nm, 11 * Nm, Nm, False)
    assert f._group('rho_next_xy') == (9 * Nm, 12 * Nm, Nm, False)
    with pytest.raises(ValueError):
        g._group('rho_next_z')
    with pytest.raises(ValueError):
        Fields(8, 8e-6, 4, 4e-6, Nm, 1e-15, current_correction='nonsense')
