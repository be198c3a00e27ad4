"""Pins the oracle's WHOLE-CYCLE restatement (oracle.OracleSim: order of operations of
Simulation.step / deposit / exchange_and_damp_EB) and fbpic_amd's host-side setup tables
against trajectories captured from the real reference (tests/golden/cycle_*.npz,
bunch_*.npz).  CPU-only: the product's HIP path is not involved."""
import numpy as np
import pytest
from conftest import golden
import helpers
from helpers import INTERP, SPECT, PTCL


def _check(o, g, tag, tol):
    Nm = o.Nm
    for m in range(Nm):
        for i, k in enumerate(INTERP):
            grp = [j for j, kk in enumerate(INTERP) if kk[0] == k[0]]
            scale = np.abs(g[tag + '_interp'][:, grp]).max()
            if scale > 0:
                err = np.abs(o.interp[m][k] - g[tag + '_interp'][m, i]).max() / scale
                assert err < tol, (tag, m, k, err)
        for i, k in enumerate(SPECT):
            grp = [j for j, kk in enumerate(SPECT) if kk[0] == k[0]]
            scale = np.abs(g[tag + '_spect'][:, grp]).max()
            if scale > 0:
                err = np.abs(o.spect[m][k] - g[tag + '_spect'][m, i]).max() / scale
                assert err < tol, (tag, 'spect', m, k, err)
    for isp, s in enumerate(o.species):
        ref = g['%s_ptcl%d' % (tag, isp)]
        for j, k in enumerate(PTCL):
            sc = np.abs(ref[j]).max()
            if sc > 0:
                assert np.abs(s[k] - ref[j]).max() / sc < tol, (tag, isp, k)


@pytest.mark.parametrize('name', ['cycle_lin_16x8_nm2', 'cycle_cub_16x8_nm2',
                                  'cycle_lin_32x16_nm3', 'cycle_cub_32x16_nm2_ions'])
def test_oracle_cycle_vs_reference(oracle, name):
    g = golden(name)
    sim = helpers.build_from_golden(g, name)
    o = oracle.from_sim(sim, nthreads=1)
    utr = bool(g['use_true_rho'])
    done = 0
    for upto, tol in ((1, 1e-13), (2, 5e-13), (5, 5e-12)):
        o.step(upto - done, use_true_rho=utr)
        done = upto
        _check(o, g, 's%d' % upto, tol)


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_oracle_bunch_vs_reference(oracle, shape):
    g = golden('bunch_' + shape)
    sim = helpers.build_from_golden(g, 'bunch_' + shape)
    o = oracle.from_sim(sim, nthreads=1)
    for it in (1, 2, 3):
        o.step(1)
        ref = g['s%d_JrJtJzrho' % it]
        for m in range(o.Nm):
            for i, k in enumerate(('Jr', 'Jt', 'Jz', 'rho')):
                grp = [0, 1, 2] if i < 3 else [3]
                tol = 1.e-13 * 2 * np.abs(ref[:, grp]).max()
                assert np.abs(o.interp[m][k] - ref[m, i]).max() <= tol, (it, m, k)


GALILEAN = ['cycle_galilean_cub_16x8', 'cycle_comoving_lin_16x8', 'cycle_galilean_lin_32x8_o8']


@pytest.mark.parametrize('name', GALILEAN)
def test_oracle_galilean_cycle_vs_reference(oracle, name):
    """Galilean / comoving-current PSATD: drifting periodic plasma, grid
    following it (Galilean) or comoving currents, infinite and finite stencil order."""
    g = golden(name)
    sim = helpers.build_from_golden(g, name)
    assert sim.use_galilean == bool(g['use_galilean']) and sim.v_comoving == float(g['v_comoving'])
    o = oracle.from_sim(sim, nthreads=1)
    done = 0
    for upto, tol in ((1, 1e-13), (2, 5e-13), (5, 5e-12)):
        o.step(upto - done)
        done = upto
        _check(o, g, 's%d' % upto, tol)
        assert abs(o.zmin - float(g['s%d_zmin' % upto])) <= 1e-15 * abs(float(g['zmax']))


CROSS = ['cycle_cross_lin_16x8', 'cycle_cross_cub_16x8', 'cycle_cross_galilean_cub_16x8']


@pytest.mark.parametrize('name', CROSS)
def test_oracle_crossdeposition_cycle_vs_reference(oracle, name):
    """current_correction='cross-deposition' (main.py:512-514, 672-716):
    plasma wave with the standard PSATD (linear, cubic) and a Galilean drifting plasma."""
    g = golden(name)
    sim = helpers.build_from_golden(g, name)
    assert sim.fld.current_correction == 'cross-deposition'
    o = oracle.from_sim(sim, nthreads=1)
    done = 0
    # (the correction divides by kz and kr: round-off a few times that of the curl-free runs)
    for upto, tol in ((1, 5e-13), (2, 2e-12), (5, 2e-11)):
        o.step(upto - done)
        done = upto
        _check(o, g, 's%d' % upto, tol)
        assert abs(o.zmin - float(g['s%d_zmin' % upto])) <= 1e-15 * abs(float(g['zmax']))
