"""Whole PIC cycle on the MI355X (Simulation.step through the C ABI) against
  (1) golden trajectories captured from the real reference (tests/golden/cycle_*.npz,
      bunch_*.npz = counterpart of the reference's tests/test_cpu_gpu_deposition.py),
  (2) the CPU oracle stepping the same inputs (oracle.OracleSim),
  (3) the reference's own physics assertions of tests/test_periodic_plasma_wave.py
      (div E - rho/eps0 < 1e-11 in spectral space; E vs linear theory atol 1.1e6 rtol 2e-2),
  (4) size-independent properties at the headline size (C2: 1024x128, Nm=2, 32 ppc).
Tolerance for (1),(2): 1e-13 * max|F| per deposition (the reference's own CPU<->GPU bound,
kernel tests); 5e-13 after one full step (deposit + 3 transforms + solver, each adding its
own summation-order rounding), 2e-12 / 2e-11 after 2 / 5 steps for fields and particles
(rounding differences are amplified by the PIC loop).
"""
import numpy as np
import pytest
from scipy.constants import c, e, m_e, epsilon_0
from conftest import golden, rel_err, achieved
import helpers
from helpers import PTCL, INTERP, SPECT

pytestmark = pytest.mark.gpu


build_from_golden = helpers.build_from_golden


def compare_state(sim, g, tag, tol_f, tol_p, spect=True):
    Nm = sim.fld.Nm
    for m in range(Nm):
        for i, k in enumerate(INTERP):
            ref = g[tag + '_interp'][m, i]
            # compare each field with the max of its group (E, B, J, rho)
            grp = [j for j, kk in enumerate(INTERP) if kk[0] == k[0]]
            scale = np.abs(g[tag + '_interp'][:, grp]).max()
            if scale == 0:
                continue
            err = np.abs(getattr(sim.fld.interp[m], k) - ref).max() / scale
            achieved(None, err, tol_f, 'interp ' + tag)
        if spect:
            for i, k in enumerate(SPECT):
                grp = [j for j, kk in enumerate(SPECT) if kk[0] == k[0]]
                scale = np.abs(g[tag + '_spect'][:, grp]).max()
                if scale == 0:
                    continue
                err = np.abs(getattr(sim.fld.spect[m], k) - g[tag + '_spect'][m, i]).max() / scale
                achieved(None, err, tol_f, 'spect ' + tag)
    for isp, s in enumerate(sim.ptcl):
        ref = g['%s_ptcl%d' % (tag, isp)]
        # the GPU path sorts particles: compare as sets keyed by (w, then position)
        got = np.array([getattr(s, k) for k in PTCL])
        o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
        o2 = np.lexsort((got[2], got[1], got[0], got[7]))
        for j, k in enumerate(PTCL[:8]):
            sc = np.abs(ref[j]).max()
            if sc == 0:
                continue
            err = np.abs(got[j][o2] - ref[j][o1]).max() / sc
            achieved(None, err, tol_p, 'particles ' + tag)


@pytest.mark.parametrize('name', ['cycle_lin_16x8_nm2', 'cycle_cub_16x8_nm2',
                                  'cycle_lin_32x16_nm3', 'cycle_cub_32x16_nm2_ions'])
def test_cycle_vs_reference_golden(name):
    g = golden(name)
    sim = build_from_golden(g, name)
    utr = bool(g['use_true_rho'])
    done = 0
    # bounds = 10 x the worst deviation measured on MI355X (profiles/r03_achieved_errors.json):
    # fields 3.1e-13 / 1.5e-13 / 6.2e-14 after 1 / 2 / 5 steps, particles 5.7e-14
    for upto, tol in ((1, 5e-13), (2, 1.5e-12), (5, 1e-12)):
        sim.step(upto - done, use_true_rho=utr)
        done = upto
        compare_state(sim, g, 's%d' % upto, tol, 5e-13)


@pytest.mark.parametrize('name', ['cycle_galilean_cub_16x8', 'cycle_comoving_lin_16x8',
                                  'cycle_galilean_lin_32x8_o8'])
def test_galilean_cycle_vs_reference_golden(name):
    """Galilean / comoving-current PSATD against the reference's
    trajectory of a plasma drifting at gamma = 3: all grids, all particle arrays, and the
    position of the (Galilean) grid after 1, 2 and 5 steps."""
    g = golden(name)
    sim = build_from_golden(g, name)
    done = 0
    # (measured: <= 1.6e-15 on fields, <= 7e-16 on particles at every snapshot)
    for upto, tol in ((1, 5e-14), (2, 5e-14), (5, 5e-14)):
        sim.step(upto - done)
        done = upto
        compare_state(sim, g, 's%d' % upto, tol, 2e-14)
        assert abs(sim.fld.interp[0].zmin - float(g['s%d_zmin' % upto])) <= 1e-15 * float(g['zmax'])


@pytest.mark.parametrize('name', ['cycle_cross_lin_16x8', 'cycle_cross_cub_16x8',
                                  'cycle_cross_galilean_cub_16x8'])
def test_crossdeposition_cycle_vs_reference_golden(name):
    """current_correction='cross-deposition' (main.py:512-514, 672-716)
    against the reference's trajectory: plasma wave with the standard PSATD (linear, cubic)
    and a Galilean drifting plasma (cubic).  The correction takes the difference of four
    nearly equal charge densities and divides by kz, kr: the (order-dependent) round-off of
    the deposition sums is amplified ~20x compared with the curl-free runs (CPU oracle, same
    summation order as the reference: 1e-13; here, atomics in arbitrary order: 2e-12)."""
    g = golden(name)
    sim = build_from_golden(g, name)
    assert sim.fld.current_correction == 'cross-deposition'
    done = 0
    # (measured: spectral fields 2.1e-12 / 1.0e-12 / 4.5e-13 after 1 / 2 / 5 steps, particles 7e-16)
    for upto, tol in ((1, 1e-11), (2, 1e-11), (5, 5e-12)):
        sim.step(upto - done)
        done = upto
        compare_state(sim, g, 's%d' % upto, tol, 2e-14)
        assert abs(sim.fld.interp[0].zmin - float(g['s%d_zmin' % upto])) <= 1e-15 * float(g['zmax'])


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_bunch_deposition_vs_reference_golden(shape):
    """Counterpart of tests/test_cpu_gpu_deposition.py: rho and J of a Gaussian bunch
    (np.random.seed(0), N=2000) over 3 steps, atol = 1e-13*(max|F_cpu| + max|F_gpu|)."""
    g = golden('bunch_' + shape)
    sim = build_from_golden(g, 'bunch_' + shape)
    for it in (1, 2, 3):
        sim.step(1)
        ref = g['s%d_JrJtJzrho' % it]
        for m in range(sim.fld.Nm):
            for i, k in enumerate(('Jr', 'Jt', 'Jz', 'rho')):
                F = getattr(sim.fld.interp[m], k)
                grp = [0, 1, 2] if i < 3 else [3]
                scale = np.abs(ref[:, grp]).max() + np.abs(F).max()
                achieved(None, np.abs(F - ref[m, i]).max() / scale, 1.e-14, 'step %d' % it)      # measured 5.7e-16; the reference's own bound is 1e-13


@pytest.mark.parametrize('shape,Nm', [('linear', 2), ('cubic', 2), ('linear', 4), ('cubic', 3)])
def test_cycle_vs_oracle_medium(oracle, shape, Nm):
    """Same seeded uniform-plasma input stepped by the HIP path and by the CPU oracle."""
    ppc = (2, 2, 4 * Nm)
    sim = helpers.uniform_plasma_sim(64, 32, Nm, ppc, shape, seed=3, u_th=0.05)
    orc = helpers.oracle_from_sim(oracle, sim, nthreads=2)
    sim.step(3)
    orc.step(3)
    for m in range(Nm):
        for k in INTERP:
            grp = [kk for kk in INTERP if kk[0] == k[0]]
            scale = max(np.abs(orc.interp[mm][kk]).max() for mm in range(Nm) for kk in grp)
            if scale == 0:
                continue
            err = np.abs(getattr(sim.fld.interp[m], k) - orc.interp[m][k]).max() / scale
            achieved(None, err, 5e-12, 'fields')          # measured <= 6.7e-13
    s, o = sim.ptcl[0], orc.species[0]
    got = np.array([getattr(s, k) for k in PTCL[:8]])
    ref = np.array([o[k] for k in PTCL[:8]])
    o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
    o2 = np.lexsort((got[2], got[1], got[0], got[7]))
    for j, k in enumerate(PTCL[:8]):
        achieved(None, np.abs(got[j][o2] - ref[j][o1]).max() / np.abs(ref[j]).max(), 1e-14, 'particles')      # 5.3e-16


def _plasma_wave(shape, Nz=64, Nr=64, Nm=2, ppc=(2, 2, 8), n_periods=1, n_order=-1):
    """tests/test_periodic_plasma_wave.py at the reference's own dz, dt, ppc (box = one
    plasma wavelength; Nm=2 with eps_2 = 0 when Nm == 2)."""
    import importlib
    from fbpic_amd.main import Simulation, GpuMemoryManager
    dz = 0.2e-6
    zmax = Nz * dz
    rmax = 20.e-6
    dt = dz / c
    n_e = 2.e24
    eps = [0.001, 0.001, 0.001 if Nm > 2 else 0.]
    w0 = 5.e-6
    k0 = 2 * np.pi / zmax * n_periods
    wp = np.sqrt(n_e * e**2 / (m_e * epsilon_0))
    N_step = int(2 * np.pi / (wp * dt) * 0.75)
    np.random.seed(0)
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, 0., zmax + dz, 0., 18.e-6, ppc[0], ppc[1], ppc[2],
                     n_e, n_order=n_order, particle_shape=shape)
    with GpuMemoryManager(sim):
        sim.deposit('rho_prev', exchange=True)
        sim.fld.spect2interp('rho_prev')
    rho_ions = [-sim.fld.interp[m].rho.copy() for m in range(Nm)]
    s = sim.ptcl[0]
    x, y, z = s.x, s.y, s.z
    r = np.sqrt(x**2 + y**2)
    ex = np.exp(-r**2 / w0**2)
    # analytic momenta of the reference test (test_periodic_plasma_wave.py:224-283), t = 0
    s.ux = (eps[0] * c / wp * 2 * x / w0**2 - eps[1] * c / wp * 2 / w0 + eps[1] * c / wp * 4 * x**2 / w0**3
            - eps[2] * c / wp * 8 * x / w0**2 + eps[2] * c / wp * 8 * x * (x**2 - y**2) / w0**4) * ex * np.sin(k0 * z)
    s.uy = (eps[0] * c / wp * 2 * y / w0**2 + eps[1] * c / wp * 4 * x * y / w0**3
            + eps[2] * c / wp * 8 * y / w0**2 + eps[2] * c / wp * 8 * y * (x**2 - y**2) / w0**4) * ex * np.sin(k0 * z)
    s.uz = (-eps[0] * c / wp * k0 - eps[1] * c / wp * k0 * 2 * x / w0
            - eps[2] * c / wp * k0 * 4 * (x**2 - y**2) / w0**2) * ex * np.cos(k0 * z)
    s.inv_gamma = 1. / np.sqrt(1 + s.ux**2 + s.uy**2 + s.uz**2)
    sim.step(N_step, correct_currents=True)
    return sim, rho_ions, (eps, k0, w0, wp)


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_periodic_plasma_wave_reference_assertions(shape):
    from fbpic_amd.main import GpuMemoryManager
    sim, rho_ions, (eps, k0, w0, wp) = _plasma_wave(shape)
    fld = sim.fld
    Nm = fld.Nm
    t = sim.time
    g0 = fld.interp[0]
    rr, zz = np.meshgrid(g0.r, g0.z)
    pref = m_e * c**2 / e
    ex = np.exp(-rr**2 / w0**2)
    Ez_th = (-eps[0] * k0 - eps[1] * k0 * 2 * rr / w0 - eps[2] * k0 * 4 * rr**2 / w0**2) * pref * ex * np.cos(k0 * zz) * np.sin(wp * t)
    Er_th = (eps[0] * 2 * rr / w0**2 - eps[1] * 2 / w0 + eps[1] * 4 * rr**2 / w0**3
             - eps[2] * 8 * rr / w0**2 + eps[2] * 8 * rr**3 / w0**4) * pref * ex * np.sin(k0 * zz) * np.sin(wp * t)
    Ez_sim = fld.interp[0].Ez.real + sum(2 * fld.interp[m].Ez.real for m in range(1, Nm))
    Er_sim = fld.interp[0].Er.real + sum(2 * fld.interp[m].Er.real for m in range(1, Nm))
    assert np.allclose(Ez_th, Ez_sim, atol=1.1e6, rtol=2e-2)     # reference :407-409
    assert np.allclose(Er_th, Er_sim, atol=1.1e6, rtol=2e-2)
    # charge conservation in spectral space, reference :313-362
    for m in range(Nm):
        fld.interp[m].rho = fld.interp[m].rho + rho_ions[m]
    with GpuMemoryManager(sim):
        fld.interp2spect('E')
        fld.interp2spect('rho_prev')
    for m in range(Nm):
        sp = fld.spect[m]
        divE = sp.kr * (sp.Ep - sp.Em) + 1.j * sp.kz * sp.Ez
        rho_eps0 = sp.rho_prev / epsilon_0
        rel = np.sqrt(np.sum(abs(divE - rho_eps0)**2) / np.sum(abs(rho_eps0)**2))
        achieved(None, rel, 1.e-11, 'divE - rho/eps0')


def test_headline_size_properties():
    """C2 (1024 x 128, Nm = 2, 32 ppc, linear): properties that need no oracle run."""
    from fbpic_amd.main import GpuMemoryManager
    sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
    s = sim.ptcl[0]
    s.keep_sort_outputs = True          # materialise cell_idx / sorted_idx for check (a)
    n = s.Ntot
    assert n == 4194304
    w_sorted = np.sort(s.w)
    q_tot = s.q * s.w.sum()
    with GpuMemoryManager(sim):
        sim.step(3)
        # (a) sortedness + prefix sum consistency after the last deposit
        ci = s.cell_idx.cpu().numpy()
        pre = s.prefix_sum.cpu().numpy()
        assert np.all(np.diff(ci) >= 0) and pre[-1] == n
        assert np.array_equal(np.cumsum(np.bincount(ci, minlength=pre.size)), pre)
        # (b) the sort is a permutation: weights are conserved as a multiset
        assert np.array_equal(np.sort(s.w.cpu().numpy()), w_sorted)
        # (c) charge conservation of the deposition (shape factors sum to one, guard
        #     cells are folded back): sum(rho * vol) == q * sum(w), before any filtering
        sim.deposit('rho_prev', update_spectral=False)
        rho0 = sim.fld.interp[0].rho.cpu().numpy()
    vol = 1. / sim.fld.interp[0].invvol
    q_grid = (rho0.real * vol[None, :]).sum()
    achieved(None, abs(q_grid - q_tot) / abs(q_tot), 2e-14, 'charge on the grid')      # 1.2e-15
    assert np.abs(rho0.imag).max() == 0.                  # mode 0 is real
    # (d) linearity of the spectral solve: transform round trip is the identity
    with GpuMemoryManager(sim):
        E0 = sim.fld.interp[1].Er.clone()
        sim.fld.interp2spect('E')
        sim.fld.spect2interp('E')
        err = (sim.fld.interp[1].Er - E0).abs().max().item() / max(E0.abs().max().item(), 1e-300)
    achieved(None, err, 2e-13, 'transform round trip')      # 1.5e-14


def test_empty_and_tiny_species_step():
    """Edge cases of the particle path: a species with no particle, one with a single particle
    sitting exactly on the axis, and both together with a normal species; the cycle must run
    and the normal species must be unaffected by the presence of the empty one."""
    from scipy.constants import e, m_e
    ref = helpers.uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'linear', seed=5)
    sim = helpers.uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'linear', seed=5)
    empty = sim.add_new_species(q=e, m=1836. * m_e)
    assert empty.Ntot == 0
    one = sim.add_new_species(q=0., m=m_e)            # neutral: gathers / deposits nothing
    for k, v in (('x', 0.), ('y', 0.), ('z', 1.e-6), ('ux', 0.), ('uy', 0.), ('uz', 0.5),
                 ('inv_gamma', 1. / np.sqrt(1.25)), ('w', 1.)):
        setattr(one, k, np.array([v]))
    one.Ntot = 1
    for k in ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz'):
        setattr(one, k, np.zeros(1))
    ref.step(3)
    sim.step(3)
    assert sim.ptcl[1].Ntot == 0 and sim.ptcl[2].Ntot == 1
    # the neutral particle just drifts: z advances by 3 * c dt * uz * inv_gamma (mod the box)
    from scipy.constants import c
    L = sim.fld.interp[0].zmax - sim.fld.interp[0].zmin
    z_exp = (1.e-6 + 3 * c * sim.dt * 0.5 / np.sqrt(1.25)) % L
    assert abs(float(sim.ptcl[2].z[0]) % L - z_exp) < 1e-12 * L
    for m in range(2):
        for k in ('Er', 'Ez', 'Bt', 'Jz', 'rho'):
            a, b = getattr(sim.fld.interp[m], k), getattr(ref.fld.interp[m], k)
            # two runs of the same input differ by the order of the deposition atomics only
            assert np.array_equal(a, b) or np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), (m, k)


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_reference_sequence_equals_fused_sequence(oracle, shape):
    """Simulation.reference_sequence (every operation of main.py:346-586 launched on its own, in
    the reference's order, rho_prev re-deposited every step) against the default fused MI355X
    sequence and against the oracle: the sanctioned orchestration differences (bench.py
    config.workload) change the results at rounding level only."""
    def build():
        return helpers.uniform_plasma_sim(64, 32, 2, (2, 2, 8), shape, seed=9, u_th=0.05)
    a, b = build(), build()
    orc = helpers.oracle_from_sim(oracle, a, nthreads=2)
    b.reference_sequence = True
    b.redeposit_rho_prev_every_step = True
    b.fuse_gather_push = False
    b.prerank_in_deposit = False
    for sp in b.ptcl:
        sp.fuse_sort_deposit_rho = False
    a.step(4)
    b.step(4)
    orc.step(4)
    worst = 0.
    for m in range(2):
        for k in INTERP:
            grp = [kk for kk in INTERP if kk[0] == k[0]]
            scale = max(np.abs(orc.interp[mm][kk]).max() for mm in range(2) for kk in grp)
            if scale == 0:
                continue
            ea = np.abs(getattr(a.fld.interp[m], k) - orc.interp[m][k]).max() / scale
            eb = np.abs(getattr(b.fld.interp[m], k) - orc.interp[m][k]).max() / scale
            worst = max(worst, ea, eb)
            achieved(None, max(ea, eb), 5e-12, 'fields vs oracle')      # measured 4.8e-13
    for sa, sb in zip(a.ptcl, b.ptcl):
        ga = np.array([getattr(sa, k) for k in PTCL[:8]])
        gb = np.array([getattr(sb, k) for k in PTCL[:8]])
        o1 = np.lexsort((ga[2], ga[1], ga[0], ga[7]))
        o2 = np.lexsort((gb[2], gb[1], gb[0], gb[7]))
        for j, k in enumerate(PTCL[:8]):
            achieved(None, np.abs(ga[j][o1] - gb[j][o2]).max() / np.abs(ga[j]).max(), 1e-14, 'particles fused vs reference sequence')
    print('reference vs fused sequence (%s): worst deviation from the oracle %.2e' % (shape, worst))
