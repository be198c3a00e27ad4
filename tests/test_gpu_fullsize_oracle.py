"""The BENCHMARKED configurations against the CPU oracle at their own size (round 6).

Until round 5 the full-size runs were held to size-independent properties only (sortedness,
charge conservation, round trips) and the one-pass / fused-spectral step to the oracle on
32 x 16 ... 64 x 32 grids.  Here the oracle steps the same input as the HIP path:

  C2  1024 x 128, Nm = 2, linear, 2 x 4 x 4 = 32 ppc, 4 194 304 macroparticles, 5 iterations:
      the sorting first iteration, three one-pass iterations (65 536 chunks, XCD walk, 32-bit lane
      offsets, sort policy), the next sorting iteration, the fused spectral launch every time;
  C5  the full 2048 x 512 grid, Nm = 4, cubic, with 1 x 1 x 16 = 16 ppc (16.8 M macroparticles:
      what the oracle steps in ~15 s per iteration), 3 iterations;
  C3  the 4096 x 256 moving-window laser-wakefield grid with the window full of plasma (5.95 M
      macroparticles; reference fixture, see test_c3_full_grid_vs_reference_golden).

Bars: B and rho <= 1e-13 of the largest component of their group (the reference's own CPU <-> GPU bar
for one deposition, /root/reference/tests/test_cpu_gpu_deposition.py:96), particles <= 1e-13 matched
one to one, cell indices of the final state bit-exact wherever the particle is not within 1e-9 of a
cell boundary (SURVEY.md 8c tie mask).  E and J: 1e-13 WITHOUT the curl-free current correction
(`correct_currents=False`), 2e-11 with it - the correction adds i k / k^2 (rho_next - rho_prev) / dt
to J (fields/numba_methods.py:63-85): the rounding difference of two depositions of a UNIFORM density
(2e-15 of n e, whatever the summation order) is divided by dt and by k >= 2 pi / (Nz dz), i.e.
multiplied by Nz / 2 pi = 163 cells and by n e c / |J| ~ 560 at u_th = 0.01 before it meets J.
Measured (tools/c2_parity_growth.py, profiles/r06_c2_parity_growth.txt): HIP against the oracle
4e-12 (E), 3.6e-12 (J) after the first step, falling to 8e-13 / 1.2e-12 by step 5; two ORACLE runs that
differ only in their OpenMP thread count (3 / 16: the same arithmetic in another summation order in the
cells a thread boundary cuts) 4e-13 ... 3e-12 over the same steps; at 32 x 16 the same quantity is 1e-13.
Particles are matched through their weights: every macroparticle gets a unique weight
w_i (1 + i 2^-44) on BOTH sides (same seeded input), so the pairing does not depend on the positions
that are being compared.
"""
import numpy as np
import pytest
import helpers
from conftest import achieved
from helpers import PTCL, INTERP

pytestmark = pytest.mark.gpu


def _tag_weights(sim):
    for s in sim.ptcl:
        s.w = s.w * (1. + np.arange(s.Ntot) * 2.**-44)
        assert np.unique(s.w).size == s.Ntot


def _compare_fields(sim, orc, Nm, tol, what):
    """tol: {'E': ..., 'B': ..., 'J': ..., 'r': ...} (first letter of the field name)."""
    worst = {}
    for m in range(Nm):
        for k in INTERP:
            grp = [kk for kk in INTERP if kk[0] == k[0]]
            scale = max(np.abs(orc.interp[mm][kk]).max() for mm in range(Nm) for kk in grp)
            if scale == 0:
                continue
            err = np.abs(np.asarray(getattr(sim.fld.interp[m], k)) - orc.interp[m][k]).max() / scale
            worst[k[0]] = max(worst.get(k[0], 0.), err)
            achieved(None, err, tol[k[0]], '%s %s' % (what, {'E': 'E', 'B': 'B', 'J': 'J', 'r': 'rho'}[k[0]]))
    return worst


def _compare_particles(oracle, sim, orc, tol, what):
    g0 = sim.fld.interp[0]
    worst = 0.
    for s, o in zip(sim.ptcl, orc.species):
        got = np.array([np.asarray(getattr(s, k)) for k in PTCL[:8]])
        ref = np.array([o[k] for k in PTCL[:8]])
        assert got.shape == ref.shape
        o2, o1 = np.argsort(got[7]), np.argsort(ref[7])
        # the weights are unique tags: identical as a set, bit for bit
        assert np.array_equal(got[7][o2], ref[7][o1])
        for j, k in enumerate(PTCL[:7]):
            err = np.abs(got[j][o2] - ref[j][o1]).max() / np.abs(ref[j]).max()
            worst = max(worst, err)
            achieved(None, err, tol, what)
        gx, gy, gz = (np.ascontiguousarray(got[j][o2]) for j in range(3))
        rx, ry, rz = (np.ascontiguousarray(ref[j][o1]) for j in range(3))
        cg = oracle.cell_index(gx, gy, gz, g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr)
        cr = oracle.cell_index(rx, ry, rz, g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr)
        rc = g0.invdr * (np.sqrt(rx**2 + ry**2) - g0.rmin) - 0.5
        zc = g0.invdz * (rz - g0.zmin) - 0.5
        clear = (np.abs(rc - np.round(rc)) > 1e-9) & (np.abs(zc - np.round(zc)) > 1e-9)
        assert clear.sum() > 0.999 * clear.size
        assert np.array_equal(cg[clear], cr[clear]), 'cell indices differ away from cell boundaries'
    return worst


@pytest.mark.parametrize('correct', [True, False])
def test_c2_full_size_vs_oracle(oracle, correct):
    """BASELINE configs[1] = the bench workload, every array of it, against the oracle."""
    from fbpic_amd.main import GpuMemoryManager
    import torch
    sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
    assert sim.ptcl[0].Ntot == 4194304
    _tag_weights(sim)
    orc = helpers.oracle_from_sim(oracle, sim, nthreads=16)
    s = sim.ptcl[0]
    s.keep_sort_outputs = True
    nstep = 5
    with GpuMemoryManager(sim):
        sim.step(nstep, correct_currents=correct)
        # the final sort of the HIP path: cell index of every particle (bit-exact below), sorted
        s.sort_particles(sim.fld)
        ci = s.cell_idx.cpu().numpy()
        xs, ys, zs = (getattr(s, k).cpu().numpy() for k in ('x', 'y', 'z'))
        torch.cuda.synchronize()
    # the path under test is the benchmarked one: one-pass iterations and the fused spectral launch
    assert s.cycle_passes >= 3 and s.cycle_sorts >= 1, (s.cycle_passes, s.cycle_sorts)
    assert sim.fld.spect_cycle_launches >= nstep - 1, sim.fld.spect_cycle_launches
    orc.step(nstep, correct_currents=correct)
    ej = 2e-11 if correct else 1e-13
    wf = _compare_fields(sim, orc, 2, {'E': ej, 'J': ej, 'B': 1e-13, 'r': 1e-13}, 'vs oracle s5')
    wp = _compare_particles(oracle, sim, orc, 1e-13, 'particles vs oracle s5')
    # cell index of the product's own sort == the oracle's cell-index arithmetic on the product's own
    # positions, for EVERY particle (same expression, no tie mask needed), and the order is sorted
    g0 = sim.fld.interp[0]
    ref_ci = oracle.cell_index(xs, ys, zs, g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr)
    assert np.array_equal(ci, ref_ci)
    assert np.all(np.diff(ci) >= 0)
    print('C2 full size, %d steps, correction %s: fields %s, particles %.2e vs the oracle; %d one-pass iterations, '
          '%d sorts' % (nstep, correct, {k: '%.1e' % v for k, v in wf.items()}, wp, s.cycle_passes, s.cycle_sorts))


@pytest.mark.parametrize('correct,nstep,one_pass', [(True, 3, False), (False, 2, False), (False, 2, True)])
def test_c5_full_grid_vs_oracle(oracle, correct, nstep, one_pass):
    """BASELINE configs[4] on its full 2048 x 512 grid, Nm = 4, cubic; 16 macroparticles per cell
    (p_nz = p_nr = 1, p_nt = 16) instead of 64 so that the oracle steps it in seconds.  E and J with the
    current correction: 2e-10 (measured 1e-11 / 4e-11: Nz = 2048 doubles the amplification named in the
    header, 16 instead of 32 macroparticles per cell raise n e c / |J|); without it 1e-13 like B and rho."""
    from fbpic_amd.main import GpuMemoryManager
    sim = helpers.uniform_plasma_sim(2048, 512, 4, (1, 1, 16), 'cubic', seed=0)
    assert sim.ptcl[0].Ntot == 2048 * 512 * 16
    _tag_weights(sim)
    orc = helpers.oracle_from_sim(oracle, sim, nthreads=16)
    # one_pass: the cubic one-pass kernel (k_cycle_cubic, opt-in) instead of the two passes bench.py times
    sim.one_pass_cubic = one_pass
    with GpuMemoryManager(sim):
        sim.step(nstep, correct_currents=correct)
    assert (sim.ptcl[0].cycle_passes > 0) == one_pass
    orc.step(nstep, correct_currents=correct)
    ej = 2e-10 if correct else 1e-13
    wf = _compare_fields(sim, orc, 4, {'E': ej, 'J': ej, 'B': 1e-13, 'r': 1e-13}, 'vs oracle s%d' % nstep)
    wp = _compare_particles(oracle, sim, orc, 1e-13, 'particles vs oracle s%d' % nstep)
    print('C5 full grid (16 ppc), %d steps, correction %s: fields %s, particles %.2e vs the oracle'
          % (nstep, correct, {k: '%.1e' % v for k, v in wf.items()}, wp))


def test_c3_full_grid_vs_reference_golden():
    """BASELINE configs[2] on its OWN grid - 4096 x 256, Nm = 2, open z with damping, moving window at c,
    a0 = 4 Gaussian pulse (docs/source/example_input/lwfa_script.py) - against the REAL reference
    (tests/golden/c3_full_grid.npz, oracle/capture_golden.py:cap_c3_full_grid: the interpreted reference
    needs ~20 min for it).  The plasma is loaded as two cells inside the pulse; the continuous injection of
    the first particle exchange then fills the window to the right of it (5.95 M macroparticles, same
    count here), and two steps run the pulse through it.  Compared: 48 z rows of every grid (20 around
    the plasma edge, 28 over the whole local grid incl. guard and damping cells), sum / sum of squares /
    maximum of EVERY grid over all 4416 x 256 cells, every 300th particle of the (w, x, y, z) order and
    sum / sum of squares of every particle attribute over all particles."""
    from scipy.constants import c
    from conftest import golden
    from fbpic_amd.main import Simulation
    from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
    g = golden('c3_full_grid')
    Nz, Nr, Nm = int(g['Nz']), int(g['Nr']), int(g['Nm'])
    assert (Nz, Nr, Nm) == (4096, 256, 2)
    zmin, zmax, rmax, dt, z_slab = (float(g[k]) for k in ('zmin', 'zmax', 'rmax', 'dt', 'z_slab'))
    dz = (zmax - zmin) / Nz
    np.random.seed(0)
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, p_zmin=z_slab, p_zmax=z_slab + 2 * dz,
                     p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24,
                     n_order=-1, particle_shape='linear', boundaries={'z': 'open', 'r': 'reflective'})
    add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
    sim.set_moving_window(v=c)
    assert sim.fld.Nz == int(g['Nz_local']) and sim.comm.n_guard == int(g['n_guard'])
    assert sim.comm.nz_damp == int(g['nz_damp']) and sim.comm.n_inject == int(g['n_inject'])
    s = sim.ptcl[0]
    ref0 = g['s0_ptcl0']
    assert s.Ntot == ref0.shape[1]
    for j, k in enumerate(PTCL[:8]):
        assert np.array_equal(np.asarray(getattr(s, k)), ref0[j]), k          # same lattice, bit for bit
    for _ in range(int(g['nstep'])):
        sim.step(1)                                # (as the capture: one call per step)
    assert sim.fld.interp[0].zmin == float(g['sf_zmin'])                       # same window motion
    assert s.Ntot == int(g['sf_ntot'])                                         # same injection
    rows = g['sf_rows']
    ref_rows, ref_sum, ref_sum2, ref_max = g['sf_interp_rows'], g['sf_interp_sum'], g['sf_interp_sum2'], g['sf_interp_max']
    for m in range(Nm):
        for i, k in enumerate(INTERP):
            grp = [j for j, kk in enumerate(INTERP) if kk[0] == k[0]]
            scale = ref_max[:, grp].max()
            if scale == 0:
                continue
            F = np.asarray(getattr(sim.fld.interp[m], k))
            what = {'E': 'E', 'B': 'B', 'J': 'J', 'r': 'rho'}[k[0]]
            achieved(None, np.abs(F[rows] - ref_rows[m, i]).max() / scale, 1e-11, 'rows ' + what)
            achieved(None, abs(F.sum() - ref_sum[m, i]) / (scale * F.size), 2e-12, 'mean ' + what)
            achieved(None, abs((np.abs(F)**2).sum() - ref_sum2[m, i]) / ref_sum2[:, grp].max(), 1e-11,
                     'sum of squares ' + what)
            achieved(None, abs(np.abs(F).max() - ref_max[m, i]) / scale, 1e-11, 'maximum ' + what)
    got = np.array([np.asarray(getattr(s, k)) for k in PTCL[:8]])
    o = np.lexsort((got[2], got[1], got[0], got[7]))
    sample = got[:, o[::300]]
    ref = g['sf_ptcl_sample']
    assert sample.shape == ref.shape
    assert np.array_equal(sample[7], ref[7])                                   # the same macroparticles
    for j, k in enumerate(PTCL[:8]):
        sc = np.abs(ref[j]).max()
        if sc > 0:
            achieved(None, np.abs(sample[j] - ref[j]).max() / sc, 1e-11, 'particle sample')
        achieved(None, abs(got[j].sum() - g['sf_ptcl_sum'][j]) / max(np.abs(got[j]).sum(), 1e-300), 1e-11, 'particle sums')
        achieved(None, abs((got[j]**2).sum() - g['sf_ptcl_sum2'][j]) / max(g['sf_ptcl_sum2'][j], 1e-300), 1e-11,
                 'particle sums of squares')
