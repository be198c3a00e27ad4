"""Every configuration of BASELINE.json under `pytest -m gpu`, at its own size.

  C1  periodic plasma wave 256 x 64, Nm = 2, p_nz, p_nr, p_nt = 1, 1, 4: the node-aligned
      lattice (with one particle per cell along r / z the macroparticles sit
      EXACTLY on cell centres, where ceil / floor of the cell coordinate is a tie) stepped by
      the HIP path and by the CPU oracle; plus the reference's own test at the reference's own
      parameters (Nz = 200, Nr = 64, Nm = 3, n_order = 16, 2 x 2 x 8 ppc, three wavelengths in
      the box, tests/test_periodic_plasma_wave.py:134-165) with its two assertions;
  C2  tests/test_gpu_cycle.py::test_headline_size_properties (+ bench.py);
  C3  laser-wakefield 4096 x 256, Nm = 2, 16 ppc, open z, moving window, continuous
      injection, a0 = 4 Gaussian pulse, stepped with the window full of plasma (> 8 M
      macroparticles); the small-size trajectory against the reference is
      tests/test_gpu_lwfa.py, the decomposed one (C4) tests/test_gpu_multirank_golden.py;
  C5  2048 x 512, Nm = 4, cubic, 64 ppc (67 M macroparticles): whole cycle + properties;
      Hankel GEMM at (2048, 512) and (4096, 256) against np.dot; cubic Nm = 4 cycle against
      the oracle at a size the oracle steps in seconds.
"""
import numpy as np
import pytest
from scipy.constants import c, e, m_e, epsilon_0
import helpers
from conftest import achieved
from helpers import PTCL, INTERP

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------ C1
def _c1_sim(shape):
    """BASELINE configs[0]: the periodic plasma wave at Nz = 256, Nr = 64, Nm = 2, 1 x 1 x 4."""
    from fbpic_amd.main import Simulation
    Nz, Nr, Nm = 256, 64, 2
    dz = 0.2e-6
    zmax, rmax = Nz * dz, 20.e-6
    n_e, w0 = 2.e24, 5.e-6
    k0 = 2 * np.pi / zmax * 3
    wp = np.sqrt(n_e * e**2 / (m_e * epsilon_0))
    np.random.seed(0)
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dz / c, 0., zmax + dz, 0., 18.e-6, 1, 1, 4, n_e,
                     n_order=-1, particle_shape=shape)
    s = sim.ptcl[0]
    x, y, z = s.x, s.y, s.z
    r = np.sqrt(x**2 + y**2)
    ex = np.exp(-r**2 / w0**2)
    eps = (0.001, 0.001)
    # impart_momenta of the reference test (:300-311) at t = 0, modes 0 and 1
    s.ux = (eps[0] * c / wp * 2 * x / w0**2 - eps[1] * c / wp * 2 / w0
            + eps[1] * c / wp * 4 * x**2 / w0**3) * ex * np.sin(k0 * z)
    s.uy = (eps[0] * c / wp * 2 * y / w0**2 + eps[1] * c / wp * 4 * x * y / w0**3) * ex * np.sin(k0 * z)
    s.uz = (-eps[0] * c / wp * k0 - eps[1] * c / wp * k0 * 2 * x / w0) * ex * np.cos(k0 * z)
    s.inv_gamma = 1. / np.sqrt(1 + s.ux**2 + s.uy**2 + s.uz**2)
    return sim


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_c1_node_aligned_lattice_vs_oracle(oracle, shape):
    from fbpic_amd.main import GpuMemoryManager
    sim = _c1_sim(shape)
    s = sim.ptcl[0]
    assert s.Ntot == 256 * 58 * 4          # 58 cells below p_rmax = 18 um, 1 x 1 x 4 per cell
    g0 = sim.fld.interp[0]
    # the lattice really is node aligned: r_cell of every particle is an integer (+- rounding)
    r_cell = g0.invdr * (np.sqrt(s.x**2 + s.y**2) - g0.rmin) - 0.5
    assert np.abs(r_cell - np.round(r_cell)).max() < 1e-6
    # (a) cell indices of the tie lattice: same arithmetic as the oracle -> bit-identical,
    #     ties included
    ref_cell = oracle.cell_index(s.x, s.y, s.z, g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr)
    orc = helpers.oracle_from_sim(oracle, sim, nthreads=4)
    s.keep_sort_outputs = True
    s.use_bin_sort = False                  # stable three-stage sort: cell_idx of every particle
    with GpuMemoryManager(sim):
        s.sort_particles(sim.fld)
        got_cell = np.sort(s.cell_idx.cpu().numpy())
    assert np.array_equal(got_cell, np.sort(ref_cell))
    s.use_bin_sort = True
    # (b) whole cycle
    nstep = 6
    sim.step(nstep)
    orc.step(nstep)
    worst = 0.
    for m in range(2):
        for k in INTERP:
            grp = [kk for kk in INTERP if kk[0] == k[0]]
            scale = max(np.abs(orc.interp[mm][kk]).max() for mm in range(2) for kk in grp)
            if scale == 0:
                continue
            err = np.abs(getattr(sim.fld.interp[m], k) - orc.interp[m][k]).max() / scale
            worst = max(worst, err)
            achieved(None, err, 2e-11, 'fields vs oracle')
    o = orc.species[0]
    got = np.array([getattr(s, k) for k in PTCL[:8]])
    ref = np.array([o[k] for k in PTCL[:8]])
    o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
    o2 = np.lexsort((got[2], got[1], got[0], got[7]))
    for j, k in enumerate(PTCL[:8]):
        achieved(None, np.abs(got[j][o2] - ref[j][o1]).max() / np.abs(ref[j]).max(), 1.5e-12, 'particles vs oracle')
    # (c) cell indices after the steps: identical wherever the particle is not within 1e-9 of
    #     a cell boundary (SURVEY.md 8c tie mask; momenta differ by ~1e-13 between the paths)
    gx, gy, gz = got[0][o2], got[1][o2], got[2][o2]
    rx, ry, rz = ref[0][o1], ref[1][o1], ref[2][o1]
    cg = oracle.cell_index(gx, gy, gz, g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr)
    cr = oracle.cell_index(rx, ry, rz, g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr)
    rc = g0.invdr * (np.sqrt(rx**2 + ry**2) - g0.rmin) - 0.5
    zc = g0.invdz * (rz - g0.zmin) - 0.5
    clear = (np.abs(rc - np.round(rc)) > 1e-9) & (np.abs(zc - np.round(zc)) > 1e-9)
    assert clear.sum() > 0.9 * clear.size
    assert np.array_equal(cg[clear], cr[clear])
    print('C1 %s: worst field deviation from the oracle after %d steps %.2e' % (shape, nstep, worst))


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_periodic_plasma_wave_at_reference_parameters(shape):
    """tests/test_periodic_plasma_wave.py as the reference runs it: Nz = 200 (not a power of
    two), Nr = 64, Nm = 3, n_order = 16, 2 x 2 x 8 ppc, N_periods = 3, 0.75 plasma period;
    assertions of :355-362 (div E - rho / eps0 < 1e-11 in spectral space, every mode) and
    :407-409 (E vs linear theory, atol 1.1e6, rtol 2e-2)."""
    from test_gpu_cycle import _plasma_wave
    from fbpic_amd.main import GpuMemoryManager
    sim, rho_ions, (eps, k0, w0, wp) = _plasma_wave(shape, Nz=200, Nr=64, Nm=3, ppc=(2, 2, 8),
                                                   n_periods=3, n_order=16)
    fld = sim.fld
    assert (fld.Nz, fld.Nr, fld.Nm) == (200, 64, 3) and sim.ptcl[0].Ntot == 200 * 58 * 32
    assert sim.iteration == int(2 * np.pi / (wp * sim.dt) * 0.75)
    t = sim.time
    g0 = fld.interp[0]
    rr, zz = np.meshgrid(g0.r, g0.z)
    pref = m_e * c**2 / e
    ex = np.exp(-rr**2 / w0**2)
    Ez_th = (-eps[0] * k0 - eps[1] * k0 * 2 * rr / w0 - eps[2] * k0 * 4 * rr**2 / w0**2) \
        * pref * ex * np.cos(k0 * zz) * np.sin(wp * t)
    Er_th = (eps[0] * 2 * rr / w0**2 - eps[1] * 2 / w0 + eps[1] * 4 * rr**2 / w0**3
             - eps[2] * 8 * rr / w0**2 + eps[2] * 8 * rr**3 / w0**4) * pref * ex * np.sin(k0 * zz) * np.sin(wp * t)
    Ez_sim = fld.interp[0].Ez.real + sum(2 * fld.interp[m].Ez.real for m in range(1, 3))
    Er_sim = fld.interp[0].Er.real + sum(2 * fld.interp[m].Er.real for m in range(1, 3))
    assert np.abs(Ez_th).max() > 1e8                       # the wave is there
    assert np.allclose(Ez_th, Ez_sim, atol=1.1e6, rtol=2e-2)
    assert np.allclose(Er_th, Er_sim, atol=1.1e6, rtol=2e-2)
    for m in range(3):
        fld.interp[m].rho = fld.interp[m].rho + rho_ions[m]
    with GpuMemoryManager(sim):
        fld.interp2spect('E')
        fld.interp2spect('rho_prev')
    for m in range(3):
        sp = fld.spect[m]
        divE = sp.kr * (sp.Ep - sp.Em) + 1.j * sp.kz * sp.Ez
        rho_eps0 = sp.rho_prev / epsilon_0
        rel = np.sqrt(np.sum(abs(divE - rho_eps0)**2) / np.sum(abs(rho_eps0)**2))
        print('plasma wave %s, mode %d: relative error on div E %.2e' % (shape, m, rel))
        achieved(None, rel, 1.e-11, 'divE - rho/eps0')


# ------------------------------------------------------------------------------ C5
@pytest.mark.parametrize('Nz,Nr', [(2048, 512), (4096, 256)])
def test_hankel_gemm_large(Nz, Nr):
    """fb_hankel at the C5 (2048 x 512) and C3 (4096 x 256) shapes against np.dot: random
    complex input, the Hankel matrices of modes 0 and 3 (asymmetric, entries over many
    decades), strided slab rows."""
    import torch
    from fbpic_amd import _capi
    from fbpic_amd.fields.spectral_transform.hankel import DHT
    rng = np.random.default_rng(Nz + Nr)
    dev = _capi.require_device()
    rmax = Nr * 0.2e-6
    mats = [DHT(0, 0, Nr, Nz, rmax).M, DHT(4, 3, Nr, Nz, rmax).invM, DHT(1, 0, Nr, Nz, rmax).M]
    a = rng.normal(size=(3, Nz, Nr)) + 1j * rng.normal(size=(3, Nz, Nr))
    slab = torch.zeros((Nz, 3, Nr + 8), dtype=torch.complex128, device=dev)
    out = torch.zeros_like(slab)
    for j in range(3):
        slab[:, j, :Nr] = torch.from_numpy(a[j]).to(dev)
    d_m = [torch.from_numpy(np.ascontiguousarray(m)).to(dev) for m in mats]
    ins = [slab[:, j, :Nr] for j in range(3)]
    outs = [out[:, j, :Nr] for j in range(3)]
    rc = _capi.lib().fb_hankel(3, _capi.ptr_array(ins), slab.stride(0), _capi.ptr_array(outs),
                               out.stride(0), _capi.ptr_array(d_m), 1., Nz, Nr, _capi.stream())
    _capi.check(rc, 'fb_hankel')
    torch.cuda.synchronize()
    for j in range(3):
        ref = np.dot(a[j], mats[j])
        got = outs[j].cpu().numpy()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print('hankel %dx%d job %d: %.2e' % (Nz, Nr, j, err))
        achieved(None, err, 2e-14, 'vs np.dot')


def test_cycle_cubic_nm4_vs_oracle(oracle):
    """C5's kernels (cubic shape, Nm = 4, p_nt = 16) on a grid the oracle steps in seconds."""
    sim = helpers.uniform_plasma_sim(64, 48, 4, (2, 2, 16), 'cubic', seed=7, u_th=0.05)
    orc = helpers.oracle_from_sim(oracle, sim, nthreads=4)
    sim.step(3)
    orc.step(3)
    for m in range(4):
        for k in INTERP:
            grp = [kk for kk in INTERP if kk[0] == k[0]]
            scale = max(np.abs(orc.interp[mm][kk]).max() for mm in range(4) for kk in grp)
            if scale == 0:
                continue
            err = np.abs(getattr(sim.fld.interp[m], k) - orc.interp[m][k]).max() / scale
            achieved(None, err, 5e-12, 'fields vs oracle')          # measured 6.8e-13
    s, o = sim.ptcl[0], orc.species[0]
    got = np.array([getattr(s, k) for k in PTCL[:8]])
    ref = np.array([o[k] for k in PTCL[:8]])
    o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
    o2 = np.lexsort((got[2], got[1], got[0], got[7]))
    for j, k in enumerate(PTCL[:8]):
        achieved(None, np.abs(got[j][o2] - ref[j][o1]).max() / np.abs(ref[j]).max(), 1e-14, 'particles vs oracle')


def test_c5_size_properties():
    """C5 at full size (2048 x 512, Nm = 4, cubic, 2 x 2 x 16 = 64 ppc, 67 108 864
    macroparticles, ~16 GB of particle data on the device): properties that need no oracle."""
    import torch
    from fbpic_amd.main import GpuMemoryManager
    sim = helpers.uniform_plasma_sim(2048, 512, 4, (2, 2, 16), 'cubic', seed=0)
    s = sim.ptcl[0]
    s.keep_sort_outputs = True
    n = s.Ntot
    assert n == 67108864
    w_sum = s.w.sum()
    q_tot = s.q * w_sum
    with GpuMemoryManager(sim):
        sim.step(2)
        # (a) sortedness and prefix sum after the last deposit
        ci = s.cell_idx
        assert bool((ci[1:] >= ci[:-1]).all()) and int(s.prefix_sum[-1]) == n
        counts = torch.bincount(ci.long(), minlength=s.prefix_sum.shape[0])
        assert torch.equal(torch.cumsum(counts, 0).int(), s.prefix_sum)
        # (b) the sort is a permutation of the weights
        assert abs(float(s.w.sum()) - w_sum) < 1e-9 * w_sum
        assert float(s.w.min()) > 0.
        # (c) charge conservation of the cubic 4-mode deposition, guard cells folded back
        sim.deposit('rho_prev', update_spectral=False)
        rho0 = sim.fld.interp[0].rho.cpu().numpy()
        rho3 = sim.fld.interp[3].rho.cpu().numpy()
        # (d) transform round trip at Nr = 512 (Hankel GEMM + z-FFT, forward and back)
        E0 = sim.fld.interp[3].Er.clone()
        sim.fld.interp2spect('E')
        sim.fld.spect2interp('E')
        err = (sim.fld.interp[3].Er - E0).abs().max().item() / max(E0.abs().max().item(), 1e-300)
        # (e) fields stay finite and the thermal plasma does not blow up
        assert bool(torch.isfinite(torch.view_as_real(sim.fld.interp[0].Ez)).all())
        umax = max(float(getattr(s, k).abs().max()) for k in ('ux', 'uy', 'uz'))
    vol = 1. / sim.fld.interp[0].invvol
    q_grid = (rho0.real * vol[None, :]).sum()
    achieved(None, abs(q_grid - q_tot) / abs(q_tot), 2e-14, 'charge on the grid')
    assert np.abs(rho0.imag).max() == 0.
    assert np.isfinite(rho3).all()
    achieved(None, err, 1.5e-12, 'transform round trip')
    assert umax < 0.1


# ------------------------------------------------------------------------------ C3
def test_c3_lwfa_full_size():
    """BASELINE configs[2]: docs/source/example_input/lwfa_script.py at 4096 x 256, Nm = 2,
    16 ppc, open z with damping, moving window at c, continuous injection, a0 = 4 Gaussian
    pulse.  The plasma profile starts inside the initial box so that the window holds
    > 8 M macroparticles from the first step (stepping 4096 x dt until the window has filled
    itself would only repeat the injection path, which the steps below exercise anyway)."""
    import torch
    from fbpic_amd.main import Simulation, GpuMemoryManager
    from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
    zmin, zmax, rmax = -10.e-6, 30.e-6, 20.e-6
    Nz, Nr, Nm = 4096, 256, 2
    dz = (zmax - zmin) / Nz
    dt = dz / c
    ramp_start, ramp_length = 5.e-6, 10.e-6

    def dens_func(z, r):
        n = np.ones_like(z)
        n = np.where(z < ramp_start + ramp_length, (z - ramp_start) / ramp_length, n)
        return np.where(z < ramp_start, 0., n)
    np.random.seed(0)
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, p_zmin=5.e-6, p_zmax=500.e-6,
                     p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24,
                     dens_func=dens_func, n_order=-1, particle_shape='linear',
                     boundaries={'z': 'open', 'r': 'reflective'})
    add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
    sim.set_moving_window(v=c)
    s = sim.ptcl[0]
    n0 = s.Ntot
    assert n0 > 8.e6
    nstep = 40
    per_cell_z = 2 * 2 * 230 * 4      # p_nz * p_nr * (230 cells below p_rmax = 18 um) * p_nt
    Emax0 = max(np.abs(sim.fld.interp[1].Er).max(), np.abs(sim.fld.interp[1].Et).max())
    with GpuMemoryManager(sim):
        sim.step(nstep)
        n1 = s.Ntot
        zmin_w = sim.fld.interp[0].zmin
        # every macroparticle is inside the local box, nothing behind the moving window
        assert float(s.z.min()) >= zmin_w and float(s.z.max()) <= sim.fld.interp[0].zmax
        for m in range(Nm):
            for k in ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz', 'Jz', 'rho'):
                assert bool(torch.isfinite(torch.view_as_real(getattr(sim.fld.interp[m], k))).all()), (m, k)
        Emax1 = max(float(sim.fld.interp[1].Er.abs().max()), float(sim.fld.interp[1].Et.abs().max()))
        # the laser is still there, undamped (it sits far from the damping layers) ...
        assert 0.5 * Emax0 < Emax1 < 1.5 * Emax0
        # ... and drives a wake: mode-0 Ez and a plasma current exist where there were none
        assert float(sim.fld.interp[0].Ez.abs().max()) > 1e6
        assert float(sim.fld.interp[0].Jz.abs().max()) > 0.
        umax = float(s.uz.abs().max())
    # the window moved by nstep cells; the plasma uncovered at the right edge was injected
    moved = (zmin_w - (zmin - (sim.comm.n_guard + sim.comm.nz_damp + sim.comm.n_inject) * dz)) / dz
    assert nstep - 1.01 <= moved <= nstep + 0.01, moved
    # continuous injection (at the particle exchanges, every exchange_period steps): whole
    # lattice planes of p_nr * 230 * p_nt macroparticles, at least the nstep cells the window
    # uncovered (the first injection also fills the damping cells up to z_inject)
    plane = per_cell_z // 2
    assert (n1 - n0) % plane == 0
    assert nstep * per_cell_z <= n1 - n0 <= (nstep + sim.comm.nz_damp + sim.comm.n_guard
                                             + 2 * sim.comm.exchange_period) * per_cell_z
    assert 0.01 < umax < 50.
    # the moving window does not rule the one-pass form out (re-keyed home cells, Particles._home_shift):
    # it is probed after the sorts - and given up again for `cycle_suspend_iterations` iterations
    # each time the wake behind the a0 = 4 pulse fills whole chunks with particles that have left
    # their home cell (the bad-chunk policy, particles.py)
    assert s.cycle_passes >= 1 and s.cycle_passes + s.cycle_sorts >= nstep - 2, (s.cycle_passes, s.cycle_sorts)
    print('C3: %d -> %d macroparticles over %d steps, %d one-pass iterations' % (n0, n1, nstep, s.cycle_passes))


def test_c2_step1_loop_costs_what_stepN_costs():
    """State carried across step() calls (fbpic_amd/main.py `_carry_signature`): at the headline
    size `for _ in range(20): sim.step(1)` runs within 10 % of `sim.step(20)` (the reference
    repeats the particle exchange, the rho_prev deposition and the E, B transform at the first
    iteration of every call, main.py:403-451; here only while somebody touched the data)."""
    import time
    import torch
    from fbpic_amd.main import GpuMemoryManager
    sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
    n = 20
    with GpuMemoryManager(sim):
        sim.step(5)
        torch.cuda.synchronize()
        one, loop = [], []
        for rep in range(4):
            t0 = time.perf_counter()
            sim.step(n)
            torch.cuda.synchronize()
            one.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            for _ in range(n):
                sim.step(1)
            torch.cuda.synchronize()
            loop.append(time.perf_counter() - t0)
            assert sim._last_call_carried
    ratio = min(loop) / min(one)
    print('step(%d): %.4f ms/step, %d x step(1): %.4f ms/step, ratio %.3f'
          % (n, 1e3 * min(one) / n, n, 1e3 * min(loop) / n, ratio))
    achieved(None, ratio, 1.10, 'time ratio')
