"""Diagnostics hook: the two hook points of Simulation.step and the
copy-out convention, with .npz dumps instead of openPMD/HDF5."""
import numpy as np
import pytest
from helpers import uniform_plasma_sim
from conftest import achieved

pytestmark = pytest.mark.gpu


def test_field_and_particle_dumps(tmp_path):
    from fbpic_amd.openpmd_diag import FieldDiagnostic, ParticleDiagnostic, Checkpoint
    from fbpic_amd.main import GpuMemoryManager
    sim = uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'linear', seed=3)
    ref = uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'linear', seed=3)
    sim.diags = [FieldDiagnostic(2, sim.fld, sim.comm, write_dir=str(tmp_path)),
                 ParticleDiagnostic(3, {'electrons': sim.ptcl[0]}, sim.comm,
                                    particle_data=('position', 'momentum', 'weighting', 'E'),
                                    write_dir=str(tmp_path))]
    sim.checkpoints = [Checkpoint(sim, 4, write_dir=str(tmp_path))]
    with GpuMemoryManager(sim):
        sim.step(5)
    ref.step(5)
    # the hooks do not perturb the run (they only force the unfused gather/push path)
    for m in range(2):
        for k in ('Er', 'Ez', 'Bt', 'Jz', 'rho'):
            a, b = getattr(sim.fld.interp[m], k), getattr(ref.fld.interp[m], k)
            achieved(None, np.abs(a - b).max() / max(np.abs(b).max(), 1e-300), 4e-13, 'run with hooks vs without')
    files = sorted(p.name for p in (tmp_path / 'npz').iterdir())
    assert files == ['checkpoint00000004.npz', 'fields00000000.npz', 'fields00000002.npz',
                     'fields00000004.npz', 'particles_electrons00000000.npz',
                     'particles_electrons00000003.npz']
    # a dump at iteration n holds the fields at time n: compare with a run stopped there
    ref2 = uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'linear', seed=3)
    ref2.step(2)
    d = np.load(tmp_path / 'npz' / 'fields00000002.npz')
    assert d['E_z'].shape == (2, 32, 16) and int(d['iteration']) == 2
    for m in range(2):
        for key, attr in (('E_r', 'Er'), ('E_z', 'Ez'), ('B_t', 'Bt'), ('rho', 'rho')):
            b = getattr(ref2.fld.interp[m], attr)
            achieved(None, np.abs(d[key][m] - b).max() / max(np.abs(b).max(), 1e-300), 5e-13, 'dump vs run stopped there')
    p = np.load(tmp_path / 'npz' / 'particles_electrons00000003.npz')
    assert p['x'].shape == (sim.ptcl[0].Ntot,) and set(p.files) >= {'ux', 'w', 'Ex', 'Ez'}
    c = np.load(tmp_path / 'npz' / 'checkpoint00000004.npz')
    assert c['species0_x'].shape == (sim.ptcl[0].Ntot,) and c['m1_Ez'].shape == (32, 16)


def _state(sim):
    from helpers import INTERP, PTCL
    f = {(m, k): np.array(getattr(sim.fld.interp[m], k)) for m in range(sim.fld.Nm) for k in INTERP}
    p = [np.array([np.array(getattr(s, k)) for k in PTCL[:8]]) for s in sim.ptcl]
    return f, p


def _assert_same(a, b, tol):
    from helpers import INTERP
    fa, pa = a
    fb, pb = b
    for (m, k), ref in fb.items():
        grp = [kk for kk in INTERP if kk[0] == k[0]]
        scale = max(np.abs(fb[(mm, kk)]).max() for (mm, kk) in fb if kk in grp)
        if scale > 0:
            achieved(None, np.abs(fa[(m, k)] - ref).max() / scale, tol, 'fields')
    for got, ref in zip(pa, pb):
        assert got.shape == ref.shape
        o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
        o2 = np.lexsort((got[2], got[1], got[0], got[7]))
        for j in range(8):
            sc = np.abs(ref[j]).max()
            if sc > 0:
                achieved(None, np.abs(got[j][o2] - ref[j][o1]).max() / sc, tol, 'particles')


def test_restart_periodic_run6_equals_run3_restart_run3(tmp_path):
    """restart_from_checkpoint (reference: openpmd_diag/checkpoint_restart.py:77-189;
    tests/test_example_docs_scripts.py:28-51): a run of 6 steps == 3 steps, checkpoint, a NEW
    Simulation filled from the checkpoint, 3 more steps."""
    from fbpic_amd.openpmd_diag import set_periodic_checkpoint, restart_from_checkpoint
    a = uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'cubic', seed=3, u_th=0.05)
    set_periodic_checkpoint(a, 3, checkpoint_dir=str(tmp_path))
    a.step(3)
    a.step(3)
    b = uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'cubic', seed=99, u_th=0.3)   # different content
    it = restart_from_checkpoint(b, 3, checkpoint_dir=str(tmp_path))
    assert it == 3 and b.iteration == 3 and abs(b.time - 3 * b.dt) < 1e-30
    b.step(3)
    assert b.iteration == a.iteration == 6
    _assert_same(_state(b), _state(a), 2e-13)          # measured 1.6e-14
    # the latest checkpoint is picked when no iteration is given
    c = uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'cubic', seed=98)
    assert restart_from_checkpoint(c, checkpoint_dir=str(tmp_path)) == 6
    with pytest.raises(RuntimeError):
        c.ptcl = []
        restart_from_checkpoint(c, checkpoint_dir=str(tmp_path))


def test_restart_lwfa_moving_window(tmp_path):
    """Restart of the laser-wakefield miniature (open z, moving window, continuous injection):
    the checkpoint carries the window position and the injector book-keeping, so the restarted
    run reproduces the uninterrupted one to rounding (the reference, which re-derives the
    injection positions from the particles, only reaches 2e-5)."""
    from test_gpu_lwfa import _build
    from fbpic_amd.openpmd_diag import set_periodic_checkpoint, restart_from_checkpoint
    from scipy.constants import c
    a = _build('linear')
    set_periodic_checkpoint(a, 6, checkpoint_dir=str(tmp_path))
    a.step(6)
    np.random.seed(5)
    a.step(7)
    from fbpic_amd.main import Simulation
    Nz, Nr, Nm = 96, 24, 2
    zmax, zmin, rmax = 12.e-6, -12.e-6, 12.e-6
    np.random.seed(11)
    b = Simulation(Nz, zmax, Nr, rmax, Nm, (zmax - zmin) / Nz / c, zmin=zmin,
                   p_zmin=2.e-6, p_zmax=1., p_rmin=0., p_rmax=10.e-6, p_nz=1, p_nr=2, p_nt=4,
                   n_e=4.e24, n_order=-1, particle_shape='linear',
                   boundaries={'z': 'open', 'r': 'reflective'}, n_guard=16,
                   n_damp={'z': 16, 'r': 8}, exchange_period=4)      # no laser: comes from the file
    restart_from_checkpoint(b, 6, checkpoint_dir=str(tmp_path))
    b.set_moving_window(v=c)
    assert b.iteration == 6 and b.fld.interp[0].zmin != zmin - 40 * (zmax - zmin) / Nz
    np.random.seed(5)
    b.step(7)
    assert b.fld.interp[0].zmin == a.fld.interp[0].zmin and b.ptcl[0].Ntot == a.ptcl[0].Ntot
    _assert_same(_state(b), _state(a), 7e-13)          # measured 6.5e-14


def test_field_dump_against_oracle(oracle, tmp_path):
    """A FieldDiagnostic dump at iteration n against the CPU ORACLE stepped n times (E, B, the
    corrected J and rho of the interpolation grid), not against another HIP run."""
    import helpers
    from fbpic_amd.openpmd_diag import FieldDiagnostic
    sim = uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'linear', seed=4, u_th=0.05)
    orc = helpers.oracle_from_sim(oracle, sim)
    sim.diags = [FieldDiagnostic(3, sim.fld, sim.comm, write_dir=str(tmp_path))]
    sim.step(5)
    orc.step(3)
    d = np.load(tmp_path / 'npz' / 'fields00000003.npz')
    for key, attrs in (('E', ('Er', 'Et', 'Ez')), ('B', ('Br', 'Bt', 'Bz')), ('J', ('Jr', 'Jt', 'Jz')),
                       ('rho', ('rho',))):
        scale = max(np.abs(orc.interp[m][k]).max() for m in range(2) for k in attrs)
        for m in range(2):
            for k in attrs:
                name = 'rho' if key == 'rho' else '%s_%s' % (key, k[-1])
                err = np.abs(d[name][m] - orc.interp[m][k]).max() / scale
                achieved(None, err, 5e-13, 'dump vs oracle')
