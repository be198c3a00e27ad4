"""Diagnostics hook: the two hook points of Simulation.step and the
copy-out convention, with .npz dumps instead of openPMD/HDF5."""
import numpy as np
import pytest
from helpers import uniform_plasma_sim

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
            assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1e-300), (m, k)
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
            assert np.abs(d[key][m] - b).max() <= 1e-11 * max(np.abs(b).max(), 1e-300), (m, key)
    p = np.load(tmp_path / 'npz' / 'particles_electrons00000003.npz')
    assert p['x'].shape == (sim.ptcl[0].Ntot,) and set(p.files) >= {'ux', 'w', 'Ex', 'Ez'}
    c = np.load(tmp_path / 'npz' / 'checkpoint00000004.npz')
    assert c['species0_x'].shape == (sim.ptcl[0].Ntot,) and c['m1_Ez'].shape == (32, 16)
