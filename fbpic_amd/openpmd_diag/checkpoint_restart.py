"""Checkpoints and restart, same two entry points as the reference
(fbpic/openpmd_diag/checkpoint_restart.py:22-189): `set_periodic_checkpoint(sim, period,
checkpoint_dir)` registers the end-of-iteration dump (main.py:564-565) and
`restart_from_checkpoint(sim, iteration, checkpoint_dir)` fills a freshly constructed
`Simulation` with it: time and iteration, position of the box (`comm` global domain, grid
zmin / zmax), E and B on the whole local grid (guard and damp cells included), size and
content of the particle arrays.  Files are `.npz` (one per rank and iteration) instead of
openPMD/HDF5.

Like the reference, everything else (diagnostics, moving window, laser antenna ...) is set up
by the input script; call `restart_from_checkpoint` BEFORE `set_moving_window`.  Beyond the
reference, the dump also carries the state that makes a restart exact instead of "equal up to
the timing of the continuous injection" (the reference's own test accepts 2e-5 for that
reason, tests/test_example_docs_scripts.py:33-38): the continuous position of the moving
window and the book-keeping of the plasma injector; they are restored when the restarted
simulation has a moving window / injector, by `set_moving_window` (which picks up
`sim._restart_window`) and here."""
import glob
import os
import re
import numpy as np
from .npz_diag import Checkpoint

_STATE = ('x', 'y', 'z', 'ux', 'uy', 'uz', 'w', 'inv_gamma')
_EB = ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz')


def set_periodic_checkpoint(sim, period, checkpoint_dir='./checkpoints'):
    """Register a checkpoint every `period` iterations (written at the END of the PIC loop,
    whereas regular diagnostics are written at its beginning)."""
    if sim.comm.rank == 0:
        os.makedirs(checkpoint_dir, exist_ok=True)
    sim.checkpoints.append(Checkpoint(sim, period, write_dir=checkpoint_dir))


def _rank_suffix(comm):
    return '' if comm.size == 1 else '_rank%d' % comm.rank


def available_iterations(sim, checkpoint_dir='./checkpoints'):
    pat = os.path.join(checkpoint_dir, 'npz', 'checkpoint*%s.npz' % _rank_suffix(sim.comm))
    its = []
    for p in glob.glob(pat):
        m = re.match(r'checkpoint(\d{8})%s\.npz$' % _rank_suffix(sim.comm), os.path.basename(p))
        if m:
            its.append(int(m.group(1)))
    return sorted(its)


def restart_from_checkpoint(sim, iteration=None, checkpoint_dir='./checkpoints'):
    """Overwrite time / iteration, box position, E, B and the particles of `sim` with a
    checkpoint (the latest one when `iteration` is None)."""
    if not os.path.exists(checkpoint_dir):
        raise RuntimeError('The directory %s, which is required to restart a simulation, '
                           'does not exist.' % checkpoint_dir)
    its = available_iterations(sim, checkpoint_dir)
    if not its:
        raise RuntimeError('No checkpoint of rank %d (of %d) in %s: a restart needs the same '
                           'number of ranks as the run that wrote the checkpoints.'
                           % (sim.comm.rank, sim.comm.size, checkpoint_dir))
    if iteration is None:
        iteration = its[-1]
    iteration = min(its, key=lambda i: abs(i - iteration))      # closest one, as the reference
    path = os.path.join(checkpoint_dir, 'npz',
                        'checkpoint%08d%s.npz' % (iteration, _rank_suffix(sim.comm)))
    d = np.load(path, allow_pickle=False)
    nsp = len([k for k in d.files if k.endswith('_x') and k.startswith('species')])
    if nsp != len(sim.ptcl):
        raise RuntimeError('Species numbers in checkpoint and simulation should be same, but got '
                           '%d and %d. Use add_new_species method to add species to simulation '
                           'or sim.ptcl = [] to remove them' % (nsp, len(sim.ptcl)))
    fld = sim.fld
    if d['m0_Er'].shape != (fld.Nz, fld.Nr) or int(d['Nm']) != fld.Nm:
        raise RuntimeError('The grid of the checkpoint (%s, Nm=%d) differs from that of the '
                           'simulation (%s, Nm=%d)' % (d['m0_Er'].shape, int(d['Nm']),
                                                       (fld.Nz, fld.Nr), fld.Nm))
    was_on_gpu = fld.data_is_on_gpu
    if was_on_gpu:
        fld.receive_fields_from_gpu()
    sim.iteration = int(d['iteration'])
    sim.time = float(d['time'])
    # particles (load_species, checkpoint_restart.py:220-275)
    for i, sp in enumerate(sim.ptcl):
        gpu = sp.data_is_on_gpu
        if gpu:
            sp.receive_particles_from_gpu()
        for k in _STATE:
            setattr(sp, k, np.array(d['species%d_%s' % (i, k)], dtype=np.float64))
        sp.Ntot = int(sp.x.shape[0])
        for k in ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz'):
            setattr(sp, k, np.zeros(sp.Ntot))
        sp.sorted = False
        sp._pending_push = None
        sp._prerank = None
        sp.cell_idx = None               # device helpers are re-sized at the next upload
        if sp.injector is not None:
            sp.injector.reset_injection_positions()
            key = 'species%d_injector' % i
            if key in d.files and np.all(np.isfinite(d[key])):
                sp.injector.z_inject, sp.injector.z_end_plasma = float(d[key][0]), float(d[key][1])
                sp.injector.nz_inject = int(d[key][2])
        if gpu:
            sp.send_particles_to_gpu()
    # fields and box position (load_fields, :277-330)
    zmin_old = fld.interp[0].zmin
    for m in range(fld.Nm):
        g = fld.interp[m]
        for k in _EB:
            setattr(g, k, np.array(d['m%d_%s' % (m, k)], dtype=np.complex128))
        g.zmin, g.zmax = float(d['zmin']), float(d['zmax'])
    sim.comm.shift_global_domain_positions(fld.interp[0].zmin - zmin_old)
    if 'zmin_global' in d.files:
        sim.comm._zmin_global_domain = float(d['zmin_global'])
    # state of the moving window, applied by set_moving_window (or here if it already exists)
    sim._restart_window = None
    if 'window' in d.files and np.all(np.isfinite(d['window'])):
        sim._restart_window = (float(d['window'][0]), float(d['window'][1]))
        if sim.comm.moving_win is not None:
            sim.comm.moving_win.zmin, sim.comm.moving_win.t_last_move = sim._restart_window
    if was_on_gpu:
        fld.send_fields_to_gpu()
    return iteration
