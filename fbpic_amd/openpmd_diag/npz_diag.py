"""Minimal field / particle diagnostics: same constructor arguments, same `write(iteration)`
protocol and same data selection as the reference's `FieldDiagnostic`
(fbpic/openpmd_diag/field_diag.py:16-224) and `ParticleDiagnostic`
(fbpic/openpmd_diag/particle_diag.py:18-210), with one `.npz` file per dump:

  <write_dir>/npz/fields<iteration:08d>[_rank<r>].npz     E_r, E_t, E_z, B_*, J_*, rho as
        complex128[Nm, Nz, Nr] of the rank's physical cells (guard and damp cells stripped,
        like field_diag.py:198-224), plus zmin, zmax, rmax, dz, dr, time, iteration
  <write_dir>/npz/particles_<species><iteration:08d>[_rank<r>].npz   x, y, z, ux, uy, uz, w
        (and the gathered E, B when asked) of the particles inside the physical domain

The data stay on the GPU during the run: only the selected slices are copied out, on the
iterations that are written (the convention of the reference's `get_dataset`, which slices
the device array before `.get()`)."""
import os
import numpy as np

_COMP = {'E': ('Er', 'Et', 'Ez'), 'B': ('Br', 'Bt', 'Bz'), 'J': ('Jr', 'Jt', 'Jz'), 'rho': ('rho',)}


def _to_host(a):
    return a.cpu().numpy() if hasattr(a, 'is_cuda') else np.asarray(a)


class _Periodic(object):
    def __init__(self, period, comm, write_dir, iteration_min, iteration_max, dt_period, dt):
        if period is None and dt_period is None:
            raise ValueError('You need to pass either `period` or `dt_period`.')
        if period is not None and dt_period is not None:
            raise ValueError('Pass either `period` or `dt_period`, not both.')
        if dt_period is not None:
            period = max(1, int(round(dt_period / dt)))
        self.period = int(period)
        self.comm = comm
        self.rank = 0 if comm is None else comm.rank
        self.iteration_min, self.iteration_max = iteration_min, iteration_max
        self.write_dir = os.path.join(os.getcwd() if write_dir is None else write_dir, 'npz')
        if self.rank == 0:
            os.makedirs(self.write_dir, exist_ok=True)

    def due(self, iteration):
        """True when `write(iteration)` will write (Simulation.step keeps the fused
        gather+push launch on the other iterations)."""
        return (iteration % self.period == 0 and self.iteration_min <= iteration < self.iteration_max)

    def _path(self, stem, iteration):
        suffix = '' if (self.comm is None or self.comm.size == 1) else '_rank%d' % self.rank
        os.makedirs(self.write_dir, exist_ok=True)
        return os.path.join(self.write_dir, '%s%08d%s.npz' % (stem, iteration, suffix))


class FieldDiagnostic(_Periodic):
    def __init__(self, period=None, fldobject=None, comm=None, fieldtypes=("rho", "E", "B", "J"),
                 write_dir=None, iteration_min=0, iteration_max=np.inf, dt_period=None):
        if fldobject is None:
            raise ValueError('A Fields object is needed')
        _Periodic.__init__(self, period, comm, write_dir, iteration_min, iteration_max,
                           dt_period, fldobject.dt)
        self.fld = fldobject
        self.fieldtypes = tuple(fieldtypes)
        self.time = 0.

    def write(self, iteration):
        if not self.due(iteration):
            return None
        fld, comm = self.fld, self.comm
        # rho / J: bring the smoothed / corrected values back from spectral space, exchange
        # the guard cells if that has not happened yet (field_diag.py:81-96)
        if 'rho' in self.fieldtypes:
            fld.spect2interp('rho_prev')
            if comm is not None and comm.size > 1 and not fld.exchanged_source['rho_prev']:
                comm.exchange_fields(fld.interp, 'rho', 'add')
        if 'J' in self.fieldtypes:
            fld.spect2interp('J')
            if comm is not None and comm.size > 1 and not fld.exchanged_source['J']:
                comm.exchange_fields(fld.interp, 'J', 'add')
        Nz = fld.Nz
        lo, hi = 0, Nz
        if comm is not None:
            # physical cells of this rank: strip guard cells and, on the end ranks, the damp
            # and injection cells
            nloc, _ = comm.get_Nz_and_iz(local=True, with_damp=False, with_guard=False, rank=comm.rank)
            lo = comm.n_guard + ((comm.nz_damp + comm.n_inject) if comm.rank == 0 else 0)
            hi = lo + nloc
        out = {}
        for ft in self.fieldtypes:
            for name in _COMP[ft]:
                key = '%s_%s' % (ft, name[-1]) if ft != 'rho' else 'rho'
                out[key] = np.stack([_to_host(getattr(fld.interp[m], name)[lo:hi, :fld.Nr])
                                     for m in range(fld.Nm)])
        g0 = fld.interp[0]
        out.update(zmin=g0.zmin + lo * g0.dz, zmax=g0.zmin + hi * g0.dz, dz=g0.dz, dr=g0.dr,
                   rmax=fld.Nr * g0.dr, iteration=iteration, time=iteration * fld.dt)
        path = self._path('fields', iteration)
        np.savez(path, **out)
        return path


class ParticleDiagnostic(_Periodic):
    def __init__(self, period=None, species=None, comm=None,
                 particle_data=("position", "momentum", "weighting"), select=None, write_dir=None,
                 iteration_min=0, iteration_max=np.inf, dt_period=None):
        if not species:
            raise ValueError('A dictionary {name: Particles} is needed')
        dt = next(iter(species.values())).dt
        _Periodic.__init__(self, period, comm, write_dir, iteration_min, iteration_max,
                           dt_period, dt)
        self.species = dict(species)
        self.particle_data = tuple(particle_data)
        self.select = select

    def write(self, iteration):
        if not self.due(iteration):
            return None
        paths = []
        for name, sp in self.species.items():
            sp.flush_pending_push()
            keys = []
            if 'position' in self.particle_data:
                keys += ['x', 'y', 'z']
            if 'momentum' in self.particle_data:
                keys += ['ux', 'uy', 'uz']
            if 'weighting' in self.particle_data:
                keys += ['w']
            if 'E' in self.particle_data:
                keys += ['Ex', 'Ey', 'Ez']
            if 'B' in self.particle_data:
                keys += ['Bx', 'By', 'Bz']
            if 'gamma' in self.particle_data:
                keys += ['inv_gamma']
            z = getattr(sp, 'z')
            mask = None
            if self.comm is not None and self.comm.size > 1:
                zmin, zmax = self.comm.get_zmin_zmax(local=True, with_damp=False, with_guard=False,
                                                     rank=self.comm.rank)
                mask = (z >= zmin) & (z < zmax)
            if self.select:
                for k, (vmin, vmax) in self.select.items():
                    q = getattr(sp, k)
                    m = True
                    if vmin is not None:
                        m = m & (q >= vmin)
                    if vmax is not None:
                        m = m & (q < vmax)
                    mask = m if mask is None else (mask & m)
            out = {k: _to_host(getattr(sp, k) if mask is None else getattr(sp, k)[mask]) for k in keys}
            out.update(q=sp.q, m=sp.m, iteration=iteration, time=iteration * sp.dt)
            path = self._path('particles_%s' % name, iteration)
            np.savez(path, **out)
            paths.append(path)
        return paths


class Checkpoint(_Periodic):
    """End-of-iteration hook (main.py:564-565): everything a restart needs
    (checkpoint_restart.restart_from_checkpoint), one file per rank: all particle arrays of every
    species, E and B on the interpolation grid including guard / damp cells, the position of
    the grid and of the global domain, time and iteration, the continuous position of the
    moving window and the book-keeping of the plasma injectors."""

    def __init__(self, sim, period, write_dir=None):
        _Periodic.__init__(self, period, sim.comm, write_dir, 0, np.inf, None, sim.dt)
        self.sim = sim

    def write(self, iteration):
        if not self.due(iteration):
            return None
        sim, out = self.sim, {}
        for i, sp in enumerate(sim.ptcl):
            sp.flush_pending_push()
            for k in ('x', 'y', 'z', 'ux', 'uy', 'uz', 'w', 'inv_gamma'):
                out['species%d_%s' % (i, k)] = _to_host(getattr(sp, k))
            inj = getattr(sp, 'injector', None)
            if inj is not None and inj.z_inject is not None:
                out['species%d_injector' % i] = np.array([inj.z_inject, inj.z_end_plasma,
                                                          float(inj.nz_inject)])
        for m in range(sim.fld.Nm):
            for k in _COMP['E'] + _COMP['B']:
                out['m%d_%s' % (m, k)] = _to_host(getattr(sim.fld.interp[m], k))
        out.update(iteration=iteration, time=sim.time, zmin=sim.fld.interp[0].zmin,
                   zmax=sim.fld.interp[0].zmax, Nm=sim.fld.Nm,
                   zmin_global=sim.comm._zmin_global_domain)
        win = sim.comm.moving_win
        if win is not None:
            out['window'] = np.array([win.zmin, win.t_last_move])
        path = self._path('checkpoint', iteration)
        np.savez(path, **out)
        return path
