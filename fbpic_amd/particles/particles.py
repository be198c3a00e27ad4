"""`Particles`: one macroparticle species (structure of arrays) and its kernels.

Same surface as the reference class (fbpic/particles/particles.py:52-1094): attributes
x, y, z, ux, uy, uz, inv_gamma, w, Ex..Bz, q, m, Ntot, cell_idx, sorted_idx, prefix_sum,
sorted, keep_fields_sorted; methods push_p, push_x, gather, deposit, sort_particles,
rearrange_particle_arrays, send_particles_to_gpu, receive_particles_from_gpu.
Every compute method launches a HIP kernel of libfbpic_amd.so; nothing runs on the host.
"""
import ctypes
import os
import numpy as np
from scipy.constants import c
from .. import _capi
from .injection import generate_evenly_spaced, ContinuousInjector

_SHAPE = {'linear': 1, 'cubic': 3}
# the cubic J deposition rides along in the fused pass too since its engine holds a quarter of the
# accumulators (2048 x 512 x 64 ppc: Nm = 4 15.4 -> 14.8 ms per step, Nm = 2 9.5 -> 8.8)
_FUSE_J_CUBIC = os.environ.get('FBPIC_AMD_FUSE_J_CUBIC', '1') == '1'
_STATE = ('x', 'y', 'z', 'ux', 'uy', 'uz', 'w', 'inv_gamma')
_FIELDS = ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz')


class Particles(object):
    def __init__(self, q, m, n, Npz, zmin, zmax, Npr, rmin, rmax, Nptheta, dt,
                 ux_m=0., uy_m=0., uz_m=0., ux_th=0., uy_th=0., uz_th=0.,
                 dens_func=None, continuous_injection=True, grid_shape=None,
                 particle_shape='linear', use_cuda=True, dz_particles=None, is_tracer=False):
        if particle_shape not in _SHAPE:
            raise ValueError("`particle_shape` should be either 'linear' or 'cubic' "
                             "but is `%s`" % particle_shape)
        self.use_cuda = use_cuda
        self.data_is_on_gpu = False
        Ntot, x, y, z, ux, uy, uz, inv_gamma, w = generate_evenly_spaced(
            Npz, zmin, zmax, Npr, rmin, rmax, Nptheta, n, dens_func,
            ux_m, uy_m, uz_m, ux_th, uy_th, uz_th)
        self.Ntot = Ntot
        self.q, self.m, self.dt = q, m, dt
        self.is_tracer = is_tracer
        self.x, self.y, self.z = x, y, z
        self.ux, self.uy, self.uz = ux, uy, uz
        self.inv_gamma, self.w = inv_gamma, w
        for k in _FIELDS:
            setattr(self, k, np.zeros(Ntot))
        # continuous injection behind a moving window (reference :194-203)
        self.continuous_injection = continuous_injection
        if continuous_injection:
            self.injector = ContinuousInjector(Npz, zmin, zmax, dz_particles, Npr, rmin, rmax,
                                               Nptheta, n, dens_func, ux_m, uy_m, uz_m,
                                               ux_th, uy_th, uz_th)
        else:
            self.injector = None
        self.tracker = None
        self.ionizer = None
        self.compton_scatterer = None
        self.n_integer_quantities = 0
        self.n_float_quantities = 8
        self.particle_shape = particle_shape
        self.keep_fields_sorted = False
        if grid_shape is None:
            raise ValueError("A `grid_shape` is needed (the HIP backend sorts particles per cell).")
        self.grid_shape = grid_shape
        self.prefix_sum_shift = 0
        self.sorted = False
        self._pending_push = None     # deferred push_x (dt, x_push, y_push, z_push)
        # Set by Simulation.step before deposit('J'): the push_x that will follow it.  The J
        # deposition then also ranks the particles for the sort after that push
        # (fb_deposit_J_rank_next); `_prerank` remembers for which push the ranks are valid
        # and is dropped by anything that touches the particle arrays in between.
        self.push_after_deposit_J = None
        self._prerank = None
        # Sort policy.  The reference re-sorts before every deposit that follows a push_x.
        # The HIP deposition does not need an exact sort (it accumulates runs of equal
        # cells), so a re-sort is only worth its cost once the particles have moved far
        # enough to fragment those runs: `sort_tolerance` is the displacement bound, in
        # cells (c * dt_push / min(dz, dr) summed since the last sort), above which
        # `deposit` sorts again.  0 restores the reference behaviour.
        self.sort_tolerance = 0.75
        self._moved_since_sort = np.inf
        # Adaptive part of the policy: the J deposition reports how many runs of equal
        # cells it met (read back asynchronously, one step late).  While the run count stays
        # below `resort_fragmentation` x the count measured right after a sort, the sort is
        # skipped even if the displacement bound is exceeded (slow plasmas), up to
        # `max_deposits_between_sorts` deposits.  resort_fragmentation = 0 (default) always
        # obeys the displacement bound: on MI355X the radix sort (~0.26 ms at 4.2 M particles)
        # costs less than the fragmentation it removes from gather + both depositions after
        # two steps even for a 0.01 c thermal plasma (measured, profiles/README.md).
        self.resort_fragmentation = 0.
        # counting sort (fb_bin_sort_particles) instead of cell_index + radix sort + permute;
        # False restores the reference-like stable three-stage sort
        self.use_bin_sort = True
        # push_x + counting sort + deposit('rho') as one destination-ordered pass
        # (fb_push_x_sort_deposit_rho) when a deferred push and a re-sort precede a rho deposit
        self.fuse_sort_deposit_rho = os.environ.get('FBPIC_AMD_FUSE_RHO', '1') != '0'
        # Set by Simulation.step before deposit('J') when push_x + deposit('rho_next') follow
        # with nothing in between: the J deposition is then not launched on its own but rides
        # along in that pass (fb_push_x_sort_deposit_J_rho).  `_pending_J` remembers the target;
        # anything that needs J on the grid or touches the particles first launches it
        # (flush_pending_J).
        self.defer_J_deposit = False
        self._pending_J = None
        # the counting sort only materialises `cell_idx` / `sorted_idx` (sorted cell of every
        # particle, permutation) when asked: nothing on the hot path reads them
        self.keep_sort_outputs = False
        self.max_deposits_between_sorts = 16
        self._runs_after_sort = None
        self._runs_latest = None
        self._deposits_since_sort = 0
        self._stat_pending = None
        # device-only helpers (allocated in send_particles_to_gpu)
        self.cell_idx = None
        self.sorted_idx = None
        self.prefix_sum = None
        self.sorting_buffer = None
        self._alt = None
        self._sort_ws = None
        self._counts_clean = False
        self._cell_size = None
        self._epoch = 0               # bumped by every host -> device copy of the arrays
        self._deferred_fields = None  # see defer_fields
        self._field_store = None
        # `prefix_sum` is the exact inclusive per-cell count of the arrays as they are now (set by
        # the sorts, dropped when particles are added / removed)
        self._prefix_valid = False
        # One-pass particle cycle (Particles.cycle, csrc/cycle.hip): the arrays are re-sorted only
        # every few steps; `cell_idx` then holds the HOME cell of every particle (its cell at that
        # sort) and stays valid while nothing permutes or re-sizes the arrays.  Re-sort policy:
        # after `cycle_sort_period` passes, or earlier when the share of particles that have left
        # their home cell (counted by the pass itself, read back one step late) exceeds
        # `cycle_stray_limit`.
        # Calls of the public compute methods from OUTSIDE Simulation.step (a script that pushes,
        # gathers, deposits or sorts by hand) change the particle data through raw pointers, which
        # torch's version counters do not see: `_ext_gen` counts them, and the state that step()
        # carries from one call to the next is keyed on it (Simulation._carry_signature).
        self._in_step = False
        self._ext_gen = 0
        self._home_valid = False
        self.record_home_in_sort_pass = False     # set by Simulation.step for its sorting iterations
        self.record_home_in_rho_sort = True       # ... and the sort in front of a rho_prev deposition records them too
        self.cycle_sort_period = int(os.environ.get('FBPIC_AMD_SORT_PERIOD', '3'))
        # (round 6: 0.5 instead of 0.12 - the kernel regroups a chunk with more than 12 strays inside the
        # wave, csrc/cycle.hip, so a quarter of the particles changing cell per step, as in a laser wake, no
        # longer makes the pass slower than the two passes of a sorting iteration: C3 0.88 against 0.39 + 0.72
        # ms; the period governs thermal plasmas either way)
        self.cycle_stray_limit = float(os.environ.get('FBPIC_AMD_STRAY_LIMIT', '0.5'))
        # ... or when more than `cycle_bad_limit` of the 64-particle chunks hold > 16 such particles.  If
        # already the FIRST pass after a sort reports that, the next `cycle_suspend_iterations` iterations
        # are two-pass (sorting) ones, then one pass probes again.  Round 5 needed this rule (a wake
        # filled whole regions with strays, every one a gather segment and a scatter of its own: C3 1.3 -
        # 1.9 ms per pass); with the regrouped chunks of round 6 it is OFF by default (limit 2.0) and
        # kept for plasmas whose strays scatter over many cells.
        self.cycle_bad_limit = 2.0
        self.cycle_suspend_iterations = 16
        # The sorting second pass (fb_push_x_sort_deposit_J_rho) runs its two depositions one after the
        # other instead of merged when more than this share of the chunks is full of cell changers (C3:
        # 0.70 against 0.77 - 0.90 ms)
        self.cycle_two_engine_limit = 0.01
        self._cycle_suspended = 0
        self.cycle_bad_fraction = None
        self._cycle_since_sort = 0
        self._cycle_stats = None          # device counters / pinned host copy / pending event
        self.cycle_sorts = 0              # diagnostics: sorts and passes of the one-pass cycle
        self.cycle_passes = 0
        self.cycle_stray_fraction = None  # latest measured share of strays (J deposition)
        self.cycle_last_stray_fraction = None     # ... not reset by a sort (for reports)

    # ---------------------------------------------------------------- host <-> device
    def _alloc_device_helpers(self):
        """Device-only helper buffers.  They are kept with ~6 % of headroom and re-used while the
        particle number fits: a hand-over that moves a few particles between ranks (every
        `exchange_period` steps) does not reallocate 14 N doubles + the sort workspace."""
        t = _capi.torch()
        dev = _capi.require_device()
        Nz, Nr = self.grid_shape
        n = self.Ntot
        ncell = Nz * (Nr + 1)
        cap = n + n // 16 + 1024

        def fit(old, dtype, itemsize):
            # a length-n tensor on the storage of `old` when it is large enough
            if old is not None and old.device == dev and old.dtype == dtype \
                    and old.untyped_storage().nbytes() >= n * itemsize \
                    and old.untyped_storage().nbytes() <= 2 * cap * itemsize + 65536:
                return t.empty(0, dtype=dtype, device=dev).set_(old.untyped_storage(), 0, (n,))
            return t.empty(cap, dtype=dtype, device=dev)[:n]
        self.cell_idx = fit(self.cell_idx, t.int32, 4)
        self.sorted_idx = fit(self.sorted_idx, t.int32, 4)
        self._cell_idx_alt = fit(getattr(self, '_cell_idx_alt', None), t.int32, 4)
        self._sorted_idx_alt = fit(getattr(self, '_sorted_idx_alt', None), t.int32, 4)
        if self.prefix_sum is None or self.prefix_sum.shape[0] != ncell or self.prefix_sum.device != dev:
            self.prefix_sum = t.zeros(ncell, dtype=t.int32, device=dev)
        olds = self._alt if self._alt is not None else [None] * 14
        self._alt = [fit(o, t.float64, 8) for o in olds]
        self.sorting_buffer = self._alt[0]
        nbytes = max(int(_capi.lib().fb_sort_workspace_bytes(n, ncell)),
                     int(_capi.lib().fb_bin_sort_workspace_bytes(n, ncell)))
        if self._sort_ws is None or self._sort_ws.shape[0] < nbytes or self._sort_ws.device != dev \
                or getattr(self, '_sort_ws_n', -1) != n:
            # the workspace layout depends on n (per-particle cell and rank arrays): when n
            # changes the per-cell counters move, so they are no longer known to be zero
            nb_cap = max(int(_capi.lib().fb_sort_workspace_bytes(cap, ncell)),
                         int(_capi.lib().fb_bin_sort_workspace_bytes(cap, ncell)), nbytes)
            if self._sort_ws is None or self._sort_ws.shape[0] < nbytes or self._sort_ws.device != dev:
                self._sort_ws = t.empty(nb_cap, dtype=t.uint8, device=dev)
            self._sort_ws_n = n
            self._counts_clean = False       # per-cell counters of the workspace known to be zero
        if getattr(self, '_nflush', None) is None or self._nflush.device != dev:
            self._nflush = t.zeros(1024, dtype=t.int64, device=dev)
            self._nflush_host = t.zeros(1024, dtype=t.int64).pin_memory()
        self._runs_after_sort = None
        self._runs_latest = None
        self._stat_pending = None

    def send_particles_to_gpu(self):
        if self.data_is_on_gpu:
            return
        # every particle array gets the head-room of the helper buffers (~6 %): the first
        # hand-over that makes a rank's particle number grow then re-uses the storage instead of
        # re-allocating 14 arrays in the middle of a run (particle_buffer_handling._resized)
        t = _capi.torch()
        dev = _capi.require_device()
        n = self.Ntot
        cap = n + n // 16 + 1024
        for k in _STATE + _FIELDS:
            a = getattr(self, k)
            if isinstance(a, t.Tensor):
                setattr(self, k, _capi.to_device(a, dtype=np.float64))
                continue
            host = np.ascontiguousarray(a, dtype=np.float64)
            d = t.empty(cap, dtype=t.float64, device=dev)[:n]
            d.copy_(t.from_numpy(host))
            setattr(self, k, d)
        if self.cell_idx is None or self.cell_idx.shape[0] != self.Ntot:
            self._alloc_device_helpers()
        self.sorted = False
        self._prefix_valid = False
        self._home_valid = False
        self._moved_since_sort = np.inf
        self._pending_push = None
        self._prerank = None
        self._epoch += 1
        self.data_is_on_gpu = True

    def receive_particles_from_gpu(self):
        if not self.data_is_on_gpu:
            return
        if self.__dict__.get('_deferred_fields') is not None:
            self._materialize_fields()
        self.flush_pending_push()
        self._prerank = None
        for k in _STATE + _FIELDS:
            setattr(self, k, _capi.to_host(getattr(self, k)))
        self.data_is_on_gpu = False

    # ---------------------------------------------------------------- deferred E, B
    def defer_fields(self, fld, rmax_gather, dt):
        """Called by Simulation.step after the last iteration of a call whose gather did not
        store E, B on the particles: `fld.d_EB_snap` holds the grids that gather read and the
        positions have since been pushed twice by dt / 2.  Ex ... Bz are taken off the object and
        evaluated by the first read (__getattr__): positions stepped back by the two half
        pushes (same expression, opposite sign: <= 2 ulp from the positions the gather saw), then
        the plain gather kernel on the saved grids - the values the reference leaves in these
        arrays (particles.py:703-800), in the CURRENT order of the particles."""
        if self.q == 0:
            return
        g0 = fld.interp[0]
        # the saved grids in the order the gather wants them: per mode Er, Et, Ez, Br, Bt, Bz
        views = [fld.d_EB_snap[:, fld.interp_index(name, m), :] for m in range(len(fld.interp))
                 for name in ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz')]
        self._deferred_fields = (views, len(fld.interp), rmax_gather,
                                 (g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr), dt)
        self._field_store = [self.__dict__.pop(k) for k in _FIELDS]

    def drop_deferred_fields(self):
        if self.__dict__.get('_deferred_fields') is not None:
            self._deferred_fields = None
            for k, a in zip(_FIELDS, self._field_store):
                setattr(self, k, a)
            self._field_store = None

    def _materialize_fields(self):
        views, Nm, rmax_gather, geom, dt = self._deferred_fields
        self.drop_deferred_fields()
        self.flush_pending_push()
        lib, p, st = _capi.lib(), _capi.ptr, _capi.stream()
        pos = [self.x.clone(), self.y.clone(), self.z.clone()]
        for _ in range(2):
            _capi.check(lib.fb_push_x(self.Ntot, p(pos[0]), p(pos[1]), p(pos[2]), p(self.ux), p(self.uy),
                                      p(self.uz), p(self.inv_gamma), c, -0.5 * dt, 1., 1., 1., st),
                        'fb_push_x')
        invdz, zmin, Nz, invdr, rmin, Nr = geom
        for k in _FIELDS:
            a = getattr(self, k)
            if a.shape[0] != self.Ntot:
                setattr(self, k, _capi.torch().empty(self.Ntot, dtype=a.dtype, device=a.device))
        rc = lib.fb_gather(_SHAPE[self.particle_shape], Nm, self.Ntot, p(pos[0]), p(pos[1]), p(pos[2]),
                           rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, _capi.ptr_array(views),
                           _capi.row_stride(views[0]), p(self.Ex), p(self.Ey), p(self.Ez), p(self.Bx),
                           p(self.By), p(self.Bz), st)
        _capi.check(rc, 'fb_gather')

    def __getattr__(self, name):
        # only reached when the normal lookup fails: Ex ... Bz while defer_fields holds them back
        if name in _FIELDS and self.__dict__.get('_deferred_fields') is not None:
            self._materialize_fields()
            return self.__dict__[name]
        raise AttributeError(name)

    def generate_continuously_injected_particles(self, time):
        """(8, N) float buffer of the plasma uncovered by the moving window since the last
        particle exchange (reference :335-374), in the buffer order x,y,z,ux,uy,uz,inv_gamma,w."""
        assert self.continuous_injection is True
        Ntot, x, y, z, ux, uy, uz, inv_gamma, w = self.injector.generate_particles(time)
        buf = np.empty((self.n_float_quantities, Ntot), dtype=np.float64)
        for i, a in enumerate((x, y, z, ux, uy, uz, inv_gamma, w)):
            buf[i, :] = a
        return buf

    def on_particle_number_changed(self):
        """Re-size the device helpers after particles were added / removed."""
        self.sorted = False
        self._prefix_valid = False
        self._home_valid = False
        self._moved_since_sort = np.inf
        self._prerank = None
        if self.data_is_on_gpu and self.x.is_cuda:
            self._alloc_device_helpers()

    def _touch(self):
        """Start of a public method that modifies (or re-orders) the particle data."""
        if not self._in_step:
            self._ext_gen += 1
            if self.__dict__.get('_deferred_fields') is not None:
                # E, B on the particles that step() left to their first read belong to the
                # positions and momenta as they are now: evaluate them before these change
                self._materialize_fields()

    def _need_gpu(self):
        if not self.data_is_on_gpu:
            raise _capi.BackendError(
                'Particle data is on the host: fbpic_amd only computes on the GPU. Call '
                'send_particles_to_gpu() (or use GpuMemoryManager / Simulation.step) first.')

    def handle_elementary_processes(self, t):
        """Ionization / Compton scattering are outside the hot path: nothing to do."""
        return

    # ---------------------------------------------------------------- pushers
    def push_p(self, t):
        """Vay push of (ux, uy, uz, inv_gamma) by one step (reference :557-636)."""
        self._touch()
        if self.q == 0:
            return
        self._need_gpu()
        p = _capi.ptr
        self._prerank = None
        rc = _capi.lib().fb_push_p(self.Ntot, p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma),
                                   p(self.Ex), p(self.Ey), p(self.Ez), p(self.Bx), p(self.By),
                                   p(self.Bz), self.q, self.m, c, self.dt, _capi.stream())
        _capi.check(rc, 'fb_push_p')

    def push_x(self, dt, x_push=1., y_push=1., z_push=1., defer=False):
        """x += c dt inv_gamma push u (reference :639-671); invalidates the cell sort.
        With `defer` the push is not launched: it is folded into the sort of the next
        `deposit` (fb_push_x_bin_sort_particles), or launched by `flush_pending_push`,
        whichever comes first.  Simulation.step uses it for the half push that precedes
        deposit('rho_next'), main.py:519-528."""
        self._touch()
        self._need_gpu()
        if not (defer and self.use_bin_sort and self._pending_push is None):
            self.flush_pending_push()
        if defer and self.use_bin_sort:
            self._pending_push = (dt, x_push, y_push, z_push)
            if self._prerank != self._pending_push:
                self._prerank = None
            self._note_push(dt, max(abs(x_push), abs(y_push), abs(z_push)))
            return
        self._launch_push_x(dt, x_push, y_push, z_push)
        self._note_push(dt, max(abs(x_push), abs(y_push), abs(z_push)))

    def flush_pending_J(self):
        """Launch a deferred J deposition now (no-op if none is pending)."""
        pj, self._pending_J = self._pending_J, None
        if pj is not None:
            self.defer_J_deposit = False
            # J belongs to the positions BEFORE a deferred push: hide the push from the sort
            # that this deposit may trigger, and count it again if that sort happened
            pend, self._pending_push = self._pending_push, None
            self.deposit(pj[0], 'J', records=pj[1])
            self._pending_push = pend
            if pend is not None and self._moved_since_sort == 0.:
                self._note_push(pend[0], max(abs(pend[1]), abs(pend[2]), abs(pend[3])))

    def flush_pending_push(self):
        """Launch a deferred push_x now (no-op if none is pending); a deferred J deposition
        belongs to the positions before it and goes first."""
        self.flush_pending_J()
        pend = self._pending_push
        if pend is not None:
            self._pending_push = None
            self._launch_push_x(*pend)

    def _note_push(self, dt, push):
        self.sorted = False
        dmin = min(self._cell_size) if self._cell_size else 0.
        self._moved_since_sort += (c * abs(dt) * push / dmin if dmin > 0 else np.inf)

    def _launch_push_x(self, dt, x_push, y_push, z_push):
        p = _capi.ptr
        self._prerank = None
        rc = _capi.lib().fb_push_x(self.Ntot, p(self.x), p(self.y), p(self.z), p(self.ux),
                                   p(self.uy), p(self.uz), p(self.inv_gamma), c, dt,
                                   x_push, y_push, z_push, _capi.stream())
        _capi.check(rc, 'fb_push_x')

    # ---------------------------------------------------------------- gather
    def gather(self, grid, comm):
        """Interpolate E, B of all modes onto the particles (reference :673-837)."""
        self._touch()
        if self.q == 0:
            return
        self._need_gpu()
        self.drop_deferred_fields()
        self.flush_pending_push()
        Nm = len(grid)
        rmax_gather = comm.get_rmax(with_damp=False)
        g0 = grid[0]
        views = []
        for m in range(Nm):
            views += [grid[m].Er, grid[m].Et, grid[m].Ez, grid[m].Br, grid[m].Bt, grid[m].Bz]
        p = _capi.ptr
        rc = _capi.lib().fb_gather(_SHAPE[self.particle_shape], Nm, self.Ntot,
                                   p(self.x), p(self.y), p(self.z), rmax_gather,
                                   g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                                   _capi.ptr_array(views), _capi.row_stride(views[0]),
                                   p(self.Ex), p(self.Ey), p(self.Ez), p(self.Bx), p(self.By),
                                   p(self.Bz), _capi.stream())
        _capi.check(rc, 'fb_gather')

    def can_split_gather(self, Nm):
        """The range-restricted gather + push (gather_push(part=...)) needs cell-sorted arrays with
        an exact prefix sum, the ranking pass, and the lane-by-lane gather kernel."""
        return bool(self.sorted and self._prefix_valid and self.use_bin_sort and self.Ntot > 0
                    and self.q != 0 and not (self.particle_shape == 'cubic' and 2 <= Nm <= 4))

    def gather_push(self, grid, comm, dt_x, store_fields=True, wrap_z=None, rank_next=None,
                    part=None, rows=None):
        """gather -> push_p -> push_x(dt_x) in one pass (fb_gather_push): the fused form of
        the three consecutive calls of Simulation.step (main.py:469-490).  Results are
        identical to calling gather(), push_p(), push_x(dt_x) one after the other.
        `rank_next` = (dt, x_push, y_push, z_push) of the push_x that will follow: the pass also
        ranks the particles for the sort after that push (fb_gather_push_rank_next).
        `part` = 'inside' / 'outside' with `rows` = (a, b): only the particles whose (sorted) cell
        row is / is not in [a, b) - Simulation.step runs the inside part while the guard-cell
        exchange of E, B is in flight and the outside part after it (can_split_gather)."""
        self._touch()
        self._need_gpu()
        self.drop_deferred_fields()
        self.flush_pending_push()
        if self.q == 0:
            if part == 'inside':
                return
            if wrap_z is not None:
                rc = _capi.lib().fb_shift_periodic(self.Ntot, _capi.ptr(self.z), float(wrap_z[0]),
                                                   float(wrap_z[1]), _capi.stream())
                _capi.check(rc, 'fb_shift_periodic')
            self.push_x(dt_x)
            return
        Nm = len(grid)
        g0 = grid[0]
        views = []
        for m in range(Nm):
            views += [grid[m].Er, grid[m].Et, grid[m].Ez, grid[m].Br, grid[m].Bt, grid[m].Bz]
        p = _capi.ptr
        eb = [p(getattr(self, k)) if store_fields else None for k in _FIELDS]
        # wrap_z = (zmin, zmax): the periodic wrap of comm.exchange_particles folded in
        wz = (0., 0.) if wrap_z is None else (float(wrap_z[0]), float(wrap_z[1]))
        ranked = (rank_next is not None and self.use_bin_sort and self.Ntot > 0 and dt_x != 0.
                  and g0.Nz * (g0.Nr + 1) == self.prefix_sum.shape[0])
        if part is not None:
            assert ranked and self.can_split_gather(Nm) and part in ('inside', 'outside')
            ncol = g0.Nr + 1
            a = min(max(rows[0] + self.prefix_sum_shift, 0), g0.Nz)
            b = min(max(rows[1] + self.prefix_sum_shift, a), g0.Nz)
            ps, es = self.prefix_sum.data_ptr(), self.prefix_sum.element_size()
            lo_ptr = None if a == 0 else ps + (a * ncol - 1) * es
            hi_ptr = None if b == 0 else ps + (b * ncol - 1) * es
            clean = int(self._counts_clean) if part == 'inside' else 1
            rc = _capi.lib().fb_gather_push_rank_next_range(
                _SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x), p(self.y), p(self.z),
                p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma),
                comm.get_rmax(with_damp=False), g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                _capi.ptr_array(views), _capi.row_stride(views[0]), *eb,
                self.q, self.m, c, self.dt, dt_x, wz[0], wz[1],
                rank_next[0], rank_next[1], rank_next[2], rank_next[3], self.prefix_sum.shape[0],
                p(self._sort_ws), self._sort_ws.shape[0], clean, lo_ptr, hi_ptr,
                1 if part == 'inside' else 2, _capi.stream())
            self._counts_clean = False
            _capi.check(rc, 'fb_gather_push_rank_next_range')
            if part == 'inside':
                return                      # the book-keeping below belongs to the completed pass
        elif ranked and self._home_valid and rank_next == (dt_x, 1., 1., 1.) \
                and self._home_shift(g0) is not None and self.particle_shape == 'linear' \
                and _capi.lib().fb_gather_push_deposit_supported(_SHAPE[self.particle_shape], Nm):
            # arrays sorted some steps ago (a sorting iteration of the one-pass cycle): segments
            # from the home cells, which a stale order does not fragment
            rc = _capi.lib().fb_gather_push_rank_next_home(
                _SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x), p(self.y), p(self.z),
                p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma), p(self.cell_idx),
                comm.get_rmax(with_damp=False), g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                _capi.ptr_array(views), _capi.row_stride(views[0]), *eb,
                self.q, self.m, c, self.dt, dt_x, wz[0], wz[1], self.prefix_sum.shape[0],
                p(self._sort_ws), self._sort_ws.shape[0], int(self._counts_clean),
                self._home_shift(g0), _capi.stream())
            self._counts_clean = False
            _capi.check(rc, 'fb_gather_push_rank_next_home')
        elif ranked:
            rc = _capi.lib().fb_gather_push_rank_next(
                _SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x), p(self.y), p(self.z),
                p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma),
                comm.get_rmax(with_damp=False), g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                _capi.ptr_array(views), _capi.row_stride(views[0]), *eb,
                self.q, self.m, c, self.dt, dt_x, wz[0], wz[1],
                rank_next[0], rank_next[1], rank_next[2], rank_next[3], self.prefix_sum.shape[0],
                p(self._sort_ws), self._sort_ws.shape[0], int(self._counts_clean), _capi.stream())
            self._counts_clean = False
            _capi.check(rc, 'fb_gather_push_rank_next')
        else:
            rc = _capi.lib().fb_gather_push(
                _SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x), p(self.y), p(self.z),
                p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma),
                comm.get_rmax(with_damp=False), g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                _capi.ptr_array(views), _capi.row_stride(views[0]), *eb,
                self.q, self.m, c, self.dt, dt_x, wz[0], wz[1], _capi.stream())
            _capi.check(rc, 'fb_gather_push')
        self._prerank = tuple(rank_next) if ranked else None
        self.sorted = False
        dmin = min(self._cell_size) if self._cell_size else 0.
        self._moved_since_sort += (c * abs(dt_x) / dmin if dmin > 0 else np.inf)

    # ---------------------------------------------------------------- one-pass cycle
    def cycle_supported(self, Nm):
        """Whether Particles.cycle has a kernel for this species (else: the separate calls)."""
        return bool(self.use_bin_sort and not self.is_tracer
                    and (self.q == 0 or _capi.lib().fb_gather_push_deposit_supported(
                        _SHAPE[self.particle_shape], Nm)))

    def _home_shift(self, g0):
        """(Cells the grid has advanced since the home cells were recorded) x (Nr + 1) - the moving
        window translates the grid by whole cells (boundaries/moving_window.py:60-239), so the
        recorded cell of a particle that stays where it is moves that many rows down and the kernels
        subtract the product from every home cell.  None: the recorded cells belong to another grid
        (re-sort)."""
        hg = getattr(self, '_home_geom', None)
        if hg is None or tuple(hg[1:]) != (g0.invdz, g0.Nz, g0.rmin, g0.invdr, g0.Nr):
            return None
        d = (g0.zmin - hg[0]) * g0.invdz
        n_move = int(round(d))
        if abs(d - n_move) > 1.e-6 or abs(n_move) >= g0.Nz:
            return None
        return n_move * (g0.Nr + 1)

    def _cycle_poll(self, lag=1):
        """Take in the counters of earlier passes, in order, leaving the `lag` most recent read-backs
        in flight.  A read-back is issued right behind its pass and its event is WAITED for here, at a
        fixed distance (Particles.cycle polls with lag 1 in front of a pass: the decision of iteration k
        rests on the passes up to k - 2, and the host can still run two iterations ahead of the device)
        - polling with query() made the sequence of one- and two-pass iterations, and with it the
        summation order of J and rho, depend on host timing."""
        st = self._cycle_stats
        while st is not None and len(st[2]) > lag:
            ev, ntot, r, buf, gen = st[2].pop(0)
            ev.synchronize()
            strays, bad = int(buf[:512].sum()), int(buf[512:].sum())
            f_stray = float(strays - st[3][0]) / max(ntot, 1)
            f_bad = float(bad - st[3][1]) / max((ntot + 63) // 64, 1)
            st[3] = (strays, bad)
            self.cycle_last_stray_fraction = f_stray
            if gen == self.cycle_sorts:
                # (a pass of the order before the latest sort says nothing about the present one)
                self.cycle_stray_fraction, self.cycle_bad_fraction = f_stray, f_bad
            if r == 1 and f_bad > self.cycle_bad_limit:
                self._cycle_suspended = self.cycle_suspend_iterations

    def _after_home_sort(self):
        """Book-keeping of a sort that has recorded the home cells in `cell_idx`."""
        self._cycle_since_sort = 0
        self.cycle_stray_fraction = None
        self.cycle_bad_fraction = None
        self.cycle_sorts += 1

    def cycle_wants_sort(self, fld):
        """True when the next Particles.cycle of this species would start with a sort (Simulation.step
        then runs the two-pass sequence for the iteration, whose second pass sorts).  Called ONCE per
        iteration and species (Simulation.step asks every species, without short-circuit): it is the
        iteration tick of the policy - it takes in the pending counter read-backs and counts the
        suspension window down."""
        if not (self.q != 0 and self.Ntot > 0):
            return False
        self._cycle_poll()
        if self._cycle_suspended > 0:
            self._cycle_suspended -= 1
            return True
        return bool(self._cycle_needs_sort(fld.interp[0]))

    def _cycle_needs_sort(self, g0):
        if not self._home_valid or self._cycle_since_sort >= self.cycle_sort_period:
            return True
        if self._home_shift(g0) is None:
            return True                      # another grid: the recorded cells are not cells any more
        self._cycle_poll()
        if self.cycle_stray_fraction is not None and self.cycle_stray_fraction > self.cycle_stray_limit:
            return True
        return self.cycle_bad_fraction is not None and self.cycle_bad_fraction > self.cycle_bad_limit

    def cycle(self, fld, comm, dt, store_fields=True, wrap_z=None):
        """gather -> push_p -> push_x(dt/2) -> deposit('J') -> push_x(dt/2) -> deposit('rho') of
        Simulation.step (main.py:469-528) in ONE pass over the particles
        (fb_gather_push_deposit_J_rho): identical to the separate calls, every attribute read and
        written once.  J and rho go to the node-major records of `fld` (already erased by the
        caller).  The arrays are re-sorted first when the policy asks for it (see __init__)."""
        self._touch()
        self._need_gpu()
        self.drop_deferred_fields()
        self.flush_pending_push()
        lib, p, st = _capi.lib(), _capi.ptr, _capi.stream()
        if self.q == 0 or self.Ntot == 0:
            # no gather, no momentum push, no deposition (reference :575, :697, :866)
            if wrap_z is not None and self.Ntot > 0:
                _capi.check(lib.fb_shift_periodic(self.Ntot, p(self.z), float(wrap_z[0]),
                                                  float(wrap_z[1]), st), 'fb_shift_periodic')
            self.push_x(0.5 * dt)
            self.push_x(0.5 * dt)
            return
        grid = fld.interp
        Nm = len(grid)
        g0 = grid[0]
        if self._cycle_needs_sort(g0):
            # (Simulation.step avoids this stand-alone sort: an iteration that needs one runs the
            # two-pass sequence, whose second pass sorts and records the home cells as it goes)
            kfs, self.keep_fields_sorted = self.keep_fields_sorted, False
            self.sort_particles(fld, record_home=True)
            self.keep_fields_sorted = kfs
            self.sorted = True
            self._after_home_sort()
        t = _capi.torch()
        if self._cycle_stats is None or self._cycle_stats[0].device != self.x.device:
            # device counters, ring of pinned host copies, pending reads (event, Ntot, passes since the
            # sort, host copy) oldest first, totals (strays, bad chunks) last read, passes measured
            self._cycle_stats = [t.zeros(1024, dtype=t.int64, device=self.x.device),
                                 [t.zeros(1024, dtype=t.int64).pin_memory() for _ in range(3)], [], (0, 0), 0]
        views = []
        for m in range(Nm):
            views += [grid[m].Er, grid[m].Et, grid[m].Ez, grid[m].Br, grid[m].Bt, grid[m].Bz]
        eb = [p(getattr(self, k)) if store_fields else None for k in _FIELDS]
        wz = (0., 0.) if wrap_z is None else (float(wrap_z[0]), float(wrap_z[1]))
        if self.particle_shape == 'linear':
            # the node-major records (one cache line per node: a linear flush is one atomic instruction)
            jv, rv = fld.record_views('J'), fld.record_views('rho')
        else:
            # cubic shape: the J | rho fields of the interpolation slab (16 nodes per cell either way)
            jv = []
            for m in range(Nm):
                jv += [grid[m].Jr, grid[m].Jt, grid[m].Jz]
            rv = [grid[m].rho for m in range(Nm)]
        suffix = 'linear' if self.particle_shape == 'linear' else 'cubic'
        ruy0 = getattr(grid[0], 'd_ruyten_%s_coef' % suffix)
        ruyh = getattr(grid[1 if Nm > 1 else 0], 'd_ruyten_%s_coef' % suffix)
        stats = self._cycle_stats
        # (the counters are cumulative - the pass adds to them; one small copy per pass, no memset)
        self._cycle_poll()
        measure = True
        rc = lib.fb_gather_push_deposit_J_rho(
            _SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x), p(self.y), p(self.z),
            p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma), p(self.w), p(self.cell_idx),
            comm.get_rmax(with_damp=False), g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
            _capi.ptr_array(views), _capi.row_stride(views[0]), *eb,
            self.q, self.m, c, self.dt, 0.5 * dt, wz[0], wz[1],
            _capi.ptr_array(jv), jv[0].stride(0), jv[0].stride(1),
            _capi.ptr_array(rv), rv[0].stride(0), rv[0].stride(1), p(ruy0), p(ruyh),
            p(stats[0]) if measure else None, self._home_shift(g0), st)
        _capi.check(rc, 'fb_gather_push_deposit_J_rho')
        if measure:
            buf = stats[1][stats[4] % 3]           # (at most two read-backs are in flight)
            stats[4] += 1
            buf.copy_(stats[0], non_blocking=True)
            ev = t.cuda.Event()
            ev.record()
            stats[2].append((ev, self.Ntot, self._cycle_since_sort + 1, buf, self.cycle_sorts))
        self._cycle_since_sort += 1
        self.cycle_passes += 1
        self._prerank = None
        self.sorted = False
        dmin = min(self._cell_size) if self._cell_size else 0.
        self._moved_since_sort += (c * abs(dt) / dmin if dmin > 0 else np.inf)

    # ---------------------------------------------------------------- sort
    def sort_particles(self, fld, record_home=False):
        """Cell index -> stable radix sort -> per-cell prefix sum -> permutation
        (reference :1049-1094).  `record_home`: `cell_idx` receives the cell of every sorted
        particle (the home cells of Particles.cycle)."""
        self._touch()
        self._need_gpu()
        self._home_valid = False
        self.flush_pending_J()          # a deferred J deposition belongs to the unsorted state
        g0 = fld.interp[0]
        lib = _capi.lib()
        p = _capi.ptr
        st = _capi.stream()
        if self.use_bin_sort:
            # counting sort specialised for the almost-sorted stream: 3 launches in all
            names = list(_STATE) + (list(_FIELDS) if self.keep_fields_sorted else [])
            src = [getattr(self, k) for k in names]
            dst = self._alt[:len(names)]
            p_cell = p(self.cell_idx) if (self.keep_sort_outputs or record_home) else None
            p_sidx = p(self.sorted_idx) if self.keep_sort_outputs else None
            pend, self._pending_push = self._pending_push, None
            preranked = int(pend is not None and self._prerank == pend)
            self._prerank = None
            if pend is not None:
                # the deferred push_x rides along: positions are written once, sorted
                rc = lib.fb_push_x_bin_sort_particles(
                    self.Ntot, self.prefix_sum.shape[0], p(self.x), p(self.y), p(self.z),
                    p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma), c, pend[0], pend[1],
                    pend[2], pend[3], g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                    len(names), _capi.ptr_array(src), _capi.ptr_array(dst), p_cell, p_sidx,
                    p(self.prefix_sum), p(self._sort_ws),
                    self._sort_ws.shape[0], preranked, st)
                _capi.check(rc, 'fb_push_x_bin_sort_particles')
            else:
                rc = lib.fb_bin_sort_particles(
                    self.Ntot, self.prefix_sum.shape[0], p(self.x), p(self.y), p(self.z),
                    g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr, len(names),
                    _capi.ptr_array(src), _capi.ptr_array(dst), p_cell, p_sidx,
                    p(self.prefix_sum), p(self._sort_ws),
                    self._sort_ws.shape[0], st)
                _capi.check(rc, 'fb_bin_sort_particles')
            self._counts_clean = True    # the scatter pass leaves the counters zeroed
            self._prefix_valid = True
            for i, k in enumerate(names):
                setattr(self, k, dst[i])
                self._alt[i] = src[i]
            self.sorting_buffer = self._alt[0]
            self.prefix_sum_shift = 0
            self._cell_size = (g0.dz, g0.dr)
            self._moved_since_sort = 0.
            self._home_valid = bool(record_home)
            self._home_geom = (g0.zmin, g0.invdz, g0.Nz, g0.rmin, g0.invdr, g0.Nr)
            return
        self.flush_pending_push()
        self._prerank = None
        rc = lib.fb_cell_index(self.Ntot, p(self.x), p(self.y), p(self.z), g0.invdz, g0.zmin,
                               g0.Nz, g0.invdr, g0.rmin, g0.Nr, p(self.cell_idx),
                               p(self.sorted_idx), st)
        _capi.check(rc, 'fb_cell_index')
        in_alt = ctypes.c_int(0)
        rc = lib.fb_sort_by_cell(self.Ntot, self.prefix_sum.shape[0], p(self.cell_idx),
                                 p(self.sorted_idx), p(self._cell_idx_alt),
                                 p(self._sorted_idx_alt), ctypes.byref(in_alt),
                                 p(self.prefix_sum), p(self._sort_ws), self._sort_ws.shape[0], st)
        _capi.check(rc, 'fb_sort_by_cell')
        self._counts_clean = False       # the radix sort used the workspace as scratch
        if in_alt.value:     # the radix sort left its result in the alternate buffers
            self.cell_idx, self._cell_idx_alt = self._cell_idx_alt, self.cell_idx
            self.sorted_idx, self._sorted_idx_alt = self._sorted_idx_alt, self.sorted_idx
        self.prefix_sum_shift = 0
        self._prefix_valid = True
        self._cell_size = (g0.dz, g0.dr)
        self._moved_since_sort = 0.
        self.rearrange_particle_arrays()
        self._home_valid = True           # the radix sort leaves the sorted keys in cell_idx
        self._home_geom = (g0.zmin, g0.invdz, g0.Nz, g0.rmin, g0.invdr, g0.Nr)

    def _push_sort_deposit_rho(self, fld, records):
        """fb_push_x_sort_deposit_rho: the deferred push_x, the counting sort and the charge
        deposition in one pass over the particles (destination-ordered)."""
        grid = fld.interp
        Nm = len(grid)
        g0 = grid[0]
        lib, p, st = _capi.lib(), _capi.ptr, _capi.stream()
        names = list(_STATE) + (list(_FIELDS) if self.keep_fields_sorted else [])
        src = [getattr(self, k) for k in names]
        dst = self._alt[:len(names)]
        self._home_valid = False
        record_home = self.record_home_in_sort_pass
        pj = self._pending_J
        if pj is not None and (pj[0] is not fld or pj[1] != records):
            # different target: J on its own, then as usual.  That deposit may re-sort the
            # particles (new order, workspace overwritten): it goes BEFORE the ranks of the
            # pending push are looked at, and flush_pending_J / sort_particles drop them
            self.flush_pending_J()
            pj = None
        self._pending_J = None
        pend, self._pending_push = self._pending_push, None
        preranked = int(pend is not None and self._prerank == pend)
        self._prerank = None
        suffix = 'linear' if self.particle_shape == 'linear' else 'cubic'
        ruy0 = getattr(grid[0], 'd_ruyten_%s_coef' % suffix)
        ruyh = getattr(grid[1 if Nm > 1 else 0], 'd_ruyten_%s_coef' % suffix)
        views = fld.record_views('rho') if records else [grid[m].rho for m in range(Nm)]
        if pj is not None:
            if records:
                jviews = fld.record_views('J')
            else:
                jviews = []
                for m in range(Nm):
                    jviews += [grid[m].Jr, grid[m].Jt, grid[m].Jz]
            rc = lib.fb_push_x_sort_deposit_J_rho(
                self.Ntot, self.prefix_sum.shape[0], p(self.x), p(self.y), p(self.z),
                p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma), c, pend[0], pend[1], pend[2],
                pend[3], g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                len(names), _capi.ptr_array(src), _capi.ptr_array(dst),
                p(self.cell_idx) if (self.keep_sort_outputs or record_home) else None, p(self.sorted_idx),
                p(self.prefix_sum), p(self._sort_ws), self._sort_ws.shape[0], preranked,
                _SHAPE[self.particle_shape], Nm, self.q, g0.zmin, _capi.ptr_array(jviews),
                jviews[0].stride(0), jviews[0].stride(1), _capi.ptr_array(views),
                views[0].stride(0), views[0].stride(1), p(ruy0), p(ruyh),
                # (a plasma that has just turned the one-pass form off - whole chunks of particles that
                # change cell every step - also wants the two depositions of this pass one after the other)
                1 if (self._cycle_suspended > 0 or (self.cycle_bad_fraction or 0.) > self.cycle_two_engine_limit) else 0,
                st)
            _capi.check(rc, 'fb_push_x_sort_deposit_J_rho')
        else:
            rc = lib.fb_push_x_sort_deposit_rho(
                self.Ntot, self.prefix_sum.shape[0], p(self.x), p(self.y), p(self.z),
                p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma), c, pend[0], pend[1], pend[2],
                pend[3], g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                len(names), _capi.ptr_array(src), _capi.ptr_array(dst),
                p(self.cell_idx) if (self.keep_sort_outputs or record_home) else None, p(self.sorted_idx),
                p(self.prefix_sum), p(self._sort_ws), self._sort_ws.shape[0], preranked,
                _SHAPE[self.particle_shape], Nm, self.q, _capi.ptr_array(views),
                views[0].stride(0), views[0].stride(1), p(ruy0), p(ruyh), st)
            _capi.check(rc, 'fb_push_x_sort_deposit_rho')
        self._counts_clean = True
        self._prefix_valid = True
        for i, k in enumerate(names):
            setattr(self, k, dst[i])
            self._alt[i] = src[i]
        self.sorting_buffer = self._alt[0]
        self.prefix_sum_shift = 0
        self._cell_size = (g0.dz, g0.dr)
        self._moved_since_sort = 0.
        self.sorted = True
        self._deposits_since_sort = 0
        self._runs_latest = None
        if record_home:
            # `cell_idx` now holds the cell of every particle at its sorted slot
            self._home_valid = True
            self._home_geom = (g0.zmin, g0.invdz, g0.Nz, g0.rmin, g0.invdr, g0.Nr)
            self._after_home_sort()

    def rearrange_particle_arrays(self):
        """Apply sorted_idx to every particle attribute in one launch (ping-pong buffers)."""
        self._touch()
        self._home_valid = False
        names = list(_STATE)
        if self.keep_fields_sorted:
            names += list(_FIELDS)
        src = [getattr(self, k) for k in names]
        dst = self._alt[:len(names)]
        rc = _capi.lib().fb_permute(self.Ntot, _capi.ptr(self.sorted_idx), len(names),
                                    _capi.ptr_array(src), _capi.ptr_array(dst), _capi.stream())
        _capi.check(rc, 'fb_permute')
        for i, k in enumerate(names):
            setattr(self, k, dst[i])
            self._alt[i] = src[i]
        self.sorting_buffer = self._alt[0]

    # ---------------------------------------------------------------- sort policy
    def _poll_stats(self):
        """Pick up the run count of an earlier J deposition if its copy has landed."""
        if self._stat_pending is not None:
            ev, was_fresh = self._stat_pending
            if ev.query():
                runs = int(self._nflush_host.sum())
                if was_fresh:
                    self._runs_after_sort = runs
                self._runs_latest = runs
                self._stat_pending = None

    def _needs_sort(self):
        if self._moved_since_sort <= self.sort_tolerance:
            return False
        if not (self.resort_fragmentation > 0) or self._moved_since_sort == np.inf:
            return True
        if self._deposits_since_sort >= self.max_deposits_between_sorts:
            return True
        self._poll_stats()
        if self._runs_after_sort is None or self._runs_latest is None:
            return True          # no measurement yet: obey the displacement bound
        return self._runs_latest > self.resort_fragmentation * self._runs_after_sort

    def _post_deposit_stats(self):
        """Asynchronous read-back of the run counter written by fb_deposit_J."""
        self._poll_stats()
        if self._stat_pending is None:
            t = _capi.torch()
            self._nflush_host.copy_(self._nflush, non_blocking=True)
            ev = t.cuda.Event()
            ev.record()
            self._stat_pending = (ev, self._deposits_since_sort == 0)
        self._nflush.zero_()
        self._deposits_since_sort += 1

    # ---------------------------------------------------------------- deposit
    def deposit(self, fld, fieldtype, records=False):
        """Deposit rho or J of this species on the interpolation grid (reference :839-1046).
        `records` (Simulation.step): deposit into the node-major record array of the Fields
        object instead (Fields.source_records), which the z-FFT reads directly."""
        self._touch()
        if self.q == 0:
            return
        assert fieldtype in ['rho', 'J']
        self._need_gpu()
        if fieldtype == 'J':
            defer, self.defer_J_deposit = self.defer_J_deposit, False
            if (defer and self.fuse_sort_deposit_rho and self.use_bin_sort and self.Ntot > 0
                    and len(fld.interp) <= 4 and self._pending_push is None
                    and (self.particle_shape == 'linear' or _FUSE_J_CUBIC)):
                self._pending_J = (fld, records)          # rides along in the rho deposition
                self.push_after_deposit_J = None
                return
        if (fieldtype == 'rho' and self.fuse_sort_deposit_rho and self.use_bin_sort
                and self._pending_push is not None and not self.sorted and self._needs_sort()
                and self.Ntot > 0):
            # the step's push_x(dt/2) -> re-sort -> deposit('rho_next'): one pass
            self._push_sort_deposit_rho(fld, records)
            return
        if fieldtype == 'rho':
            self.flush_pending_J()
        if not self.sorted and self._needs_sort():
            # (inside step() the sort also records the home cells of the one-pass cycle: the rho_prev
            # deposition that follows a particle hand-over then leaves a FRESH order behind, and the
            # iteration runs one pass instead of a second, sorting pair - C3: 0.86 against 1.13 ms)
            home = bool(self._in_step and self.use_bin_sort and fieldtype == 'rho' and self.record_home_in_rho_sort)
            self.sort_particles(fld=fld, record_home=home)
            self.sorted = True
            self._deposits_since_sort = 0
            self._runs_latest = None
            if home and self._home_valid:
                self._after_home_sort()
        self.flush_pending_push()
        grid = fld.interp
        Nm = len(grid)
        g0 = grid[0]
        weight = self.w
        suffix = 'linear' if self.particle_shape == 'linear' else 'cubic'
        ruy0 = getattr(grid[0], 'd_ruyten_%s_coef' % suffix)
        ruyh = getattr(grid[1 if Nm > 1 else 0], 'd_ruyten_%s_coef' % suffix)
        lib = _capi.lib()
        p = _capi.ptr
        adaptive = self.resort_fragmentation > 0
        if fieldtype == 'rho':
            views = fld.record_views('rho') if records else [grid[m].rho for m in range(Nm)]
            rc = lib.fb_deposit_rho(_SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x),
                                    p(self.y), p(self.z), p(weight), self.q, g0.invdz, g0.zmin,
                                    g0.Nz, g0.invdr, g0.rmin, g0.Nr, _capi.ptr_array(views),
                                    views[0].stride(0), views[0].stride(1),
                                    p(self.prefix_sum), p(ruy0),
                                    p(ruyh), None, _capi.stream())
            _capi.check(rc, 'fb_deposit_rho')
        else:
            views = []
            for m in range(Nm):
                views += [grid[m].Jr, grid[m].Jt, grid[m].Jz]
            if records:
                views = fld.record_views('J')
            hint, self.push_after_deposit_J = self.push_after_deposit_J, None
            if hint is not None and self._prerank == tuple(hint):
                hint = None                 # already ranked for that push (gather_push)
            if hint is not None and self.use_bin_sort and self.Ntot > 0:
                # the deposition also ranks the particles for the sort that follows `hint`
                rc = lib.fb_deposit_J_rank_next(
                    _SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x), p(self.y), p(self.z),
                    p(weight), self.q, p(self.ux), p(self.uy), p(self.uz), p(self.inv_gamma), c,
                    g0.invdz, g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr, _capi.ptr_array(views),
                    views[0].stride(0), views[0].stride(1), p(ruy0), p(ruyh),
                    p(self._nflush) if adaptive else None, hint[0], hint[1], hint[2], hint[3],
                    self.prefix_sum.shape[0], p(self._sort_ws), self._sort_ws.shape[0],
                    int(self._counts_clean), _capi.stream())
                self._counts_clean = False
                _capi.check(rc, 'fb_deposit_J_rank_next')
                self._prerank = tuple(hint)
            else:
                rc = lib.fb_deposit_J(_SHAPE[self.particle_shape], Nm, self.Ntot, p(self.x),
                                      p(self.y), p(self.z), p(weight), self.q, p(self.ux),
                                      p(self.uy), p(self.uz), p(self.inv_gamma), c, g0.invdz,
                                      g0.zmin, g0.Nz, g0.invdr, g0.rmin, g0.Nr,
                                      _capi.ptr_array(views), views[0].stride(0),
                                      views[0].stride(1), p(self.prefix_sum), p(ruy0), p(ruyh),
                                      p(self._nflush) if adaptive else None, _capi.stream())
                _capi.check(rc, 'fb_deposit_J')
            if adaptive:
                self._post_deposit_stats()
