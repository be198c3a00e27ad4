"""`Simulation`: the PIC-cycle driver, same interface as the reference
(fbpic/main.py:51-1111) for the hot path: constructor arguments, `step`, `deposit`,
`exchange_and_damp_EB`, `add_new_species`, attributes `fld`, `ptcl`, `comm`, `time`,
`iteration`, `diags`.  The order of operations in `step` follows main.py:346-586
exactly; every operation is a HIP kernel launch (fbpic_amd has no CPU path)."""
import os
import numpy as np
from scipy.constants import m_e, m_p, e, c
from . import _capi
from .particles.particles import Particles
from .fields import Fields
from .boundaries.boundary_communicator import BoundaryCommunicator
from .boundaries.moving_window import MovingWindow


def send_data_to_gpu(simulation):
    """fbpic/utils/cuda.py:101-118"""
    for species in simulation.ptcl:
        species.send_particles_to_gpu()
    simulation.fld.send_fields_to_gpu()


def receive_data_from_gpu(simulation):
    """fbpic/utils/cuda.py:120-137"""
    for species in simulation.ptcl:
        species.receive_particles_from_gpu()
    simulation.fld.receive_fields_from_gpu()


class GpuMemoryManager(object):
    """Context manager that keeps the simulation data on the GPU for its duration
    (fbpic/utils/cuda.py:139-182); inside it `Simulation.step` does not copy anything."""

    def __init__(self, simulation):
        self.sim = simulation
        self.fields_were_on_gpu = simulation.fld.data_is_on_gpu
        self.species_were_on_gpu = [s.data_is_on_gpu for s in simulation.ptcl]

    def __enter__(self):
        if not self.fields_were_on_gpu:
            self.sim.fld.send_fields_to_gpu()
        for i, s in enumerate(self.sim.ptcl):
            if not self.species_were_on_gpu[i]:
                s.send_particles_to_gpu()
        return self.sim

    def __exit__(self, type, value, traceback):
        if not self.fields_were_on_gpu:
            self.sim.fld.receive_fields_from_gpu()
        for i, s in enumerate(self.sim.ptcl):
            if i >= len(self.species_were_on_gpu) or not self.species_were_on_gpu[i]:
                s.receive_particles_from_gpu()


_VERSION_COUNTER_OK = []


def _version_counter_works():
    """The carried state relies on `Tensor._version` (a private torch attribute) to notice that
    the user wrote into a state tensor between two step() calls.  Checked once per process: the
    attribute exists and an in-place write, a slice assignment and a copy_ each advance it; if
    a torch build ever behaves differently, nothing is carried (every call starts like the
    reference's)."""
    if not _VERSION_COUNTER_OK:
        import warnings
        ok = False
        try:
            import torch
            t = torch.zeros(4, dtype=torch.float64)
            v0 = t._version
            t += 1.
            v1 = t._version
            t[1:3] = 2.
            v2 = t._version
            t.copy_(torch.ones(4, dtype=torch.float64))
            v3 = t._version
            ok = v0 < v1 < v2 < v3
        except Exception:
            ok = False
        if not ok:
            warnings.warn('torch.Tensor._version does not track in-place writes in this torch build: '
                          'fbpic_amd does not carry device state between step() calls')
        _VERSION_COUNTER_OK.append(ok)
    return _VERSION_COUNTER_OK[0]


class Simulation(object):
    def __init__(self, Nz, zmax, Nr, rmax, Nm, dt,
                 p_zmin=-np.inf, p_zmax=np.inf, p_rmin=0, p_rmax=np.inf,
                 p_nz=None, p_nr=None, p_nt=None, n_e=None, zmin=0.,
                 n_order=-1, dens_func=None, filter_currents=True,
                 v_comoving=None, use_galilean=True,
                 initialize_ions=False, use_cuda=True, n_guard=None,
                 n_damp={'z': 64, 'r': 32}, exchange_period=None,
                 current_correction='curl-free',
                 boundaries={'z': 'periodic', 'r': 'reflective'},
                 gamma_boost=None, use_all_mpi_ranks=True,
                 particle_shape='linear', verbose_level=1,
                 smoother=None, use_ruyten_shapes=True, use_modified_volume=True):
        if not use_cuda:
            raise ValueError('fbpic_amd executes only on the GPU (use_cuda=True); the CPU '
                             'path of the reference is not part of this backend.')
        if gamma_boost is not None:
            raise NotImplementedError('boosted-frame input conversion (gamma_boost) is outside '
                                      'the scope of fbpic_amd: pass boosted-frame quantities')
        self.use_cuda = True
        self.use_threading = False
        self.cpu_threads = 1
        # Galilean / comoving-current PSATD (main.py:269-273)
        self.v_comoving = v_comoving
        self.use_galilean = use_galilean if v_comoving is not None else False
        self.boost = None
        self.dt = dt
        cdt_over_dr = c * dt / (rmax / Nr)
        self.comm = BoundaryCommunicator(Nz, zmin, zmax, Nr, rmax, Nm, dt, self.v_comoving,
                                         self.use_galilean, boundaries, n_order, n_guard, n_damp, cdt_over_dr,
                                         None, exchange_period, use_all_mpi_ranks)
        self.use_pml = self.comm.use_pml
        zmin, zmax, Nz = self.comm.divide_into_domain()
        Nr = self.comm.get_Nr(with_damp=True)
        rmax = self.comm.get_rmax(with_damp=True)
        self.fld = Fields(Nz, zmax, Nr, rmax, Nm, dt, n_order=n_order, zmin=zmin,
                          v_comoving=self.v_comoving, use_galilean=self.use_galilean,
                          current_correction=current_correction, use_cuda=True,
                          smoother=smoother, use_ruyten_shapes=use_ruyten_shapes,
                          use_modified_volume=use_modified_volume)
        self.grid_shape = self.fld.interp[0].Ez.shape
        self.particle_shape = particle_shape
        # rank the particles for the next sort inside the J deposition (fb_deposit_J_rank_next)
        self.prerank_in_deposit = os.environ.get('FBPIC_AMD_PRERANK', '1') != '0'
        self.ptcl = []
        if n_e is not None:
            self.add_new_species(q=-e, m=m_e, n=n_e, dens_func=dens_func,
                                 p_nz=p_nz, p_nr=p_nr, p_nt=p_nt, p_zmin=p_zmin, p_zmax=p_zmax,
                                 p_rmin=p_rmin, p_rmax=p_rmax)
            if initialize_ions:
                self.add_new_species(q=e, m=m_p, n=n_e, dens_func=dens_func,
                                     p_nz=p_nz, p_nr=p_nr, p_nt=p_nt, p_zmin=p_zmin,
                                     p_zmax=p_zmax, p_rmin=p_rmin, p_rmax=p_rmax)
        self.time = 0.
        self.iteration = 0
        self.filter_currents = filter_currents
        self.external_fields = []
        self.diags = []
        self.checkpoints = []
        self._J_transform_pending = False
        self._rho_already_erased = False
        self.laser_antennas = []
        self.mirrors = []
        # On a single z-periodic domain the reference re-deposits rho_prev at every step
        # (exchange_period = 1, main.py:435-449) although no particle was added or removed:
        # the spectral rho_prev left by push_rho (= the previous step's filtered rho_next)
        # already is that charge density up to rounding.  The re-deposit is skipped after
        # the first iteration of a step() call unless this flag is set.
        self.redeposit_rho_prev_every_step = False
        self._in_step = False
        # gather + push_p + push_x(dt/2) in one kernel when no hook sits between them
        self.fuse_gather_push = True
        # True: launch every operation of the reference's step on its own, in the reference's
        # order (no fused transforms / solver launch, no deferred push, identity FFT round trip
        # of E, B kept): the baseline the fused sequence is measured against (bench.py
        # --reference-sequence); results identical within the test tolerances
        self.reference_sequence = False
        # deposit('J') + push_x(dt/2) + re-sort + deposit('rho_next') as ONE pass over the
        # particles (fb_push_x_sort_deposit_J_rho) when nothing sits between them in step()
        self.fuse_J_into_rho = os.environ.get('FBPIC_AMD_FUSE_J', '1') != '0'
        self.skip_unobserved_first_J = True
        self._defer_J_ok = False
        # State carried from one step() call to the next while everything stays on the GPU and
        # nobody touched the particle / field tensors in between (torch version counters, see
        # _carry_signature): the first iteration of the next call then runs like an interior
        # one - no re-transform of E, B, no particle exchange / rho_prev re-deposit "because
        # the user may have changed the particles" (reference main.py:435-449), and J / rho
        # come back to the interpolation grid only if something reads them (Fields.defer_sources).
        # `for _ in range(n): sim.step(1)` then costs what sim.step(n) costs.
        self.carry_state_between_calls = os.environ.get('FBPIC_AMD_CARRY', '1') != '0'
        self._carry = None
        # Decomposed / open-boundary runs: what runs on a second stream next to the particle
        # work of the next step (reference schedule: main.py:719-769 then :469-490, serial).
        #   'off'   (default) nothing - everything on the compute stream;
        #   'fft'   the forward FFT of the exchanged (z-real) E, B back to spectral space (only
        #           the NEXT field push reads it), next to the gather + push;
        #   'split' also the message itself and the inverse Hankel transform of the guard rows,
        #           while the main stream transforms the rows the exchange does not touch and
        #           gathers + pushes the particles of those rows, then the remaining ones.
        # Measured on one MI355X with a device copy standing in for the transport (C2 per rank,
        # tools/loopback_multirank.py): off 0.58, fft 0.59, split 0.64 ms per step - the particle
        # kernels already fill the GPU, a concurrent kernel only time-shares with them, and the
        # split costs two extra launches.  'split' can only pay when a real message latency
        # (xGMI, RCCL) exceeds ~60 us; it is kept selectable for that measurement
        # (FBPIC_AMD_OVERLAP) and pinned by tests/test_gpu_multirank_golden.py.
        self.overlap_guard_exchange = os.environ.get('FBPIC_AMD_OVERLAP', 'off')
        # the first half of a particle hand-over (selection, messages, request of the host read) is posted
        # behind the particle pass of the iteration before (see _handover_can_start_early)
        self.early_handover = os.environ.get('FBPIC_AMD_EARLY_HANDOVER', '1') != '0'
        # The particle work of an iteration as ONE pass (Particles.cycle, csrc/cycle.hip) with a
        # re-sort every few steps only, where the conditions of _one_pass_ok hold; else the
        # two-pass sequence (gather + push + rank | deposit J + push + sort + deposit rho).
        self.one_pass_cycle = os.environ.get('FBPIC_AMD_ONE_PASS', '1') != '0'
        # ... for the cubic shape too (round 6: k_cycle_cubic) - built, pinned, and OFF by default: at C5 (2048 x 512,
        # Nm = 4, 64 ppc) the pass takes 15.5 ms where the two passes take 5.3 + 5.5 (its stencil sums are
        # lane-by-lane LDS reads - 393 KB per 64 particles - where the two-pass gather runs on the matrix
        # cores, and 256 VGPRs + 18 KB of LDS leave 2 waves per SIMD): profiles/r06_c5_onepass_vs_twopass.txt
        self.one_pass_cubic = False
        self._eb_pending = None
        self._comm_stream = None

    # -------------------------------------------------------------------- PIC cycle
    def step(self, N=1, correct_currents=True, correct_divE=False, use_true_rho=False,
             move_positions=True, move_momenta=True, show_progress=False):
        """Perform N PIC cycles (main.py:346-586)."""
        ptcl, fld, dt = self.ptcl, self.fld, self.dt
        if correct_divE:
            raise NotImplementedError('correct_divE is outside the fbpic_amd hot path')
        if self.comm.size > 1 and use_true_rho and correct_currents:
            raise ValueError('`use_true_rho` cannot be used together with '
                             '`correct_currents` in multi-proc mode.')
        if self.comm.moving_win is not None:
            for species in ptcl:
                if species.continuous_injection:
                    species.injector.initialize_injection_positions(
                        self.comm, self.comm.moving_win.v, species.z, self.dt)
        was_on_gpu = fld.data_is_on_gpu and all(s.data_is_on_gpu for s in ptcl)
        send_data_to_gpu(self)
        self._in_step = True
        fld._in_step = True
        for species in ptcl:
            species._in_step = True
        carried = (was_on_gpu and self._carry is not None and N > 0
                   and self._carry == self._carry_signature())
        self._carry = None
        self._last_call_carried = carried
        self._data_stays_on_gpu = was_on_gpu
        try:
            self._step_loop(N, correct_currents, use_true_rho, move_positions, move_momenta,
                            carried=carried)
        finally:
            self._in_step = False
            fld._in_step = False
            for species in ptcl:
                species._in_step = False
        if not was_on_gpu:
            receive_data_from_gpu(self)
        elif self.carry_state_between_calls and N > 0:
            self._carry = self._carry_signature()

    def _carry_signature(self):
        """What must be unchanged for the device state left by one step() call to be the state
        the next call starts from: the tensors themselves (address + torch version counter: any
        in-place modification through the public attributes bumps it; this package's kernels
        write through raw pointers and do not), the host <-> device epochs, the particle
        numbers, the grid position.  Only for z-periodic runs without external hooks: an open
        boundary is damped again at the start of every call in the reference (main.py:403-411),
        which this keeps."""
        fld, comm = self.fld, self.comm
        if comm.size > 1:
            # every rank would have to take the same decision (a rank that re-exchanges while its
            # neighbour carries on posts messages nobody receives): not carried when decomposed
            return None
        if not self.carry_state_between_calls or comm.nz_damp != 0 or comm.moving_win is not None \
                or self.use_galilean or self.external_fields or self.mirrors or self.laser_antennas \
                or self.reference_sequence or fld.current_correction == 'cross-deposition':
            return None
        if not (fld.data_is_on_gpu and all(s.data_is_on_gpu for s in self.ptcl)):
            return None
        if not _version_counter_works():
            return None
        sig = [self.iteration, fld._epoch, getattr(fld, '_ext_gen', 0), fld.d_interp.data_ptr(),
               fld.d_interp._version,
               fld.d_spect.data_ptr(), fld.d_spect._version, fld.interp[0].zmin,
               comm._zmin_global_domain, len(self.ptcl), self.dt, self.filter_currents]
        for sp in self.ptcl:
            sig.append((id(sp), sp.Ntot, sp.q, sp.m, sp._epoch, sp._ext_gen, sp._pending_push,
                        sp._pending_J is None))
            for k in ('x', 'y', 'z', 'ux', 'uy', 'uz', 'w', 'inv_gamma'):
                a = getattr(sp, k)
                sig.append((a.data_ptr(), a._version))
        return tuple(sig)

    def _step_loop(self, N, correct_currents, use_true_rho, move_positions, move_momenta,
                   carried=False):
        """The N iterations of one step() call.  Per iteration: particle exchange / rho_prev
        (_start_of_iteration), the particle work in one of three forms (_particles_one_pass,
        _particles_two_pass, _particles_reference_order), the field update (_field_update), the
        E, B tail (exchange_and_damp_EB)."""
        ptcl, fld, dt = self.ptcl, self.fld, self.dt
        # J / rho (and the particles' E, B) of the previous call that nobody read are not brought
        # back any more
        fld.drop_deferred_sources()
        for species in ptcl:
            species.drop_deferred_fields()
        if not carried:
            # E and B go to spectral space once; afterwards only spectral -> interp
            self.comm.exchange_fields(fld.interp, 'EB', 'replace')
            self.comm.damp_EB_open_boundary(fld.interp)
            fld.interp2spect('EB')
        for i_step in range(N):
            first = (i_step == 0 and not carried)    # "the user may have changed the particles"
            diag_due = any(getattr(d, 'due', lambda it: True)(self.iteration) for d in self.diags)
            fused = (self.fuse_gather_push and move_momenta and move_positions
                     and not self.external_fields and not diag_due)
            wrap_z = self._start_of_iteration(first, fused, diag_due, use_true_rho)
            for species in ptcl:
                species.keep_fields_sorted = True
            lazy_eb = False
            if fused and i_step == N - 1 and self._can_defer_particle_fields():
                # Last iteration of the call: the reference leaves the gathered E, B of this step
                # in the particle arrays (48 B per particle written for whoever reads them).
                # Instead, keep a copy of the E, B grids the gather reads (25 MB at 1024 x 128)
                # and evaluate species.Ex ... on first use (Particles.defer_fields).
                fld.snapshot_EB()
                lazy_eb = True
            store = (i_step == N - 1 and not lazy_eb)
            cross = bool(correct_currents) and fld.current_correction == 'cross-deposition'
            one_pass = fused and self._one_pass_ok(correct_currents, use_true_rho, cross)
            # An iteration that has to re-sort runs the two-pass sequence instead: its second pass
            # walks the particles in destination order anyway and records the home cells for the
            # one-pass iterations that follow (a stand-alone sort moves 176 B per particle)
            # (every species is asked, whatever the others answer: the question also takes in the
            # species' pending counter read-backs and counts its suspension window down)
            sorting = one_pass and any([sp.cycle_wants_sort(fld) for sp in ptcl])
            for species in ptcl:
                species.record_home_in_sort_pass = sorting
            if one_pass and not sorting:
                self._particles_one_pass(store, wrap_z, correct_currents, use_true_rho)
            else:
                if fused:
                    self._gather_push_fused(store, wrap_z, cross)
                else:
                    self._gather_push_reference_order(move_momenta, move_positions)
                self._deposit_push_deposit(correct_currents, use_true_rho, move_positions, cross)
            if self._handover_can_start_early(i_step, N):
                # the next iteration hands particles over: selection, messages and the request of the
                # host read go into the stream HERE, behind the particle pass (the positions are
                # final); the host read then waits for that point of the stream only, and the host's
                # part of the hand-over overlaps the field kernels queued below
                for species in ptcl:
                    self.comm.begin_exchange_particles(species, fld)
            shifted_by = self._field_update(correct_currents, use_true_rho, cross)
            if self.comm.moving_win is not None:
                self.comm.move_grids(fld, ptcl, dt, self.time, spect_shifted_by=shifted_by)
            self.exchange_and_damp_EB()
            self.time += dt
            self.iteration += 1
            if lazy_eb:
                for species in ptcl:
                    species.defer_fields(fld, self.comm.get_rmax(with_damp=False), dt)
            if any(ck.due(self.iteration) for ck in self.checkpoints):
                self._wait_eb()          # the dump reads the guard rows of E, B
            for checkpoint in self.checkpoints:
                checkpoint.write(self.iteration)
        self._wait_eb()
        if self.carry_state_between_calls and self.comm.size == 1 and not self.reference_sequence:
            # J and rho go back to the interpolation grid when something reads them there
            # (interp[m].Jr ..., receive_fields_from_gpu, a direct deposit), not at every call
            fld.defer_sources(self._sources_to_interp)
        else:
            self._sources_to_interp()

    def _start_of_iteration(self, first, fused, diag_due, use_true_rho):
        """Particle exchange, rho_prev and the diagnostics-only J of main.py:435-451.  Returns the
        periodic box (zmin, zmax) when the wrap of z is left to the particle pass that follows."""
        ptcl, fld = self.ptcl, self.fld
        wrap_z = None
        if self.iteration % self.comm.exchange_period == 0 or first:
            need_rho_prev = (first or self.comm.n_guard != 0
                             or self.redeposit_rho_prev_every_step or use_true_rho)
            if fused and not need_rho_prev and self.comm.n_guard == 0:
                # single periodic domain: the wrap of z into the box rides along in the
                # gather+push launch that comes next (nothing reads z in between)
                wrap_z = (fld.interp[0].zmin, fld.interp[0].zmax)
            else:
                self._wait_eb()      # one exchange at a time on the communicator
                for species in ptcl:
                    self.comm.exchange_particles(species, fld, self.time)
            if need_rho_prev:
                self.deposit('rho_prev', exchange=(use_true_rho is True))
        if first and (diag_due or self.reference_sequence or not self.skip_unobserved_first_J):
            # "For the field diagnostics of the first step: deposit J" (reference main.py:
            # 448-451; not the corrected current).  Nothing else reads it - the J of this step
            # is erased and deposited again below - so it is only launched when a diagnostic
            # is due at this iteration (or the reference launch sequence is asked for).
            self.deposit('J', exchange=True)
        return wrap_z

    def _handover_can_start_early(self, i_step, N):
        """The hand-over of the NEXT iteration may be begun behind this iteration's particle pass: a
        decomposed domain whose box does not move in between (no moving window: the ownership rule
        compares z with the box edges of the iteration it belongs to), the next iteration inside this
        call (between calls the user may change the particles), one exchange at a time on the
        communicator (no E, B tail on the second stream)."""
        comm = self.comm
        return bool(self.early_handover and comm.size > 1 and comm.n_guard != 0
                    and comm.moving_win is None and i_step + 1 < N
                    and (self.iteration + 1) % comm.exchange_period == 0
                    and self.overlap_guard_exchange == 'off' and self._eb_pending is None
                    and all(getattr(sp.z, 'is_cuda', False) for sp in self.ptcl))

    def _one_pass_ok(self, correct_currents, use_true_rho, cross):
        """Whether the particle work of this iteration can be the single pass of
        Particles.cycle: grids that do not move between the gather and the depositions (a Galilean
        grid does; a moving window advances at the END of an iteration, main.py:559-563 - between
        iterations the home cells are re-keyed, Particles._home_shift), the in-step (record)
        deposition target, nothing on a second stream waiting for a split gather, every species
        supported by the kernel."""
        fld, comm = self.fld, self.comm
        if not (self.one_pass_cycle and not self.reference_sequence and not cross
                and not self.use_galilean):
            return False
        if self.particle_shape != 'linear' and not self.one_pass_cubic:
            return False
        if comm.size > 1 and ((correct_currents is False) or (use_true_rho is True)):
            return False             # those deposits exchange their guard cells on the interpolation grid
        pend = self._eb_pending
        if pend is not None and pend[1] is not None:
            return False
        return all(sp.cycle_supported(fld.Nm) for sp in self.ptcl)

    def _spectral_cycle_ok(self):
        """Whether the forward Hankel transform of J, rho_next may wait for the solver step and
        run with it and the inverse transform of E, B as one launch (Fields.spect_cycle): inside
        step() on a single z-periodic domain with the standard PSATD scheme, where exactly
        psatd_step and spect2interp('EB') follow the deposition's transform."""
        comm = self.comm
        return bool(self._in_step and not self.reference_sequence and comm.size == 1
                    and comm.nz_damp == 0 and comm.moving_win is None and self.v_comoving is None
                    and not self.mirrors and self.fld.current_correction == 'curl-free')

    def _hankel_deferral(self):
        """How the forward Hankel transform of the freshly deposited J, rho_next is run: True - with
        the solver step and the inverse transform of E, B (single periodic domain, _spectral_cycle_ok);
        'correct' - with the curl-free correction, in front of the J guard exchange of a decomposed
        domain (the in-step transform is only reached there when the correction will run: the
        uncorrected deposits exchange their guard cells on the interpolation grid); False - on its own."""
        if self._spectral_cycle_ok():
            return True
        if (self._in_step and not self.reference_sequence and self.comm.size > 1
                and self.v_comoving is None and not self.mirrors
                and self.fld.current_correction == 'curl-free'):
            return 'correct'
        return False

    def _particles_one_pass(self, store_fields, wrap_z, correct_currents, use_true_rho):
        """gather, push_p, push_x, deposit('J'), push_x, deposit('rho_next') (main.py:469-528) as
        one pass per species; J and rho_next are then transformed together."""
        fld = self.fld
        self._wait_eb()
        self._flush_J_transform()
        records = self.particle_shape == 'linear'
        if records:
            fld.erase_source_records()
        else:
            fld.erase('J+rho')       # cubic shape: the J | rho fields of the interpolation slab
        for species in self.ptcl:
            species.cycle(fld, self.comm, self.dt, store_fields=store_fields, wrap_z=wrap_z)
            species.keep_fields_sorted = False
        fld.interp2spect_J_and_rho_next(fuse_filter=self.filter_currents, from_records=records,
                                        defer_hankel=self._hankel_deferral() if records else False)
        # (what deposit() records: single domain only when these are True, see _one_pass_ok)
        fld.exchanged_source['J'] = (correct_currents is False)
        fld.exchanged_source['rho_next'] = (use_true_rho is True)

    def _gather_push_fused(self, store_fields, wrap_z, cross):
        """gather + push_p + push_x(dt/2) of main.py:469-490 as one launch per species: nothing
        observes the particles in between.  The gathered E, B are consumed in registers; the
        per-particle Ex..Bz arrays are only materialised where something can observe them (the
        last iteration of a call; diagnostics take _gather_push_reference_order).  The pass also
        ranks the particles for the sort after the second half push (not with a Galilean grid:
        zmin moves in between; not with cross-deposition: other pushes)."""
        ptcl, fld, dt = self.ptcl, self.fld, self.dt
        hint = (0.5 * dt, 1., 1., 1.) if (self.prerank_in_deposit and not self.use_galilean
                                          and not cross) else None
        pend = self._eb_pending
        if pend is not None and pend[1] is not None and hint is not None and wrap_z is None \
                and all(sp.can_split_gather(fld.Nm) for sp in ptcl if sp.q != 0):
            # (a neutral species has nothing to gather: gather_push pushes it once, with the
            # 'outside' part)
            # the rows [lo, hi) of the interpolation grid are final; a particle of cell
            # row iz_upper reads rows iz_upper - 2 ... iz_upper + 1 at most (cubic shape)
            rows = (pend[1] + 2, pend[2] - 2)
            for species in ptcl:
                species.gather_push(fld.interp, self.comm, 0.5 * dt, store_fields=store_fields,
                                    rank_next=hint, part='inside', rows=rows)
            self._wait_eb()
            for species in ptcl:
                species.gather_push(fld.interp, self.comm, 0.5 * dt, store_fields=store_fields,
                                    rank_next=hint, part='outside', rows=rows)
        else:
            self._wait_eb(rows_only=True)
            for species in ptcl:
                species.gather_push(fld.interp, self.comm, 0.5 * dt, store_fields=store_fields,
                                    wrap_z=wrap_z, rank_next=hint)

    def _gather_push_reference_order(self, move_momenta, move_positions):
        """main.py:469-490 launch by launch (external fields, diagnostics in between)."""
        ptcl, fld, dt = self.ptcl, self.fld, self.dt
        self._wait_eb()
        for species in ptcl:
            species.gather(fld.interp, self.comm)
        for ext_field in self.external_fields:
            ext_field.apply_expression(ptcl, self.time)
        for diag in self.diags:
            diag.write(self.iteration)
        if move_momenta:
            for species in ptcl:
                species.push_p(self.time + 0.5 * dt)
        if move_positions:
            for species in ptcl:
                species.push_x(0.5 * dt)

    def _deposit_push_deposit(self, correct_currents, use_true_rho, move_positions, cross):
        """deposit('J'), push_x(dt/2), deposit('rho_next') of main.py:492-528 (with the Galilean
        shifts and the cross-deposition in between); in the default sequence the three ride in
        one destination-ordered pass (Particles.deposit decides)."""
        ptcl, fld, dt = self.ptcl, self.fld, self.dt
        if self.use_galilean:
            self.shift_galilean_boundaries(0.5 * dt)
        for species in ptcl:
            species.handle_elementary_processes(self.time + 0.5 * dt)
        for species in ptcl:
            species.keep_fields_sorted = False
        if move_positions and not self.use_galilean and not cross and self.prerank_in_deposit:
            # the J deposition also ranks the particles for the sort after the push below
            # (not with a Galilean grid: zmin moves between this deposit and that sort)
            for species in ptcl:
                species.push_after_deposit_J = (0.5 * dt, 1., 1., 1.)
        self._defer_J_ok = (move_positions and not self.use_galilean and not cross
                            and self.fuse_J_into_rho)
        self.deposit('J', exchange=(correct_currents is False),
                     defer_transform=(not cross) and not self.reference_sequence)
        self._defer_J_ok = False
        for species in ptcl:
            species.push_after_deposit_J = None
        if cross:
            self.cross_deposit(move_positions)
        if move_positions:
            # deferred: the push is folded into the sort that deposit('rho_next') triggers
            for species in ptcl:
                species.push_x(0.5 * dt, defer=not self.reference_sequence)
        if self.use_galilean:
            self.shift_galilean_boundaries(0.5 * dt)
        self.deposit('rho_next', exchange=(use_true_rho is True))
        for species in ptcl:
            species.flush_pending_push()      # species that did not deposit

    def _field_update(self, correct_currents, use_true_rho, cross):
        """Current correction, J guard exchange, PSATD push of E, B (main.py:530-557).  Returns the
        number of cells by which the push has already moved the window (or None)."""
        fld = self.fld
        shifted_by = None
        if self.v_comoving is not None:
            # Galilean / comoving scheme: per-mode correction and push (complex tables)
            if correct_currents:
                fld.correct_currents(check_exchanges=(self.comm.size > 1))
                if self.comm.size > 1:
                    fld.spect2partial_interp('J')
                    self.comm.exchange_fields(fld.interp, 'J', 'add')
                    fld.partial_interp2spect('J')
                fld.exchanged_source['J'] = True
            fld.push(use_true_rho, check_exchanges=(self.comm.size > 1))
        elif self.comm.size == 1 and self.reference_sequence:
            if correct_currents:
                fld.correct_currents()
                fld.exchanged_source['J'] = True
            fld.push(use_true_rho)
        elif self.comm.size == 1:
            # single domain: correction, push and rho shift are cell-local -> one launch,
            # which also translates the fields when the moving window advances right after
            if cross:
                fld.correct_currents()
            if self.comm.moving_win is not None:
                shifted_by = self.comm.moving_win.peek_n_move(self.comm, self.time)
            fld.psatd_step(correct_currents and not cross, use_true_rho,
                           n_move=(shifted_by or 0))
            if correct_currents:
                fld.exchanged_source['J'] = True
        else:
            if correct_currents:
                assert fld.exchanged_source['J'] is False
                if cross:
                    fld.correct_currents(check_exchanges=True)
                else:
                    fld.psatd_step(use_true_rho=use_true_rho, only_correct=True)
                fld.spect2partial_interp('J')
                self.comm.exchange_fields(fld.interp, 'J', 'add')
                fld.partial_interp2spect('J')
                fld.exchanged_source['J'] = True
            assert fld.exchanged_source['J'] is True
            fld.psatd_step(correct_currents=False, use_true_rho=use_true_rho)
        return shifted_by

    def _can_defer_particle_fields(self):
        """Same conditions as the carried state: a z-periodic single domain whose grid does not
        move, positions and momenta both advanced by the fused passes (the deferred evaluation
        steps the positions back by one full push)."""
        comm = self.comm
        # (a call that copies everything back to the host at its end stores them in the gather)
        return bool(self.carry_state_between_calls and getattr(self, '_data_stays_on_gpu', False)
                    and comm.size == 1 and comm.nz_damp == 0
                    and comm.moving_win is None and not self.use_galilean
                    and self.fld.current_correction != 'cross-deposition'
                    and all(sp.use_bin_sort for sp in self.ptcl))

    def _sources_to_interp(self):
        """Tail of step (main.py:572-586): J and rho_prev from spectral space to the
        interpolation grid, guard cells added if that has not happened."""
        fld = self.fld
        fld.spect2interp('J')
        if (not fld.exchanged_source['J']) and (self.comm.size > 1):
            self.comm.exchange_fields(fld.interp, 'J', 'add')
        fld.spect2interp('rho_prev')
        if (not fld.exchanged_source['rho_prev']) and (self.comm.size > 1):
            self.comm.exchange_fields(fld.interp, 'rho', 'add')

    def cross_deposit(self, move_positions):
        """Cross-deposition (main.py:672-716), called with the particles at t = n+1/2:
        charge density at (z[n], x[n+1]) -> rho_next_xy, at (z[n+1], x[n]) -> rho_next_z,
        then back to n+1/2.  Each push rides along in the sort of the deposit that follows."""
        dt = self.dt
        if self.laser_antennas:
            raise NotImplementedError('laser antennas are outside the fbpic_amd scope')

        def push(frac, x_push, y_push, z_push, defer):
            if move_positions:
                for species in self.ptcl:
                    species.push_x(frac * dt, x_push=x_push, y_push=y_push, z_push=z_push,
                                   defer=defer)
        push(0.5, 1., 1., -1., True)          # z[n+1/2], x[n+1/2] => z[n], x[n+1]
        if self.use_galilean:
            self.shift_galilean_boundaries(-0.5 * dt)
        self.deposit('rho_next_xy')
        for species in self.ptcl:
            species.flush_pending_push()
        push(1., -1., -1., 1., True)          # z[n], x[n+1] => z[n+1], x[n]
        if self.use_galilean:
            self.shift_galilean_boundaries(dt)
        self.deposit('rho_next_z')
        for species in self.ptcl:
            species.flush_pending_push()
        push(0.5, 1., 1., -1., False)         # z[n+1], x[n] => z[n+1/2], x[n+1/2]
        if self.use_galilean:
            self.shift_galilean_boundaries(-0.5 * dt)

    def _flush_J_transform(self):
        """Transform a J whose interp2spect was deferred (see deposit)."""
        if self._J_transform_pending:
            self._J_transform_pending = False
            self._rho_already_erased = False
            if self.particle_shape == 'linear':
                self.fld.unpack_source_records()  # J was deposited into the record array
            self.fld.interp2spect('J', fuse_divide_by_volume=True, fuse_filter=self.filter_currents)

    def shift_galilean_boundaries(self, dt):
        """Shift the interpolation grids by v_comoving * dt: only the position attributes
        change (main.py:772-790)."""
        shift_distance = self.v_comoving * dt
        self.comm.shift_global_domain_positions(shift_distance)
        for m in range(self.fld.Nm):
            self.fld.interp[m].zmin += shift_distance
            self.fld.interp[m].zmax += shift_distance

    def deposit(self, fieldtype, exchange=False, update_spectral=True, species_list=None,
                defer_transform=False):
        """Deposit rho or J on the interpolation grid, then transform and filter
        (main.py:588-670).  `defer_transform` (used by step for the J deposit that is followed
        by deposit('rho_next')): J stays on the interpolation grid and is transformed
        together with rho_next, in one FFT and one Hankel launch."""
        fld = self.fld
        self._wait_eb()          # the transforms below use the scratch slab of a pending exchange
        if species_list is None:
            species_list = [s for s in self.ptcl if not s.is_tracer]
        if fieldtype.startswith('rho'):
            kind = 'rho'
        elif fieldtype == 'J':
            kind = 'J'
        else:
            raise ValueError('Unknown fieldtype: %s' % fieldtype)
        in_step = self._in_step and not self.reference_sequence
        fused = in_step and update_spectral and not (exchange and self.comm.size > 1)
        records = False
        # node-major record target (one cache line per node): measured on MI355X, linear shape
        # deposit J 108 -> 88 us, rho 62 -> 57 us, against +10 us for the z-FFT that then
        # gathers 16-B pieces; with the cubic shape (16 nodes per cell either way) it loses
        use_records = self.particle_shape == 'linear'
        if fused and defer_transform and fieldtype == 'J':
            # deposit('rho_next') follows: both source groups are zeroed in one launch and
            # transformed together
            self._rho_already_erased = True
            if use_records:
                fld.erase_source_records()
                records = True
            else:
                fld.erase('J+rho')
            # the deposition itself may ride along in the pass that pushes, sorts and deposits
            # rho_next (Particles.deposit decides; anything in between launches it first)
            for species in species_list:
                species.defer_J_deposit = self._defer_J_ok
        elif fused and kind == 'rho' and self._rho_already_erased and fieldtype == 'rho_next':
            self._rho_already_erased = False
            records = use_records
        else:
            self._flush_J_transform()
            self._rho_already_erased = False
            fld.erase(kind)
        for species in species_list:
            species.deposit(fld, kind, records=records)
        fld.sum_reduce_deposition_array(kind)
        if in_step and update_spectral and not (exchange and self.comm.size > 1):
            # inside step(): divide-by-volume and filter ride along in the Hankel GEMM
            # (the interpolation-grid J / rho are overwritten from spectral space before
            # anything reads them, main.py:572-577)
            if defer_transform and fieldtype == 'J':
                self._J_transform_pending = True
                fld.exchanged_source[fieldtype] = exchange
                return
            if fieldtype == 'rho_next' and self._J_transform_pending:
                self._J_transform_pending = False
                fld.interp2spect_J_and_rho_next(fuse_filter=self.filter_currents,
                                                from_records=records,
                                                defer_hankel=self._hankel_deferral())
            else:
                self._flush_J_transform()
                fld.interp2spect(fieldtype, fuse_divide_by_volume=True,
                                 fuse_filter=self.filter_currents)
            fld.exchanged_source[fieldtype] = exchange
            return
        self._flush_J_transform()
        fld.divide_by_volume(kind)
        if exchange and self.comm.size > 1:
            self.comm.exchange_fields(fld.interp, kind, 'add')
        if update_spectral:
            fld.interp2spect(fieldtype)
            if self.filter_currents:
                fld.filter_spect(fieldtype)
            fld.exchanged_source[fieldtype] = exchange

    def exchange_and_damp_EB(self):
        """E/B guard exchange + open-boundary damping in (z-real, r-spectral) space, then
        back to the interpolation grid (main.py:719-769).  On a single periodic rank the
        iFFT/FFT round trip of main.py:741-766 is the identity (nothing touches the
        partial-interp fields) and is skipped (difference ~1e-16 relative)."""
        fld = self.fld
        needs_partial = ((self.comm.size > 1) or (self.comm.nz_damp != 0) or len(self.mirrors) > 0
                         or self.reference_sequence)
        if needs_partial and not self.mirrors and not self.reference_sequence:
            # the (z-real, r-spectral) fields live in the scratch slab; after the exchange and
            # the damping, the interpolation grid follows from them by the inverse Hankel
            # transform alone: one 6*Nm-field FFT launch less than via the spectral fields
            scr = fld.d_scratch
            self._wait_eb()
            fld.spect2partial_interp('EB', to_scratch=True)
            mode = self.overlap_guard_exchange if scr.is_cuda else 'off'
            if mode in ('fft', 'split'):
                t = _capi.torch()
                if self._comm_stream is None:
                    self._comm_stream = t.cuda.Stream()
                    self._ev_ready, self._ev_done = t.cuda.Event(), t.cuda.Event()
                main, side = t.cuda.current_stream(), self._comm_stream
            if mode == 'split' and self.comm.size > 1:
                lo, hi = self.comm.rows_untouched_by_EB_exchange(fld.Nz)
                self._ev_ready.record(main)
                with t.cuda.stream(side):
                    side.wait_event(self._ev_ready)
                    # exchanged / damped rows: message, then their half of the work
                    self.comm.exchange_fields(fld.interp, 'EB', 'replace', slab=scr)
                    self.comm.damp_EB_open_boundary(fld.interp, slab=scr)
                    fld.partial_interp2spect('EB', from_scratch=True)
                    fld.partial2interp('EB', rows=(0, lo))
                    fld.partial2interp('EB', rows=(hi, fld.Nz))
                    self._ev_done.record(side)
                # meanwhile: the rows the exchange does not touch (the transform is local in z)
                fld.partial2interp('EB', rows=(lo, hi))
                self._eb_pending = (self._ev_done, lo, hi)
                return
            self.comm.exchange_fields(fld.interp, 'EB', 'replace', slab=scr)
            self.comm.damp_EB_open_boundary(fld.interp, slab=scr)
            if mode == 'fft':
                # scratch -> spectral E, B on the side stream; the interpolation grid is complete
                # on the main stream, so the next gather does not wait for it (lo = hi: no rows
                # are pending); _wait_eb orders it before the next user of the spectral E, B or
                # of the scratch slab (the deposition's transform)
                self._ev_ready.record(main)
                with t.cuda.stream(side):
                    side.wait_event(self._ev_ready)
                    fld.partial_interp2spect('EB', from_scratch=True)
                    self._ev_done.record(side)
                fld.partial2interp('EB')
                self._eb_pending = (self._ev_done, None, None)
                return
            fld.partial_interp2spect('EB', from_scratch=True)
            fld.partial2interp('EB')
            return
        if needs_partial:
            fld.spect2partial_interp('EB')
            self.comm.exchange_fields(fld.interp, 'EB', 'replace')
            self.comm.damp_EB_open_boundary(fld.interp)
            for mirror in self.mirrors:
                mirror.set_fields_to_zero(fld.interp, self.comm, self.time)
            fld.partial_interp2spect('EB')
        fld.spect2interp('EB')

    def _wait_eb(self, rows_only=False):
        """Make the main stream wait for the E, B tail of the previous step that is still in
        flight on the second stream (no-op otherwise): before anything reads the guard rows of
        the interpolation grid, the spectral E, B, or re-uses the scratch slab.  `rows_only`:
        the caller only reads the interpolation grid - nothing to wait for when just the forward
        FFT of E, B is pending."""
        pend = self._eb_pending
        if pend is None or (rows_only and pend[1] is None):
            return
        self._eb_pending = None
        _capi.torch().cuda.current_stream().wait_event(pend[0])

    def set_moving_window(self, v=c, **deprecated):
        """Attach a window moving at velocity v to the simulation (main.py:1004-1032)."""
        self.comm.moving_win = MovingWindow(self.comm, self.dt, v, self.time)
        # after restart_from_checkpoint: continue from the window's saved continuous position
        saved = getattr(self, '_restart_window', None)
        if saved is not None:
            self.comm.moving_win.zmin, self.comm.moving_win.t_last_move = saved
            self._restart_window = None

    # -------------------------------------------------------------------- species
    def add_new_species(self, q, m, n=None, dens_func=None, p_nz=None, p_nr=None, p_nt=None,
                        p_zmin=-np.inf, p_zmax=np.inf, p_rmin=0, p_rmax=np.inf,
                        uz_m=0., ux_m=0., uy_m=0., uz_th=0., ux_th=0., uy_th=0.,
                        continuous_injection=True, boost_positions_in_dens_func=False,
                        is_tracer=False):
        """Create a species and (if n is given) its evenly-spaced macroparticles
        (main.py:792-1001)."""
        if n is not None:
            for var in [p_nz, p_nr, p_nt]:
                if var is None:
                    raise ValueError('If the density `n` is passed to `add_new_species`,\n'
                                     'then the arguments `p_nz`, `p_nr` and `p_nt` need '
                                     'to be passed too.')
            zmin_l, zmax_l = self.comm.get_zmin_zmax(local=True, rank=self.comm.rank,
                                                     with_damp=False, with_guard=False)
            p_zmin = max(zmin_l, p_zmin)
            p_zmax = min(zmax_l, p_zmax)
            p_rmax = min(self.comm.get_rmax(with_damp=False), p_rmax)
            p_zmin, p_zmax, Npz = adapt_to_grid(self.fld.interp[0].z, p_zmin, p_zmax, p_nz)
            p_rmin, p_rmax, Npr = adapt_to_grid(self.fld.interp[0].r, p_rmin, p_rmax, p_nr)
            dz_particles = self.comm.dz / p_nz
        else:
            n = 0
            p_zmin = p_zmax = p_rmin = p_rmax = 0
            Npz = Npr = p_nt = 0
            continuous_injection = False
            dz_particles = 0.
        new_species = Particles(q=q, m=m, n=n, dens_func=dens_func, Npz=Npz, zmin=p_zmin,
                                zmax=p_zmax, Npr=Npr, rmin=p_rmin, rmax=p_rmax, Nptheta=p_nt,
                                dt=self.dt, particle_shape=self.particle_shape, use_cuda=True,
                                grid_shape=self.grid_shape, ux_m=ux_m, uy_m=uy_m, uz_m=uz_m,
                                ux_th=ux_th, uy_th=uy_th, uz_th=uz_th,
                                continuous_injection=continuous_injection,
                                dz_particles=dz_particles, is_tracer=is_tracer)
        if self.fld.data_is_on_gpu:
            new_species.send_particles_to_gpu()
        self.ptcl.append(new_species)
        return new_species


def adapt_to_grid(x, p_xmin, p_xmax, p_nx, ncells_empty=0):
    """Snap [p_xmin, p_xmax] to cell edges of the grid x and count p_nx particles per
    enclosed cell (main.py:1056-1111)."""
    xmin, xmax = x.min(), x.max()
    dx = x[1] - x[0]
    if p_xmin < xmin - 0.5 * dx:
        p_xmin = xmin - 0.5 * dx
    if p_xmax > xmax + (0.5 - ncells_empty) * dx:
        p_xmax = xmax + (0.5 - ncells_empty) * dx
    x_load = x[(x > p_xmin) & (x < p_xmax)]
    Npx = len(x_load) * p_nx
    if Npx > 0:
        p_xmin = x_load.min() - 0.5 * dx
        p_xmax = x_load.max() + 0.5 * dx
    return p_xmin, p_xmax, Npx
