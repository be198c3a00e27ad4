"""Moving window: the grid follows the laser / bunch at velocity v.

Restates fbpic/boundaries/moving_window.py:14-239.  Each time the window has advanced by
at least one cell, the spectral fields E, B and rho_prev are translated backwards by
n_move cells -- a multiplication by exp(i kz_true dz)^n_move in spectral space (device
kernel fb_shift_spect) --, the grid edges move by n_move*dz and the positions between which
new plasma is injected at the right edge are advanced.
"""
from .. import _capi


class MovingWindow(object):
    def __init__(self, comm, dt, v, time):
        if ((comm.rank == comm.size - 1) and (comm.right_proc is not None)) \
                or ((comm.rank == 0) and (comm.left_proc is not None)):
            raise ValueError('The simulation is using a moving window, but the boundaries are '
                             'periodic.\n Please select open boundaries when initializing '
                             'the Simulation object.')
        self.v = v
        self.t_last_move = time - dt
        zmin_global, _ = comm.get_zmin_zmax(local=False, with_damp=False, with_guard=False)
        # every rank keeps the window position: n_move is then computed redundantly and
        # identically on all ranks instead of being broadcast (moving_window.py:89-98)
        self.zmin = zmin_global

    def peek_n_move(self, comm, time):
        """Number of cells by which the next move_grids(.., time) will shift the grids (same
        arithmetic, nothing modified): lets Simulation.step fold the spectral translation
        into the field push that precedes it."""
        zmin = self.zmin + self.v * (time - self.t_last_move)
        zmin_global, _ = comm.get_zmin_zmax(local=False, with_damp=False, with_guard=False)
        return int((zmin - zmin_global) / comm.dz)

    def move_grids(self, fld, ptcl, comm, time, spect_shifted_by=None):
        """`spect_shifted_by`: the spectral fields have already been translated by that many
        cells (by the field push, fb_psatd_step_standard_shift)."""
        dz = comm.dz
        self.zmin += self.v * (time - self.t_last_move)
        zmin_global, _ = comm.get_zmin_zmax(local=False, with_damp=False, with_guard=False)
        n_move = int((self.zmin - zmin_global) / dz)
        if spect_shifted_by is not None:
            assert spect_shifted_by == n_move, (spect_shifted_by, n_move)
        if n_move != 0:
            comm.shift_global_domain_positions(n_move * dz)
            for m in range(len(fld.interp)):
                fld.interp[m].zmin += n_move * fld.interp[m].dz
                fld.interp[m].zmax += n_move * fld.interp[m].dz
            if spect_shifted_by is None:
                self.shift_spect_grids(fld, n_move)
        for species in ptcl:
            species.prefix_sum_shift += n_move
        if comm.rank == comm.size - 1:
            for species in ptcl:
                if species.continuous_injection:
                    species.injector.increment_injection_positions(self.v, time - self.t_last_move)
        self.t_last_move = time

    def shift_spect_grids(self, fld, n_move, shift_rho=True, shift_currents=True):
        """E, B, rho_prev and J (the reference's actual defaults, moving_window.py:133-134)
        of every mode, one launch per mode (the shift factor
        exp(i kz_true dz) is the same for all modes, the call is kept per mode to mirror
        shift_spect_grid of the reference)."""
        for m in range(fld.Nm):
            self.shift_spect_grid(fld.spect[m], n_move, shift_rho, shift_currents)

    def shift_spect_grid(self, grid, n_move, shift_rho=True, shift_currents=True):
        names = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz']
        if shift_rho:
            names.append('rho_prev')
        if shift_currents:
            names += ['Jp', 'Jm', 'Jz']
        arrs = [getattr(grid, k) for k in names]
        rc = _capi.lib().fb_shift_spect(len(arrs), _capi.ptr_array(arrs), _capi.row_stride(arrs[0]),
                                        _capi.ptr(grid.d_field_shift), n_move, grid.Nz, grid.Nr,
                                        _capi.stream())
        _capi.check(rc, 'fb_shift_spect')
