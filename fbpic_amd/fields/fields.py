"""`Fields`: all grid data of the simulation and the interpolation <-> spectral transforms.

Same public surface as the reference's `Fields` (fbpic/fields/fields.py:20-625):
`interp[m]`, `spect[m]`, `trans[m]`, `psatd[m]`, `push`, `correct_currents`,
`interp2spect`, `spect2interp`, `spect2partial_interp`, `partial_interp2spect`, `erase`,
`divide_by_volume`, `filter_spect`, `sum_reduce_deposition_array`, `exchanged_source`.

MI355X data layout.  While on the GPU, all interpolation-grid fields of all modes live in
ONE z-major slab `complex128[Nz, 10*Nm, Nr]` and all spectral fields in a second slab
`complex128[Nz, 11*Nm, Nr]`; `interp[m].Ez` etc. are (Nz, Nr) views with row stride
`NF*Nr`.  Consequences:
  * one rocFFT call transforms a whole group (e.g. Er,Et,Ez of every mode = 3*Nm*Nr
    columns, stride NF*Nr, distance 1) -- no per-field launches, no transposes;
  * one MFMA launch applies the 3*Nm different Hankel matrices of that group;
  * a z-range of guard cells of every field is one contiguous block (multi-GPU exchange).
Field order inside a slab: E | B | J | rho(_prev | _next), each vector group ordered
[mode0: (r|p, t|m, z), mode1: ...].
"""
import os
import numpy as np
from .. import _capi
from .interpolation_grid import InterpolationGrid, INTERP_FIELDS
from .spectral_grid import SpectralGrid, SPECT_FIELDS
from .psatd_coefs import PsatdCoeffs
from .smoothing import BinomialSmoother
from .utility_methods import get_modified_k
from .spectral_transform.spectral_transformer import SpectralTransformer
from .spectral_transform.fourier import fft_exec

_VEC = {'E': 0, 'B': 1, 'J': 2}
# extra complex128 elements between consecutive z rows of a slab (see Fields._alloc_slab)
SLAB_PAD = 8
_I_COMP = ('r', 't', 'z')
_S_COMP = ('p', 'm', 'z')


class Fields(object):
    def __init__(self, Nz, zmax, Nr, rmax, Nm, dt, zmin=0., n_order=-1, v_comoving=None,
                 use_pml=False, use_galilean=True, current_correction='curl-free',
                 use_cuda=True, smoother=None, create_threading_buffers=False,
                 use_ruyten_shapes=True, use_modified_volume=True):
        if current_correction not in ('curl-free', 'cross-deposition'):
            raise ValueError('Unkown current correction:%s' % current_correction)
        self.Nz, self.Nr, self.rmax, self.Nm, self.dt = Nz, Nr, rmax, Nm, dt
        self.n_order = n_order
        self.v_comoving = v_comoving
        self.use_galilean = use_galilean if v_comoving is not None else False
        self.smoother = smoother if smoother is not None else \
            BinomialSmoother(n_passes=1, compensator=False)
        self.use_cuda = use_cuda
        self.data_is_on_gpu = False
        self.current_correction = current_correction
        self.trans = [SpectralTransformer(Nz, Nr, m, rmax, use_cuda=use_cuda) for m in range(Nm)]
        self.interp = [InterpolationGrid(Nz, Nr, m, zmin, zmax, rmax, use_pml=use_pml,
                                         use_cuda=use_cuda, use_ruyten_shapes=use_ruyten_shapes,
                                         use_modified_volume=use_modified_volume)
                       for m in range(Nm)]
        for g in self.interp:
            g._owner = self
        dz = (zmax - zmin) / Nz
        kz_true = 2 * np.pi * np.fft.fftfreq(Nz, dz)
        kz_modified = get_modified_k(kz_true, n_order, dz)
        self.spect = []
        self.psatd = []
        for m in range(Nm):
            kr = 2 * np.pi * self.trans[m].dht0.get_nu()
            self.spect.append(SpectralGrid(kz_modified, kr, m, kz_true, self.interp[m].dz,
                                           self.interp[m].dr, current_correction, self.smoother,
                                           use_pml=use_pml, use_cuda=use_cuda))
            self.psatd.append(PsatdCoeffs(self.spect[m].kz, self.spect[m].kr, m, dt, Nz, Nr,
                                          V=self.v_comoving, use_galilean=self.use_galilean,
                                          use_cuda=use_cuda))
        self.exchanged_source = {'J': False, 'rho_prev': False, 'rho_new': False,
                                 'rho_next_xy': False, 'rho_next_z': False}
        # device state (allocated at the first send_fields_to_gpu)
        self.NFi = 10 * Nm
        # + rho_next_z, rho_next_xy of every mode with the cross-deposition correction
        self.NFs = (13 if current_correction == 'cross-deposition' else 11) * Nm
        self.NFx = 6 * Nm
        self.d_interp = None
        self.d_spect = None
        self.d_scratch = None
        self.d_src_rec = None
        self._epoch = 0                  # bumped by every host -> device copy of the grids
        self._deferred_sources = None    # see defer_sources
        # Fused r-spectral part of a step (spect_cycle / fb_spect_cycle_standard): the forward
        # Hankel transform of J, rho_next that interp2spect_J_and_rho_next has left pending
        # (`_pending_hankel` = fuse_filter flag, the z-FFT'd sources wait in the J | rho fields of
        # the interpolation slab), and E, B already in (kz, r) space in the scratch slab
        # (`_EB_in_kz_r`) for the spect2interp('EB') that follows.
        self.fuse_spectral_cycle = os.environ.get('FBPIC_AMD_FUSE_SPECT', '1') != '0'
        self.spect_cycle_launches = 0        # diagnostics: fused forward Hankel + solver + inverse Hankel launches
        self._pending_hankel = None
        self._EB_in_kz_r = False

    # ---------------------------------------------------------------- deferred J / rho
    def _touch(self, keep_pending=False):
        """Start of a public method that modifies the grids: calls from outside Simulation.step are
        counted (the state carried between step() calls is keyed on the count: the kernels
        write through raw pointers, which torch's version counters do not see)."""
        if not getattr(self, '_in_step', False):
            self._ext_gen = getattr(self, '_ext_gen', 0) + 1
        if not keep_pending:
            self._finish_pending_transforms()

    def _finish_pending_transforms(self):
        """Complete what the fused spectral sequence has left half-way (see __init__): anything
        but psatd_step / spect2interp('EB') in their places finds the grids as after the separate
        calls."""
        if self._pending_hankel is not None:
            (fuse_filter, _), self._pending_hankel = self._pending_hankel, None
            self._hankel_J_and_rho_next(self._src_kz_views(), fuse_filter)
        if self._EB_in_kz_r:
            self._EB_in_kz_r = False
            self._backward_zfft('EB')

    def defer_sources(self, bring_back):
        """Simulation.step ends with J and rho_prev going from spectral space to the
        interpolation grid (main.py:572-586): two inverse transforms per call that only matter
        if something reads interp[m].Jr / Jt / Jz / rho afterwards.  Instead of running them,
        remember `bring_back` (the callable that does) and take the four attributes off the
        grids: the first read of one of them (InterpolationGrid.__getattr__), a copy of the
        grids to the host, or a direct erase / deposit runs it.  The next step() call drops it
        (the arrays would have been overwritten by then anyway)."""
        self._deferred_sources = bring_back
        for g in self.interp:
            for name in ('Jr', 'Jt', 'Jz', 'rho'):
                g.__dict__.pop(name, None)

    def snapshot_EB(self):
        """Copy of E, B of all modes on the interpolation grid (the first 6 Nm fields of the
        slab) as they are now: what a gather launched at this point reads
        (Particles.defer_fields evaluates the particles' E, B from it later)."""
        self._need_gpu()
        n = 6 * self.Nm
        if getattr(self, 'd_EB_snap', None) is None:
            self.d_EB_snap = self._alloc_slab(n)
        self.d_EB_snap.copy_(self.d_interp[:, :n, :])
        return self.d_EB_snap

    def _restore_source_views(self):
        for m in range(self.Nm):
            for name in ('Jr', 'Jt', 'Jz', 'rho'):
                setattr(self.interp[m], name, self.d_interp[:, self.interp_index(name, m), :])

    def drop_deferred_sources(self):
        if self._deferred_sources is not None:
            self._deferred_sources = None
            self._restore_source_views()

    def materialize_sources(self):
        bring_back, self._deferred_sources = self._deferred_sources, None
        if bring_back is not None:
            self._restore_source_views()
            bring_back()

    # ---------------------------------------------------------------- slab indexing
    def interp_index(self, name, m):
        if name == 'rho':
            return 9 * self.Nm + m
        return 3 * self.Nm * _VEC[name[0]] + 3 * m + _I_COMP.index(name[1])

    def spect_index(self, name, m):
        if name == 'rho_prev':
            return 9 * self.Nm + m
        if name == 'rho_next':
            return 10 * self.Nm + m
        if name == 'rho_next_z':
            return 11 * self.Nm + m
        if name == 'rho_next_xy':
            return 12 * self.Nm + m
        return 3 * self.Nm * _VEC[name[0]] + 3 * m + _S_COMP.index(name[1])

    # ---------------------------------------------------------------- host <-> device
    def send_fields_to_gpu(self):
        """Move all grid data to the MI355X; attributes become views of the device slabs."""
        if self.data_is_on_gpu:
            return
        t = _capi.torch()
        dev = _capi.require_device()
        Nz, Nr, Nm = self.Nz, self.Nr, self.Nm
        hi = np.empty((Nz, self.NFi, Nr), dtype=np.complex128)
        hs = np.empty((Nz, self.NFs, Nr), dtype=np.complex128)
        for m in range(Nm):
            for name in INTERP_FIELDS:
                hi[:, self.interp_index(name, m), :] = getattr(self.interp[m], name)
            for name in self.spect[m].field_names:
                hs[:, self.spect_index(name, m), :] = getattr(self.spect[m], name)
        if self.d_interp is None:
            self.d_interp = self._alloc_slab(self.NFi)
            self.d_spect = self._alloc_slab(self.NFs)
            self.d_scratch = self._alloc_slab(self.NFx)
            self._build_job_tables()
        self.d_interp.copy_(t.from_numpy(hi))
        self.d_spect.copy_(t.from_numpy(hs))
        self._epoch += 1
        self._deferred_sources = None
        for m in range(Nm):
            for name in INTERP_FIELDS:
                setattr(self.interp[m], name, self.d_interp[:, self.interp_index(name, m), :])
            for name in self.spect[m].field_names:
                setattr(self.spect[m], name, self.d_spect[:, self.spect_index(name, m), :])
            self.interp[m].upload_tables()
            self.spect[m].upload_tables()
            self.psatd[m].device_tables()
        self.data_is_on_gpu = True

    def _alloc_slab(self, nfields):
        """Zeroed z-major slab complex128[Nz, nfields, Nr] whose z rows are SLAB_PAD elements
        apart more than nfields*Nr: row strides that are large multiples of a power of two
        make the strided z-FFT hit a few HBM channels only (measured: 2048x512, Nm=4 forward
        J transform 187 us -> 128 us with the pad)."""
        t = _capi.torch()
        rs = nfields * self.Nr + SLAB_PAD
        base = t.zeros(self.Nz * rs, dtype=t.complex128, device=_capi.require_device())
        return base.as_strided((self.Nz, nfields, self.Nr), (rs, self.Nr, 1))

    def receive_fields_from_gpu(self):
        """Copy all grid data back to host NumPy arrays (C-contiguous, like `.get()`)."""
        if not self.data_is_on_gpu:
            return
        self._finish_pending_transforms()
        self.materialize_sources()
        hi = self.d_interp.cpu().numpy()
        hs = self.d_spect.cpu().numpy()
        for m in range(self.Nm):
            for name in INTERP_FIELDS:
                setattr(self.interp[m], name,
                        np.ascontiguousarray(hi[:, self.interp_index(name, m), :]))
            for name in self.spect[m].field_names:
                setattr(self.spect[m], name,
                        np.ascontiguousarray(hs[:, self.spect_index(name, m), :]))
        self.data_is_on_gpu = False

    def _need_gpu(self):
        if not self.data_is_on_gpu:
            raise _capi.BackendError(
                'Field data is on the host: fbpic_amd only computes on the GPU. Call '
                'send_fields_to_gpu() (or use GpuMemoryManager / Simulation.step) first.')

    # ---------------------------------------------------------------- batched transforms
    def _build_job_tables(self):
        """Per vector group (3*Nm fields, slab order) and per scalar group (Nm fields):
        host arrays of device pointers to the Hankel matrices."""
        fwd, inv = [], []
        fwd_s, inv_s = [], []
        for m in range(self.Nm):
            tr = self.trans[m]
            for d in (tr.dhtp, tr.dhtm, tr.dht0):      # slab order: (r|p, t|m, z)
                M, iM = d.device_matrices()
                fwd.append(M)
                inv.append(iM)
            M, iM = tr.dht0.device_matrices()
            fwd_s.append(M)
            inv_s.append(iM)
        self._mats = {'vec_fwd': fwd, 'vec_inv': inv, 'scal_fwd': fwd_s, 'scal_inv': inv_s}

    def _field_views(self, slab, f0, nf):
        return [slab[:, f0 + j, :] for j in range(nf)]

    def _group(self, fieldtype):
        """(interp first field, spect first field, n fields, is_vector)."""
        Nm = self.Nm
        if fieldtype == 'EB':       # E and B are adjacent in both slabs: one batched transform
            return 0, 0, 6 * Nm, True
        if fieldtype in _VEC:
            f0 = 3 * Nm * _VEC[fieldtype]
            return f0, f0, 3 * Nm, True
        if fieldtype in ('rho_prev', 'rho_next'):
            return 9 * Nm, (9 if fieldtype == 'rho_prev' else 10) * Nm, Nm, False
        if fieldtype in ('rho_next_z', 'rho_next_xy'):
            if self.current_correction != 'cross-deposition':
                raise ValueError('%s needs current_correction="cross-deposition"' % fieldtype)
            return 9 * Nm, (11 if fieldtype == 'rho_next_z' else 12) * Nm, Nm, False
        if fieldtype in ('E_pml', 'B_pml'):
            raise NotImplementedError('%s is outside the fbpic_amd scope' % fieldtype)
        raise ValueError('Invalid string for fieldtype: %s' % fieldtype)

    def _rt_pairs(self, scr_f, n_vec):
        """Job tables of fb_hankel_rt_to_pm_scaled for `n_vec` vector fields (triples r, t, z
        in `scr_f`) followed by scalar fields: (in, in2, sign) per job."""
        import ctypes
        ins, in2, sgn = [], [], []
        for j, f in enumerate(scr_f):
            if j < n_vec and j % 3 == 0:        # p = 0.5 (r - i t)
                ins.append(scr_f[j]); in2.append(scr_f[j + 1]); sgn.append(-1.)
            elif j < n_vec and j % 3 == 1:      # m = 0.5 (r + i t)
                ins.append(scr_f[j - 1]); in2.append(scr_f[j]); sgn.append(+1.)
            else:
                ins.append(f); in2.append(None); sgn.append(0.)
        return ins, in2, (ctypes.c_double * len(sgn))(*sgn)

    def interp2spect(self, fieldtype, fuse_divide_by_volume=False, fuse_filter=False):
        """FFT(z) then DHT(r) of one field group, all modes at once
        (reference: fields.py:313-368 + spectral_transformer.py:157-223).
        `fieldtype` may also be 'EB' (E and B in one batch).  The two optional flags fold
        the divide-by-volume pass (it commutes with the z-FFT) and the spectral filter pass
        into the Hankel GEMM (fb_hankel_scaled) instead of two extra sweeps over the grids;
        with `fuse_divide_by_volume` the interpolation-grid arrays are left un-normalised."""
        self._touch()
        self._need_gpu()
        fi, fs, nf, vec = self._group(fieldtype)
        Nz, Nr = self.Nz, self.Nr
        lib, pa, st = _capi.lib(), _capi.ptr_array, _capi.stream()
        src = self.d_interp[:, fi, :]
        scr = self.d_scratch[:, 0, :]
        fft_exec(src, scr, -1, ncols=nf * Nr)
        scr_f = self._field_views(self.d_scratch, 0, nf)
        out = self._field_views(self.d_spect, fs, nf)
        mats = self._mats['vec_fwd' if vec else 'scal_fwd']
        if fieldtype == 'EB':
            mats = mats + mats
        per = 3 if vec else 1
        mode_of = [(j // per) % self.Nm for j in range(nf)]
        sk = [self.interp[m].d_invvol if fuse_divide_by_volume else None for m in mode_of]
        fz = [self.spect[m].d_filter_array_z if fuse_filter else None for m in mode_of]
        fr = [self.spect[m].d_filter_array_r if fuse_filter else None for m in mode_of]
        if vec:
            # (r, t) -> (p, m) rides along in the operand load of the Hankel GEMM
            ins, in2, sgn = self._rt_pairs(scr_f, nf)
            _capi.check(lib.fb_hankel_rt_to_pm_scaled(
                nf, pa(ins), pa(in2), sgn, self.d_scratch.stride(0), pa(out), self.d_spect.stride(0),
                pa(mats), pa(sk), pa(fz), pa(fr), 1.0, Nz, Nr, st), 'fb_hankel_rt_to_pm_scaled')
        elif fuse_divide_by_volume or fuse_filter:
            _capi.check(lib.fb_hankel_scaled(nf, pa(scr_f), self.d_scratch.stride(0), pa(out), self.d_spect.stride(0),
                                             pa(mats), pa(sk), pa(fz), pa(fr), 1.0, Nz, Nr, st),
                        'fb_hankel_scaled')
        else:
            _capi.check(lib.fb_hankel(nf, pa(scr_f), self.d_scratch.stride(0), pa(out), self.d_spect.stride(0),
                                      pa(mats), 1.0, Nz, Nr, st), 'fb_hankel')

    # ---------------------------------------------------------------- node-major source records
    def source_records(self):
        """complex128[Nz, Nr, 4*Nm] staging array of the in-step deposition: record of node
        (iz, ir) = (J of mode 0: r, t, z | mode 1: r, t, z | ... | rho of mode 0, 1, ...), i.e.
        the field order of the J | rho part of the interpolation slab.  With Nm = 2 a record
        is exactly one 128-B cache line, so the atomics that flush a cell touch 2-4 lines
        instead of 24-48 (see DepGrids in csrc/deposit.hip)."""
        if getattr(self, 'd_src_rec', None) is None:
            t = _capi.torch()
            rec = 4 * self.Nm
            rs = self.Nr * rec + SLAB_PAD
            base = t.zeros(self.Nz * rs, dtype=t.complex128, device=_capi.require_device())
            self.d_src_rec = base.as_strided((self.Nz, self.Nr, rec), (rs, rec, 1))
            self._records_clean = True
        return self.d_src_rec

    def erase_source_records(self):
        """Zero the records for a new deposition; a no-op when the transform that consumed
        them left them zeroed (fb_zfft_from_records_consume).  They count as dirty from here
        on: the caller is about to deposit into them."""
        S = self.source_records()
        if self._records_clean:
            self._records_clean = False
            return
        self._records_clean = False
        row = _capi.torch().as_strided(S, (self.Nz, self.Nr * S.shape[2]), (S.stride(0), 1))
        _capi.check(_capi.lib().fb_erase(1, _capi.ptr_array([row]), S.stride(0), self.Nz,
                                         self.Nr * S.shape[2], _capi.stream()), 'fb_erase')

    def record_views(self, kind):
        """Per-field (Nz, Nr) views of the record array, in the order the deposition wants:
        J -> [m0: Jr, Jt, Jz, m1: ...], rho -> [m0, m1, ...]."""
        S, Nm = self.source_records(), self.Nm
        if kind == 'J':
            return [S[:, :, f] for f in range(3 * Nm)]
        return [S[:, :, 3 * Nm + m] for m in range(Nm)]

    def unpack_source_records(self):
        """Copy the records into the J | rho fields of the interpolation slab (only needed
        when something other than interp2spect_J_and_rho_next wants them)."""
        S, Nm = self.source_records(), self.Nm
        self.d_interp[:, 6 * Nm:10 * Nm, :].copy_(S.permute(0, 2, 1))
        self._records_clean = False

    def spect_cycle_supported(self):
        """Whether the fused forward Hankel + PSATD step + inverse Hankel launch applies to this
        grid (the caller checks the scheme: single domain, standard PSATD, no window shift)."""
        lib = _capi.lib()
        return bool(self.fuse_spectral_cycle and self.v_comoving is None
                    and self.current_correction == 'curl-free'
                    and lib.fb_spect_cycle_supported(self.Nm, self.Nr) and lib.fb_zfft_supported(self.Nz))

    def _src_kz_views(self):
        """Where the z-FFT'd J | rho wait for a pending forward Hankel transform: the J | rho fields
        of the interpolation slab (not otherwise used inside step(): the deposition goes to the
        records, and J, rho on the interpolation grid are rebuilt from spectral space when read)."""
        return self._field_views(self.d_interp, 6 * self.Nm, 4 * self.Nm)

    def interp2spect_J_and_rho_next(self, fuse_filter=False, from_records=False, defer_hankel=False):
        """interp2spect('J') and interp2spect('rho_next') of freshly deposited (un-normalised)
        sources in ONE z-FFT launch and ONE Hankel launch: J and rho are adjacent in the
        interpolation slab, and a Hankel launch takes any list of jobs.  Same arithmetic per
        field as the two separate calls with fuse_divide_by_volume=True."""
        self._touch()
        self._need_gpu()
        Nm, Nz, Nr = self.Nm, self.Nz, self.Nr
        lib, pa, st = _capi.lib(), _capi.ptr_array, _capi.stream()
        nJ, nf = 3 * Nm, 4 * Nm
        if from_records and lib.fb_zfft_supported(Nz):
            # the z-FFT gathers its columns straight from the deposition's records
            S = self.source_records()
            defer_hankel = defer_hankel if (defer_hankel and self.spect_cycle_supported()) else False
            dst = self.d_interp[:, 6 * Nm, :] if defer_hankel else self.d_scratch[:, 0, :]
            dst_rs = self.d_interp.stride(0) if defer_hankel else self.d_scratch.stride(0)
            _capi.check(lib.fb_zfft_from_records_consume(
                Nz, nf, Nr, S.data_ptr(), S.stride(0), S.shape[2], dst.data_ptr(), dst_rs, st),
                'fb_zfft_from_records_consume')
            self._records_clean = True
            if defer_hankel:
                # psatd_step() runs the transform together with the solver step and the inverse
                # transform of E, B (spect_cycle) - or, defer_hankel = 'correct' (decomposed domain),
                # together with the current correction only; anything else first finishes it on its own
                self._pending_hankel = (bool(fuse_filter), defer_hankel)
                return
        elif from_records and lib.fb_fft_generic_from_records_supported(Nz):
            # lengths of the two-sweep generic FFT (4416 = 192 x 23): its head gathers the records
            from .spectral_transform.fourier import generic_scratch
            S = self.source_records()
            scr2 = generic_scratch(Nz, nf * Nr, S.device)
            _capi.check(lib.fb_fft_generic_from_records_consume(
                Nz, nf, Nr, S.data_ptr(), S.stride(0), S.shape[2], self.d_scratch[:, 0, :].data_ptr(),
                self.d_scratch.stride(0), scr2.data_ptr(), scr2.stride(0), st),
                'fb_fft_generic_from_records_consume')
            self._records_clean = True
        else:
            if from_records:
                self.unpack_source_records()
            fft_exec(self.d_interp[:, 6 * Nm, :], self.d_scratch[:, 0, :], -1, ncols=nf * Nr)
        self._hankel_J_and_rho_next(self._field_views(self.d_scratch, 0, nf), fuse_filter)

    def _hankel_J_and_rho_next(self, scr_f, fuse_filter):
        """Forward Hankel transform (with divide-by-volume, (r,t) -> (p,m), filter) of the z-FFT'd
        J | rho_next in the (kz, r) views `scr_f`."""
        Nm, Nz, Nr = self.Nm, self.Nz, self.Nr
        lib, pa, st = _capi.lib(), _capi.ptr_array, _capi.stream()
        nJ, nf = 3 * Nm, 4 * Nm
        out = self._field_views(self.d_spect, 6 * Nm, nJ) + self._field_views(self.d_spect, 10 * Nm, Nm)
        mats = self._mats['vec_fwd'] + self._mats['scal_fwd']
        mode_of = [(j // 3) % Nm for j in range(nJ)] + list(range(Nm))
        sk = [self.interp[m].d_invvol for m in mode_of]
        fz = [self.spect[m].d_filter_array_z if fuse_filter else None for m in mode_of]
        fr = [self.spect[m].d_filter_array_r if fuse_filter else None for m in mode_of]
        ins, in2, sgn = self._rt_pairs(scr_f, nJ)
        _capi.check(lib.fb_hankel_rt_to_pm_scaled(
            nf, pa(ins), pa(in2), sgn, scr_f[0].stride(0), pa(out), self.d_spect.stride(0),
            pa(mats), pa(sk), pa(fz), pa(fr), 1.0, Nz, Nr, st), 'fb_hankel_rt_to_pm_scaled')

    def spect_cycle(self, correct_currents, use_true_rho, only_correct=False):
        """The pending forward Hankel transform of J | rho_next, the solver step
        (correct_currents + push, as psatd_step) and the inverse Hankel transform of the new E, B
        in one launch (fb_spect_cycle_standard); E, B are left in (kz, r) space in the scratch slab
        for spect2interp('EB').  `only_correct` (decomposed domain): the transform and the
        curl-free correction only - the J guard exchange and the push follow."""
        from scipy.constants import c, epsilon_0, mu_0
        Nm = self.Nm
        lib, pa, st = _capi.lib(), _capi.ptr_array, _capi.stream()
        (fuse_filter, _), self._pending_hankel = self._pending_hankel, None
        src = self._src_kz_views()                 # [J m0 r,t,z | J m1 ... | rho m0 | rho m1 ...]
        fields, tables, srcs, outs, fwd, inv = [], [], [], [], [], []
        scr = self._field_views(self.d_scratch, 0, 6 * Nm)
        for m in range(Nm):
            sp, tb = self.spect[m], self.psatd[m].device_tables()
            fields += [getattr(sp, k) for k in SPECT_FIELDS]
            tables += [tb['rho_prev_coef'], tb['rho_next_coef'], tb['j_coef'], tb['C'], tb['S_w'],
                       sp.d_kr, sp.d_kz, sp.d_inv_k2]
            srcs += src[3 * m:3 * m + 3] + [src[3 * Nm + m]]
            outs += scr[3 * m:3 * m + 3] + scr[3 * Nm + 3 * m:3 * Nm + 3 * m + 3]
            fwd += self._mats['vec_fwd'][3 * m:3 * m + 3]
            inv += self._mats['vec_inv'][3 * m:3 * m + 3]
        iv = [self.interp[m].d_invvol for m in range(Nm)]
        fz = [self.spect[m].d_filter_array_z for m in range(Nm)] if fuse_filter else None
        fr = [self.spect[m].d_filter_array_r for m in range(Nm)] if fuse_filter else None
        rc = lib.fb_spect_cycle_standard(
            Nm, pa(srcs), self.d_interp.stride(0), pa(iv), pa(fwd), pa(inv),
            pa(fz) if fz else None, pa(fr) if fr else None, pa(fields), self.d_spect.stride(0), pa(tables),
            self.dt, 2 if only_correct else int(bool(correct_currents)), int(bool(use_true_rho)),
            c, epsilon_0, mu_0, pa(outs), self.d_scratch.stride(0), self.Nz, self.Nr, st)
        _capi.check(rc, 'fb_spect_cycle_standard')
        self.spect_cycle_launches += 1
        self._EB_in_kz_r = not only_correct

    def spect2interp(self, fieldtype):
        """inverse DHT(r) then inverse FFT(z) (reference: fields.py:370-429)."""
        if fieldtype == 'EB' and self._EB_in_kz_r and self._pending_hankel is None:
            # spect_cycle has already applied the inverse Hankel matrices
            self._touch(keep_pending=True)
            self._EB_in_kz_r = False
            self._backward_zfft('EB')
            return
        self._touch()
        self._need_gpu()
        fi, fs, nf, vec = self._group(fieldtype)
        Nz, Nr = self.Nz, self.Nr
        lib, pa, st = _capi.lib(), _capi.ptr_array, _capi.stream()
        inp = self._field_views(self.d_spect, fs, nf)
        scr_f = self._field_views(self.d_scratch, 0, nf)
        mats = self._mats['vec_inv' if vec else 'scal_inv']
        if fieldtype == 'EB':
            mats = mats + mats
        _capi.check(lib.fb_hankel(nf, pa(inp), self.d_spect.stride(0), pa(scr_f), self.d_scratch.stride(0),
                                  pa(mats), 1.0, Nz, Nr, st), 'fb_hankel')
        self._backward_zfft(fieldtype)

    def _backward_zfft(self, fieldtype):
        """Second half of spect2interp: (kz, r) fields 0 .. nf-1 of the scratch slab -> the
        interpolation grid."""
        fi, fs, nf, vec = self._group(fieldtype)
        Nz, Nr = self.Nz, self.Nr
        lib, pa, st = _capi.lib(), _capi.ptr_array, _capi.stream()
        scr_f = self._field_views(self.d_scratch, 0, nf)
        if vec and lib.fb_zfft_supported(Nz):
            # (p, m) -> (r, t) rides along in the first pass of the backward z-FFT
            _capi.check(lib.fb_zfft_pm_to_rt(Nz, nf * Nr, self.d_scratch[:, 0, :].data_ptr(),
                                             self.d_scratch.stride(0),
                                             self.d_interp[:, fi, :].data_ptr(),
                                             self.d_interp.stride(0), Nr, st), 'fb_zfft_pm_to_rt')
            return
        if vec:
            p = pa(scr_f[0::3])
            mm = pa(scr_f[1::3])
            _capi.check(lib.fb_pm_to_rt(nf // 3, p, mm, p, mm, self.d_scratch.stride(0), Nz, Nr, st),
                        'fb_pm_to_rt')
        fft_exec(self.d_scratch[:, 0, :], self.d_interp[:, fi, :], +1, ncols=nf * Nr)

    def spect2partial_interp(self, fieldtype, to_scratch=False):
        """inverse FFT only: spectral (p,m,z) -> interpolation-grid storage (r,t,z slots),
        reference fields.py:431-483.  `to_scratch`: the z-real / r-spectral fields go to the
        scratch slab instead (fields 0 .. nf-1), for `partial2interp`."""
        self._touch()
        self._need_gpu()
        fi, fs, nf, _ = self._group(fieldtype)
        dst = self.d_scratch[:, 0, :] if to_scratch else self.d_interp[:, fi, :]
        fft_exec(self.d_spect[:, fs, :], dst, +1, ncols=nf * self.Nr)

    def partial_interp2spect(self, fieldtype, from_scratch=False):
        """forward FFT only, reference fields.py:485-536."""
        self._touch()
        self._need_gpu()
        fi, fs, nf, _ = self._group(fieldtype)
        src = self.d_scratch[:, 0, :] if from_scratch else self.d_interp[:, fi, :]
        fft_exec(src, self.d_spect[:, fs, :], -1, ncols=nf * self.Nr)

    def partial2interp(self, fieldtype, rows=None):
        """z-real / r-spectral (p, m, z) fields in the scratch slab -> interpolation grid:
        inverse Hankel transform with (p, m) -> (r, t) folded into the GEMM, one launch.
        Equals partial_interp2spect followed by spect2interp (the z-FFT round trip is the
        identity, main.py:741-766) without the two FFTs.  `rows` = (first, last+1): only that
        block of z rows (the transform is local in z: the rows a pending guard-cell exchange
        does not touch can go first, the exchanged ones when they have arrived)."""
        self._need_gpu()
        fi, _, nf, vec = self._group(fieldtype)
        assert vec and nf <= self.NFx
        r0, r1 = (0, self.Nz) if rows is None else rows
        if r1 <= r0:
            return
        lib, pa, st = _capi.lib(), _capi.ptr_array, _capi.stream()
        scr_f = [v[r0:r1] for v in self._field_views(self.d_scratch, 0, nf)]
        out = [v[r0:r1] for v in self._field_views(self.d_interp, fi, nf)]
        mats = self._mats['vec_inv']
        if fieldtype == 'EB':
            mats = mats + mats
        ins, in2, o1, o2, m1, m2 = [], [], [], [], [], []
        for g in range(0, nf, 3):
            # pair job: (p, m) -> (r, t); then the z component as a plain job
            ins += [scr_f[g], scr_f[g + 2]]
            in2 += [scr_f[g + 1], None]
            o1 += [out[g], out[g + 2]]
            o2 += [out[g + 1], None]
            m1 += [mats[g], mats[g + 2]]
            m2 += [mats[g + 1], None]
        _capi.check(lib.fb_hankel_pm_to_rt(
            len(ins), pa(ins), pa(in2), self.d_scratch.stride(0), pa(o1), pa(o2),
            self.d_interp.stride(0), pa(m1), pa(m2), 1.0, r1 - r0, self.Nr, st), 'fb_hankel_pm_to_rt')

    # ---------------------------------------------------------------- solver steps
    def push(self, use_true_rho=False, check_exchanges=False):
        self._touch()
        self._need_gpu()
        if check_exchanges:
            assert self.exchanged_source['J'] is True
            if use_true_rho:
                assert self.exchanged_source['rho_prev'] is True
                assert self.exchanged_source['rho_next'] is True
        for m in range(self.Nm):
            self.spect[m].push_eb_with(self.psatd[m], use_true_rho)
            self.spect[m].push_rho()

    def psatd_step(self, correct_currents=True, use_true_rho=False, only_correct=False, n_move=0):
        """correct_currents() + push() for all modes in ONE launch (the three updates are
        cell-local).  Used by Simulation.step on a single domain, where no guard-cell
        exchange of J separates the correction from the push (main.py:530-542).  On a
        decomposed domain the step calls it twice around that exchange: `only_correct`
        (correction of all modes, one launch), then correct_currents=False (push + rho
        shift of all modes, one launch).  `n_move` != 0: the moving window's translation of
        E, B, rho_prev and J by n_move cells rides along (fb_psatd_step_standard_shift)."""
        if self._pending_hankel is not None and not n_move and not self._EB_in_kz_r \
                and (self._pending_hankel[1] == 'correct') == bool(only_correct):
            self._touch(keep_pending=True)
            self._need_gpu()
            self.spect_cycle(correct_currents, use_true_rho, only_correct=only_correct)
            return
        self._touch()
        self._need_gpu()
        from scipy.constants import c, epsilon_0, mu_0
        fields, tables = [], []
        for m in range(self.Nm):
            sp, tb = self.spect[m], self.psatd[m].device_tables()
            fields += [getattr(sp, k) for k in SPECT_FIELDS]
            tables += [tb['rho_prev_coef'], tb['rho_next_coef'], tb['j_coef'], tb['C'], tb['S_w'],
                       sp.d_kr, sp.d_kz, sp.d_inv_k2]
        if n_move:
            rc = _capi.lib().fb_psatd_step_standard_shift(
                self.Nm, _capi.ptr_array(fields), self.d_spect.stride(0), _capi.ptr_array(tables),
                self.dt, int(bool(correct_currents)), int(bool(use_true_rho)), c, epsilon_0, mu_0,
                self.Nz, self.Nr, _capi.ptr(self.spect[0].d_field_shift), int(n_move), _capi.stream())
            _capi.check(rc, 'fb_psatd_step_standard_shift')
            return
        rc = _capi.lib().fb_psatd_step_standard(
            self.Nm, _capi.ptr_array(fields), self.d_spect.stride(0), _capi.ptr_array(tables),
            self.dt, 2 if only_correct else int(bool(correct_currents)), int(bool(use_true_rho)),
            c, epsilon_0, mu_0,
            self.Nz, self.Nr, _capi.stream())
        _capi.check(rc, 'fb_psatd_step_standard')

    def correct_currents(self, check_exchanges=False):
        self._touch()
        self._need_gpu()
        if check_exchanges:
            assert self.exchanged_source['rho_prev'] is False
            assert self.exchanged_source['rho_next'] is False
            assert self.exchanged_source['J'] is False
            if self.current_correction == 'cross-deposition':
                assert self.exchanged_source['rho_next_xy'] is False
                assert self.exchanged_source['rho_next_z'] is False
        for m in range(self.Nm):
            self.spect[m].correct_currents(self.dt, self.psatd[m], self.current_correction)

    def correct_divE(self):
        raise NotImplementedError('correct_divE is outside the fbpic_amd hot path')

    def erase(self, fieldtype):
        """Zero a field group on the interpolation grid, all modes in one launch."""
        self._touch()
        self._need_gpu()
        if fieldtype not in ('E', 'B', 'J', 'rho', 'J+rho'):
            raise ValueError('Invalid string for fieldtype: %s' % fieldtype)
        if fieldtype not in ('E', 'B'):
            self.materialize_sources()       # a direct deposit outside step(): keep the other group
        if fieldtype == 'J+rho':      # adjacent in the slab: both source groups in one launch
            fi, nf = 6 * self.Nm, 4 * self.Nm
        else:
            fi, _, nf, _ = self._group('rho_prev' if fieldtype == 'rho' else fieldtype)
        views = self._field_views(self.d_interp, fi, nf)
        _capi.check(_capi.lib().fb_erase(nf, _capi.ptr_array(views), self.d_interp.stride(0),
                                         self.Nz, self.Nr, _capi.stream()), 'fb_erase')

    def sum_reduce_deposition_array(self, fieldtype):
        """No-op: the HIP deposition accumulates directly into the grid (LDS-privatised
        tiles), there are no per-thread copies to reduce (reference fields.py:566-593)."""
        return

    def filter_spect(self, fieldtype):
        self._touch()
        self._need_gpu()
        for m in range(self.Nm):
            self.spect[m].filter(fieldtype)

    def divide_by_volume(self, fieldtype):
        self._touch()
        self._need_gpu()
        for m in range(self.Nm):
            self.interp[m].divide_by_volume(fieldtype)
