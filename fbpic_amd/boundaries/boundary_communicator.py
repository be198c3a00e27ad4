"""Domain decomposition in z, guard-cell exchange and particle exchange.

Mirrors the part of the reference's BoundaryCommunicator that the PIC cycle touches
(fbpic/boundaries/boundary_communicator.py:28-826): decomposition arithmetic
(`get_Nz_and_iz`, `get_zmin_zmax`, `divide_into_domain`), `exchange_period`,
`exchange_fields` (E/B replace, J/rho add), `exchange_particles`.

One process per GPU.  Ranks talk through `torch.distributed` point-to-point send/recv
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" for the CPU tests): nearest-neighbour
only, no collective on the data path.  Because the device field slabs are z-major
(fields.py), the guard region of *all* components and modes of a field group is a single
contiguous block: a message is one `slab[z0:z1, f0:f1, :]` slice, no pack kernels.
"""
import ctypes
import os
import numpy as np
from scipy.constants import c
from .. import _capi
from ..fields.utility_methods import get_stencil_reach


def _dist():
    import torch.distributed as dist
    return dist



_GPU_SELECTED = False


def select_gpu_for_this_rank():
    """One rank per GPU (the reference's mpi_select_gpus, fbpic/utils/cuda.py:60-99, called when
    main.py is imported): bind this process to GPU LOCAL_RANK % device_count unless the script
    already chose a device - i.e. a HIP context exists (torch.cuda.set_device / any device
    allocation happened before the Simulation was built; set_device(0) counts) - or
    FBPIC_AMD_NO_GPU_BINDING=1 is set.  Under the `nccl` (= RCCL) backend two ranks of one host
    on the same GPU cannot exchange: that is reported here instead of as an RCCL hang; the
    (host, gpu) pairs travel over the rendezvous store (TCP, host side), not over RCCL, which
    could itself hang in exactly that situation.  (gloo - the CPU tests, several ranks sharing
    the single GPU of a test box - is exempt.)"""
    global _GPU_SELECTED
    if _GPU_SELECTED:
        return
    _GPU_SELECTED = True
    import socket
    t = _capi.torch()
    dist = _dist()
    if not t.cuda.is_available():
        return
    ndev = t.cuda.device_count()
    local_rank = os.environ.get('LOCAL_RANK')
    opt_out = os.environ.get('FBPIC_AMD_NO_GPU_BINDING', '0') == '1'
    if local_rank is not None and ndev > 1 and not t.cuda.is_initialized() and not opt_out:
        t.cuda.set_device(int(local_rank) % ndev)
    dev = t.cuda.current_device()
    _capi.check(_capi.lib().fb_set_device(dev), 'fb_set_device')
    if dist.get_backend() == 'nccl':
        mine = (socket.gethostname(), dev)
        everyone = _host_side_all_gather(dist, mine)
        if len(set(everyone)) != len(everyone):
            raise _capi.BackendError(
                'Two ranks are bound to the same GPU %s (all ranks: %s).  Launch one process per '
                'GPU (torchrun sets LOCAL_RANK) or call torch.cuda.set_device before building the '
                'Simulation.' % (mine, everyone))


def _host_side_all_gather(dist, obj):
    """all_gather of a small picklable object through the process group's key-value store (the
    TCP rendezvous): no device communicator is involved.  Falls back to all_gather_object when
    the group exposes no store (fake groups of the profiling tools)."""
    import pickle
    store = None
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    if store is None:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    rank, size = dist.get_rank(), dist.get_world_size()
    store.set('fbpic_amd/gpu_of_rank/%d' % rank, pickle.dumps(obj))
    return [pickle.loads(store.get('fbpic_amd/gpu_of_rank/%d' % r)) for r in range(size)]


class BoundaryCommunicator(object):
    def __init__(self, Nz, zmin, zmax, Nr, rmax, Nm, dt, v_comoving, use_galilean,
                 boundaries, n_order, n_guard, n_damp, cdt_over_dr, n_inject=None,
                 exchange_period=None, use_all_mpi_ranks=True):
        self.Nm = Nm
        self._Nr = Nr
        self._Nz_global_domain = Nz
        self._zmin_global_domain = zmin
        self.dz = (zmax - zmin) / Nz
        self.dr = rmax / Nr
        if type(boundaries) is str:
            boundaries = {'z': boundaries, 'r': 'reflective'}
        elif type(boundaries) is not dict or 'z' not in boundaries or 'r' not in boundaries:
            raise ValueError("The argument `boundaries` should be a dictionary "
                             "whose keys are 'z' and 'r'.")
        if boundaries['z'] not in ['periodic', 'open']:
            raise ValueError("Unrecognized `boundaries['z']`: '%s'" % boundaries['z'])
        if boundaries['r'] not in ['reflective', 'open']:
            raise ValueError("Unrecognized `boundaries['r']`: '%s'" % boundaries['r'])
        if boundaries['r'] == 'open':
            raise NotImplementedError("boundaries['r']='open' (PML) is outside the fbpic_amd scope")
        self.boundaries = boundaries
        self.use_all_mpi_ranks = use_all_mpi_ranks
        # one rank per GPU: torch.distributed takes the place of mpi4py
        dist = _dist()
        if use_all_mpi_ranks and dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank()
            self.size = dist.get_world_size()
        else:
            self.rank, self.size = 0, 1
        if self.size > 1:
            select_gpu_for_this_rank()
        self.mpi_comm = None
        self.left_proc = self.rank - 1
        self.right_proc = self.rank + 1
        if boundaries['z'] == 'periodic':
            if self.rank == 0:
                self.left_proc = self.size - 1
            if self.rank == self.size - 1:
                self.right_proc = 0
        else:
            if self.rank == 0:
                self.left_proc = None
            if self.rank == self.size - 1:
                self.right_proc = None
        # guard cells (boundary_communicator.py:229-254)
        if n_guard is None:
            if n_order == -1:
                self.n_guard = 64
                if self.size != 1:
                    raise ValueError(
                        'When running with domain decomposition, you need to set the argument '
                        '`n_order` of the `Simulation` object to a positive value (e.g. 32).')
            else:
                self.n_guard = get_stencil_reach(self._Nz_global_domain, self.dz, c * dt,
                                                 n_order, v_comoving, use_galilean) + 1
        else:
            self.n_guard = n_guard
        if boundaries['z'] == 'periodic' and self.size == 1:
            self.n_guard = 0
        self.nz_damp = n_damp['z']
        self.nr_damp = 0
        if boundaries['z'] == 'periodic':
            self.nz_damp = 0
            self.n_inject = 0
        else:
            self.n_inject = int(self.n_guard / 2) if n_inject is None else n_inject
        self.use_pml = False
        # particle exchange period (boundary_communicator.py:281-298)
        if exchange_period is None:
            cells_per_step = 2. * c * dt / self.dz
            self.exchange_period = int(((self.n_guard / 2) - 3) / cells_per_step)
            if self.size == 1 and boundaries['z'] == 'periodic':
                self.exchange_period = 1
            if self.exchange_period < 1:
                raise ValueError('Guard region size is too small for chosen timestep.')
        else:
            self.exchange_period = exchange_period
        self.moving_win = None
        if (self.nz_damp + self.n_inject) > 0:
            if self.left_proc is None:
                self.left_damp = self.generate_damp_array(self.n_guard, self.nz_damp, self.n_inject)
            if self.right_proc is None:
                self.right_damp = self.generate_damp_array(self.n_guard, self.nz_damp, self.n_inject)
        self.d_left_damp = None
        self.d_right_damp = None
        self._guard_bufs = {}
        # transport of device buffers between ranks: 'rccl' = fb_exchange, RCCL send/recv inside
        # the library on the compute stream (one ctypes call per exchange, no host
        # synchronisation; default when torch.distributed runs on the nccl backend, i.e. one GPU
        # per rank), 'torch' = torch.distributed point-to-point (batch_isend_irecv; the only
        # choice under gloo, which stages through the host)
        default = 'torch'
        if self.size > 1 and dist.is_initialized() and dist.get_backend() == 'nccl':
            default = 'rccl'
        self.transport = os.environ.get('FBPIC_AMD_TRANSPORT', default)
        self._rccl_comm = None

    # ---------------------------------------------------------------- decomposition
    def divide_into_domain(self):
        zmin_l, zmax_l = self.get_zmin_zmax(local=True, with_damp=True, with_guard=True,
                                            rank=self.rank)
        Nz_l, _ = self.get_Nz_and_iz(local=True, with_damp=True, with_guard=True, rank=self.rank)
        if Nz_l < 4 * self.n_guard:
            raise ValueError('The boundary guard region is larger than the physical domain size. '
                             'Use fewer ranks or a smaller order of the field solver.')
        return zmin_l, zmax_l, Nz_l

    def get_Nr(self, with_damp):
        return self._Nr + self.nr_damp if with_damp else self._Nr

    def get_rmax(self, with_damp):
        return (self._Nr + self.nr_damp) * self.dr if with_damp else self._Nr * self.dr

    def get_Nz_and_iz(self, local, with_damp, with_guard, rank=None):
        """Number of cells and index of the first cell (counted from the first physical
        cell of the global domain) of the global or of a rank-local grid
        (boundary_communicator.py:399-473)."""
        if local and rank is None:
            raise ValueError('For a local number of cells, the rank considered is needed.')
        if local:
            per = int(self._Nz_global_domain / self.size)
            Nz = per
            iz = rank * per
            if rank == self.size - 1:
                Nz += self._Nz_global_domain % self.size
            if with_damp:
                if rank == 0:
                    Nz += self.nz_damp + self.n_inject
                    iz -= self.nz_damp + self.n_inject
                if rank == self.size - 1:
                    Nz += self.nz_damp + self.n_inject
            if with_guard:
                Nz += 2 * self.n_guard
                iz -= self.n_guard
        else:
            Nz = self._Nz_global_domain
            iz = 0
            if with_damp:
                Nz += 2 * (self.nz_damp + self.n_inject)
                iz -= self.nz_damp + self.n_inject
            if with_guard:
                Nz += 2 * self.n_guard
                iz -= self.n_guard
        return Nz, iz

    def get_zmin_zmax(self, local, with_damp, with_guard, rank=None):
        Nz, iz0 = self.get_Nz_and_iz(local=local, with_damp=with_damp, with_guard=with_guard,
                                     rank=rank)
        zmin = self._zmin_global_domain + iz0 * self.dz
        return zmin, zmin + Nz * self.dz

    def shift_global_domain_positions(self, z_shift):
        self._zmin_global_domain += z_shift

    def move_grids(self, fld, ptcl, dt, time, spect_shifted_by=None):
        """Advance the moving window (boundary_communicator.py:533-553)."""
        self.moving_win.move_grids(fld, ptcl, self, time, spect_shifted_by)

    # ---------------------------------------------------------------- damping (open z)
    def generate_damp_array(self, n_guard, nz_damp, n_inject):
        """Damping profile of the open-z boundary: zero over the outer n_guard + n_inject
        cells, sin^2 rise over nz_damp/2 cells, then 1 (boundary_communicator.py:909-945)."""
        i_cell = np.arange(n_guard + nz_damp + n_inject)
        i0 = n_guard + n_inject
        rise = np.sin((i_cell - i0) * np.pi / (2 * nz_damp / 2.))**2
        damp = np.where(i_cell < i0 + nz_damp / 2., rise, 1.)
        return np.where(i_cell < i0, 0., damp)

    def damp_EB_open_boundary(self, interp, slab=None):
        """Multiply E and B by the damping profile in the damp cells of the end ranks
        (boundary_communicator.py:828-907).  No-op for periodic z."""
        if self.nz_damp == 0:
            return
        t = _capi.torch()
        names = ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz')
        owner = getattr(interp[0], '_owner', None)
        base = slab
        slab = None
        if owner is not None and owner.data_is_on_gpu and len(interp) == owner.Nm:
            # E and B of all modes are the first 6*Nm fields of the z-major slab: one
            # multiplication per end instead of 6*Nm
            if base is None:
                base = owner.d_interp
            slab = base[:, 0:6 * owner.Nm, :]
        if slab is not None:
            # one launch for both ends (cuda_damp_EB_left / cuda_damp_EB_right)
            dev = interp[0].Er.device
            if self.left_proc is None and self.d_left_damp is None:
                self.d_left_damp = t.as_tensor(self.left_damp, device=dev)
            if self.right_proc is None and self.d_right_damp is None:
                self.d_right_damp = t.as_tensor(self.right_damp[::-1].copy(), device=dev)
            dl = self.d_left_damp if self.left_proc is None else None
            dr = self.d_right_damp if self.right_proc is None else None
            rc = _capi.lib().fb_damp_rows(
                _capi.ptr(slab), base.stride(0), slab.shape[1] * slab.shape[2],
                _capi.ptr(dl), 0 if dl is None else dl.shape[0],
                _capi.ptr(dr), 0 if dr is None else dr.shape[0], slab.shape[0], _capi.stream())
            _capi.check(rc, 'fb_damp_rows')
            return
        # per-array path (a subset of the modes, or arrays that are not the Fields slab)
        if not hasattr(interp[0].Er, 'is_cuda'):
            raise _capi.BackendError('damp_EB_open_boundary: the fields are host arrays; fbpic_amd '
                                     'only computes on the GPU (send_fields_to_gpu() first).')
        dev = interp[0].Er.device
        if self.left_proc is None:
            if self.d_left_damp is None:
                self.d_left_damp = t.as_tensor(self.left_damp, device=dev)
            nd = self.d_left_damp.shape[0]
            for g in interp:
                for k in names:
                    getattr(g, k)[:nd, :] *= self.d_left_damp[:, None]
        if self.right_proc is None:
            if self.d_right_damp is None:
                self.d_right_damp = t.as_tensor(self.right_damp[::-1].copy(), device=dev)
            nd = self.d_right_damp.shape[0]
            for g in interp:
                for k in names:
                    getattr(g, k)[-nd:, :] *= self.d_right_damp[:, None]

    def rows_untouched_by_EB_exchange(self, Nz):
        """(lo, hi): the z rows [lo, hi) of a local grid that neither the 'replace' exchange of
        the guard cells nor the open-boundary damping modifies."""
        lo = self.n_guard if self.left_proc is not None else \
            (self.n_guard + self.nz_damp + self.n_inject if self.nz_damp else 0)
        hi = Nz - (self.n_guard if self.right_proc is not None else
                   (self.n_guard + self.nz_damp + self.n_inject if self.nz_damp else 0))
        return lo, max(hi, lo)

    # ---------------------------------------------------------------- field exchange
    def exchange_fields(self, interp, fldtype, method, slab=None):
        """Guard-cell exchange with the two z neighbours (boundary_communicator.py:556-671):
        'replace': my guard cells [0,ng) / [Nz-ng,Nz) <- neighbour's valid [Nz-2ng,Nz-ng) /
        [ng,2ng);  'add': my [0,2ng) / [Nz-2ng,Nz) += neighbour's [Nz-2ng,Nz) / [0,2ng)."""
        if self.size == 1:
            return
        ng = self.n_guard
        # 'EB': E and B in ONE message per neighbour (they are adjacent in the slab and always
        # exchanged together, main.py:745-746) - half the point-to-point latency per step
        names = {'E': ('Er', 'Et', 'Ez'), 'B': ('Br', 'Bt', 'Bz'), 'J': ('Jr', 'Jt', 'Jz'),
                 'EB': ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz'), 'rho': ('rho',)}[fldtype]
        t = _capi.torch()
        Nz = getattr(interp[0], names[0]).shape[0]
        if method == 'replace':
            s_l = slice(ng, 2 * ng); s_r = slice(Nz - 2 * ng, Nz - ng)
            d_l = slice(0, ng); d_r = slice(Nz - ng, Nz)
        elif method == 'add':
            s_l = slice(0, 2 * ng); s_r = slice(Nz - 2 * ng, Nz)
            d_l, d_r = s_l, s_r
        else:
            raise ValueError('Unknown method: %s' % method)
        owner = getattr(interp[0], '_owner', None)
        has_l, has_r = self.left_proc is not None, self.right_proc is not None
        if owner is not None and owner.data_is_on_gpu and len(interp) == owner.Nm:
            # z-major slab: the whole group (all modes and components) is `nf * Nr` adjacent
            # values of every z row -> ONE launch packs both message buffers, one unpacks them
            # (the reference's copy_*_to_gpu_buffer / replace_* / add_*_from_gpu_buffer)
            f0, _, nf, _ = owner._group('rho_prev' if fldtype == 'rho' else fldtype)
            base = owner.d_interp
            if slab is not None:            # the group sits at fields 0 .. nf-1 of `slab`
                base, f0 = slab, 0
            region = base[:, f0:f0 + nf, :]
            nrows, ncontig = s_l.stop - s_l.start, nf * region.shape[2]
            if method == 'replace' and slab is not None and nf == base.shape[1] and f0 == 0 \
                    and base.stride(1) == base.shape[2] and base.stride(2) == 1:
                # The group is the whole slab: a block of z rows is ONE contiguous piece of memory
                # (row stride included), so the messages are sent from and received into the
                # rows themselves - no pack / unpack launch, no message buffers
                rs = base.stride(0)

                def rows(sl, present):
                    if not present:
                        return None
                    return t.as_strided(base, ((sl.stop - sl.start) * rs,), (1,),
                                        base.storage_offset() + sl.start * rs)
                self.exchange_domains(rows(s_l, has_l), rows(s_r, has_r), rows(d_l, has_l), rows(d_r, has_r))
                return
            key = (fldtype, method, nrows, ncontig)
            bufs = self._guard_bufs.get(key)
            if bufs is None:        # persistent message buffers: stable addresses for RCCL
                bufs = [t.empty((nrows, ncontig), dtype=t.complex128, device=region.device)
                        if side else None for side in (has_l, has_r, has_l, has_r)]
                self._guard_bufs[key] = bufs
            send_l, send_r, recv_l, recv_r = bufs
            lib, p, st = _capi.lib(), _capi.ptr, _capi.stream()
            rs = base.stride(0)
            _capi.check(lib.fb_guard_buffers(0, p(region), rs, ncontig, s_l.start, s_r.start, nrows,
                                             p(send_l), p(send_r), st), 'fb_guard_buffers')
            self.exchange_domains(send_l, send_r, recv_l, recv_r)
            _capi.check(lib.fb_guard_buffers(1 if method == 'replace' else 2, p(region), rs, ncontig,
                                             d_l.start, d_r.start, nrows, p(recv_l), p(recv_r), st),
                        'fb_guard_buffers')
            return
        targets = [getattr(g, k) for g in interp for k in names]
        if not all(hasattr(a, 'is_cuda') for a in targets):
            raise _capi.BackendError('exchange_fields: the fields are on the host; fbpic_amd only '
                                     'computes on device (or, in the CPU tests, torch) tensors.')
        send_l = t.stack([a[s_l] for a in targets]).contiguous() if has_l else None
        send_r = t.stack([a[s_r] for a in targets]).contiguous() if has_r else None
        recv_l = t.empty_like(send_l) if has_l else None
        recv_r = t.empty_like(send_r) if has_r else None
        self.exchange_domains(send_l, send_r, recv_l, recv_r)
        for i, a in enumerate(targets):
            if has_l:
                if method == 'replace':
                    a[d_l] = recv_l[i]
                else:
                    a[d_l] += recv_l[i]
            if has_r:
                if method == 'replace':
                    a[d_r] = recv_r[i]
                else:
                    a[d_r] += recv_r[i]

    def exchange_domains(self, send_left, send_right, recv_left, recv_right, skip_empty=False):
        """Nearest-neighbour exchange (boundary_communicator.py:674-707) as one batch of
        point-to-point operations.  Complex tensors travel as their real view.  With
        `skip_empty`, zero-length messages are not posted (both sides know the lengths)."""
        if self.left_proc is None and self.right_proc is None:
            return
        dist = _dist()
        t = _capi.torch()
        if self.transport == 'rccl' and all(x is None or x.is_cuda for x in
                                            (send_left, send_right, recv_left, recv_right)):
            self._rccl_communicator()          # first call: creation + handshake (may fall back)
            if self.transport == 'rccl':
                self._exchange_rccl(send_left, send_right, recv_left, recv_right)
                return
        # RCCL moves device buffers directly over xGMI.  Under the gloo backend (CPU tests,
        # or several ranks sharing one GPU) device tensors are staged through the host.
        stage = (dist.get_backend() == 'gloo')
        staged = []

        def rv(x, receiving=False):
            y = t.view_as_real(x) if x.is_complex() else x
            if stage and y.is_cuda:
                h = y.cpu()
                if receiving:
                    staged.append((y, h))
                return h
            return y

        def want(x):
            return x is not None and not (skip_empty and x.numel() == 0)
        sends, recvs = [], []
        if self.left_proc is not None:
            if want(send_left):
                sends.append(dist.P2POp(dist.isend, rv(send_left), self.left_proc))
            if want(recv_left):
                recvs.append(dist.P2POp(dist.irecv, rv(recv_left, True), self.left_proc))
        if self.right_proc is not None:
            if want(send_right):
                sends.append(dist.P2POp(dist.isend, rv(send_right), self.right_proc))
            if want(recv_right):
                # a 2-rank periodic ring has the same peer on both sides: messages between
                # one pair of ranks match in posting order, so the receive that pairs with
                # the peer's FIRST send (its send-to-left = my from-right) must come first
                if self.size == 2 and self.left_proc == self.right_proc:
                    recvs.insert(0, dist.P2POp(dist.irecv, rv(recv_right, True), self.right_proc))
                else:
                    recvs.append(dist.P2POp(dist.irecv, rv(recv_right, True), self.right_proc))
        ops = sends + recvs
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for dev_t, host_t in staged:
            dev_t.copy_(host_t)

    def _rccl_communicator(self):
        """Communicator of the library's own transport: rank 0 creates the id, torch.distributed
        (any backend) carries it to the other ranks once."""
        if self._rccl_comm is None:
            lib = _capi.lib()
            dist = _dist()
            buf = ctypes.create_string_buffer(128)
            if self.rank == 0:
                _capi.check(lib.fb_comm_unique_id(buf), 'fb_comm_unique_id')
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0)
            comm = ctypes.c_void_p()
            rc = lib.fb_comm_init(ctypes.c_char_p(box[0]), self.rank, self.size, ctypes.byref(comm))
            self._rccl_comm = comm if rc == 0 else False
            if rc == 0:
                import atexit
                atexit.register(self.close)
            self._rccl_handshake(init_error=(None if rc == 0 else
                                             lib.fb_last_error().decode(errors='replace')))
        return self._rccl_comm

    def close(self):
        """Release the library's RCCL communicator (fb_comm_destroy); also runs at interpreter
        exit.  Safe to call more than once."""
        comm, self._rccl_comm = self._rccl_comm, None
        if comm:
            try:
                _capi.lib().fb_comm_destroy(comm)
            except Exception:
                pass

    def _rccl_handshake(self, init_error=None):
        """One message per neighbour carrying (sender rank, side), checked on arrival: a routing
        mistake of the in-library transport (left / right swapped, same-peer order on a 2-rank
        ring) is caught at start-up instead of corrupting guard cells.  On a mismatch the
        exchange falls back to torch.distributed point-to-point (also RCCL) with a warning."""
        t = _capi.torch()
        dev = _capi.require_device()

        def msg(side):
            return t.tensor([float(self.rank), float(side)], dtype=t.float64, device=dev)
        has_l, has_r = self.left_proc is not None, self.right_proc is not None
        send_l, send_r = (msg(0) if has_l else None), (msg(1) if has_r else None)
        recv_l = t.full((2,), -1., dtype=t.float64, device=dev) if has_l else None
        recv_r = t.full((2,), -1., dtype=t.float64, device=dev) if has_r else None
        # every rank must have a communicator before anyone posts a message
        inits = [None] * self.size
        _dist().all_gather_object(inits, init_error is None)
        ok = all(inits)
        try:
            if ok:
                self._exchange_rccl(send_l, send_r, recv_l, recv_r)
            if has_l:
                ok = ok and recv_l.tolist() == [float(self.left_proc), 1.]
            if has_r:
                ok = ok and recv_r.tolist() == [float(self.right_proc), 0.]
        except _capi.BackendError:
            ok = False
        flags = [None] * self.size
        _dist().all_gather_object(flags, bool(ok))
        if not all(flags):
            if self.rank == 0 or init_error:
                print('fbpic_amd: the in-library RCCL exchange failed its start-up handshake on '
                      'rank(s) %s%s; using torch.distributed point-to-point instead'
                      % ([i for i, f in enumerate(flags) if not f],
                         ' (%s)' % init_error if init_error else ''))
            self.transport = 'torch'

    def _exchange_rccl(self, send_left, send_right, recv_left, recv_right):
        def pb(x):
            return (None, 0) if x is None else (x.data_ptr(), x.numel() * x.element_size())
        for x in (send_left, send_right, recv_left, recv_right):
            assert x is None or x.is_contiguous()
        sl, sr, rl, rr = pb(send_left), pb(send_right), pb(recv_left), pb(recv_right)
        left = -1 if self.left_proc is None else self.left_proc
        right = -1 if self.right_proc is None else self.right_proc
        rc = _capi.lib().fb_exchange(self._rccl_communicator(), left, right, sl[0], sl[1], sr[0], sr[1],
                                     rl[0], rl[1], rr[0], rr[1], _capi.stream())
        _capi.check(rc, 'fb_exchange')

    # ---------------------------------------------------------------- gathering (diagnostics)
    def gather_grid_array(self, array, root=0, with_damp=False):
        """Global array (physical cells of all ranks, z ascending; optionally with the damp
        cells of the two ends) on rank `root`, None elsewhere
        (boundary_communicator.py:1011-1072).  `array`: local (Nz_local, Nr) array or tensor
        including guard and damp cells.  Not on the hot path: host arrays, object gather."""
        Nz_global, iz_start_global = self.get_Nz_and_iz(local=False, with_damp=with_damp,
                                                        with_guard=False)
        Nr = self.get_Nr(with_damp=with_damp)
        Nz_local, iz_start_local_domain = self.get_Nz_and_iz(
            local=True, with_damp=with_damp, with_guard=False, rank=self.rank)
        _, iz_start_local_array = self.get_Nz_and_iz(local=True, with_damp=True, with_guard=True,
                                                     rank=self.rank)
        iz_in_array = iz_start_local_domain - iz_start_local_array
        if hasattr(array, 'detach'):
            array = array.detach().cpu().numpy()
        local_array = np.ascontiguousarray(array[iz_in_array:iz_in_array + Nz_local, :Nr])
        if self.size == 1:
            return local_array
        dist = _dist()
        pieces = [None] * self.size
        dist.all_gather_object(pieces, local_array)
        if self.rank != root:
            return None
        gathered_array = np.zeros((Nz_global, Nr), dtype=local_array.dtype)
        for k, piece in enumerate(pieces):
            _, iz_k = self.get_Nz_and_iz(local=True, with_damp=with_damp, with_guard=False, rank=k)
            gathered_array[iz_k - iz_start_global: iz_k - iz_start_global + piece.shape[0]] = piece
        return gathered_array

    def gather_grid(self, grid, root=0):
        """InterpolationGrid of the global physical domain (no guard, no damp cells) holding
        the gathered fields, on rank `root` (boundary_communicator.py:964-1009)."""
        from ..fields.interpolation_grid import InterpolationGrid, INTERP_FIELDS
        gathered_grid = None
        if self.rank == root:
            Nz_global, _ = self.get_Nz_and_iz(local=False, with_guard=False, with_damp=False)
            zmin_global, zmax_global = self.get_zmin_zmax(local=False, with_guard=False,
                                                          with_damp=False)
            gathered_grid = InterpolationGrid(Nz_global, self.get_Nr(with_damp=False), grid.m,
                                              zmin_global, zmax_global,
                                              self.get_rmax(with_damp=False))
        for field in INTERP_FIELDS:
            gathered_array = self.gather_grid_array(getattr(grid, field), root)
            if self.rank == root:
                setattr(gathered_grid, field, gathered_array)
        return gathered_grid

    # ---------------------------------------------------------------- particle exchange
    def exchange_particles(self, species, fld, time):
        """Single periodic domain: wrap z into [zmin, zmax) (particle_buffer_handling.py:
        514-556).  Decomposed domain: hand the particles that left the local physical
        range to the neighbours (boundary_communicator.py:750-826)."""
        species._touch()
        species.flush_pending_push()
        species._prerank = None
        if self.n_guard == 0:
            rc = _capi.lib().fb_shift_periodic(species.Ntot, _capi.ptr(species.z),
                                               fld.interp[0].zmin, fld.interp[0].zmax,
                                               _capi.stream())
            _capi.check(rc, 'fb_shift_periodic')
        else:
            self.exchange_particles_aperiodic_subdomain(species, fld, time)

    def exchange_particles_aperiodic_subdomain(self, species, fld, time):
        from .particle_buffer_handling import exchange_particles_between_ranks, finish_particle_handover
        begun = species.__dict__.pop('_handover_begun', None)
        if begun is not None:           # the first half is already in the stream (begin_exchange_particles)
            finish_particle_handover(self, species, fld, time, begun)
            return
        exchange_particles_between_ranks(self, species, fld, time)

    def begin_exchange_particles(self, species, fld):
        """First half of the hand-over of a decomposed domain, posted ahead of its place in the step
        (Simulation._step_loop: behind the particle pass of the iteration before): selection, packing,
        the two messages and the request of the host read; `exchange_particles` then completes it.  All
        ranks post it at the same point of the step, so the order of the point-to-point operations is the
        same on both ends of every link."""
        from .particle_buffer_handling import begin_particle_handover
        species._touch()
        species.flush_pending_push()
        species._prerank = None
        species._handover_begun = begin_particle_handover(self, species, fld)
        self.early_handovers = getattr(self, 'early_handovers', 0) + 1      # (diagnostics / tests)
