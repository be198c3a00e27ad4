"""Particle hand-over between neighbouring z-slabs (one rank per GPU).

Semantics of fbpic/boundaries/particle_buffer_handling.py:17-172 (remove_outside_particles)
and :289-417 (add_buffers_to_particles) + boundary_communicator.py:750-826: particles whose
z left the local *physical* range [zmin + ng dz, zmax - ng dz] are removed and sent to the
left / right neighbour (dropped at an open end); received particles are appended as
(from-left | stayed | from-right); particles that wrapped around the periodic box are
shifted by +-L.  ONE ownership rule on every rank and every backend: the z comparison of the
reference's CPU path (the parity target).  Runs once every `exchange_period` (~14) steps.

Device arrays (`_exchange_on_device`): selection, packing and compaction are library kernels
(csrc/handover.hip) and every count stays on the device until the payload has been posted:
one launch selects + packs both sides into fixed-capacity messages whose header carries the
count, ONE exchange, one host read of the four counts, then compaction and append.  Host
tensors (the gloo tests of the transport logic) take the tensor-operation path below
(`_leaving_indices`, counts first, then payloads, as the reference).
"""
import os
from .. import _capi

_STATE = ('x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w')     # reference buffer order
_FIELDS = ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz')


_PRIMED = set()


def _prime_device_ops(t, dev):
    """Rehearse the hand-over once on a few dummy values, through the same functions.  The
    device code of a tensor operation is loaded the first time it runs (tens of ms each, 0.1 s
    and more for the whole hand-over); without this, that cost lands on the first step in which
    particles really cross a boundary - `exchange_period` steps into a run, inside whatever is
    being timed - instead of on the first (warm-up) step."""
    if dev in _PRIMED or dev.type != 'cuda':
        return
    _PRIMED.add(dev)
    # sizes chosen so that the operations take the same code paths as a real hand-over (e.g.
    # index_select switches kernels above 16 indices: rehearsed with 2-7 indices only, the first
    # real hand-over still spent 9 ms loading the large-index variant)
    n = 4096
    arrs = [t.arange(n, dtype=t.float64, device=dev) + i for i in range(2)]
    ps = t.arange(n, dtype=t.int32, device=dev)
    offs = t.stack([ps[i] for i in (40, 100, n - 120, n - 30)]).tolist()
    z = arrs[0]
    idx_l = t.cat((t.arange(0, offs[0], device=dev),
                   offs[0] + t.nonzero(z[offs[0]:offs[1]] < 70.).reshape(-1)))
    idx_r = t.cat((offs[2] + t.nonzero(z[offs[2]:offs[3]] > n - 80.).reshape(-1),
                   t.arange(offs[3], n, device=dev)))
    send = t.stack([a.index_select(0, idx_l) for a in arrs]).contiguous()
    cnt = t.tensor([send.shape[1]], dtype=t.int64, device=dev)
    got = t.zeros(1, dtype=t.int64, device=dev)
    got.copy_(cnt)
    recv = t.empty((2, int(got.item())), dtype=t.float64, device=dev)
    recv.copy_(send)
    recv[1] += 1.
    out, n_new = _compact_and_append(t, arrs, n, idx_l, idx_r, recv, recv[:, :40].contiguous())
    f = _resized(t, out[0], 0, n_new)
    f.zero_()
    _resized(t, out[1], n_new, 4 * n)
    float(out[1].sum().item())


def _leaving_indices(t, species, fld, ng, zbox_min, zbox_max):
    """int64 index tensors of the particles that leave to the left / to the right, by the rule
    of the reference's CPU path (particle_buffer_handling.py:58-172): left if z < zbox_min,
    right if z > zbox_max.

    Cell-sorted device arrays: only the particles of the cell rows next to the two box edges
    can be on either side; everything before / after those rows is known from the per-cell
    prefix sum, so the comparison runs on a few cell rows instead of the whole arrays.
    Unlike the cell-based cut of the reference's GPU path (:177-236), which hands a particle
    over half a cell late, the result is identical to the CPU rule on every rank."""
    z = species.z
    n = species.Ntot
    dev = z.device
    fast = (z.is_cuda and getattr(species, 'use_bin_sort', False) and n > 0)
    if not fast:
        return (t.nonzero(z < zbox_min).reshape(-1), t.nonzero(z > zbox_max).reshape(-1))
    if not species.sorted:
        species.sort_particles(fld=fld)
        species.sorted = True
        z = species.z
    Nz, Nr = fld.Nz, fld.Nr
    shift = species.prefix_sum_shift            # window moves since the sort
    ps = species.prefix_sum

    def row(r):                                  # clamp to [0, Nz]
        return min(max(r, 0), Nz)
    # zbox_min lies in cell row ng, zbox_max in row Nz - ng (iz_upper = ceil(z_cell)); one row
    # of margin on each side absorbs the rounding of the two different expressions
    rows = [row(ng + shift - 1), row(ng + shift + 2), row(Nz - ng + shift - 1), row(Nz - ng + shift + 2)]
    idx = [max(r * (Nr + 1) - 1, 0) for r in rows]
    offs = t.stack([ps[i] for i in idx]).tolist()
    o = [0 if r == 0 else int(v) for r, v in zip(rows, offs)]
    o0, o1, o2, o3 = o[0], max(o[1], o[0]), max(o[2], o[1], o[0]), max(o[3], o[2], o[1], o[0])
    left = [t.arange(0, o0, device=dev)] if o0 > 0 else []
    left.append(o0 + t.nonzero(z[o0:o1] < zbox_min).reshape(-1))
    right = [o2 + t.nonzero(z[o2:o3] > zbox_max).reshape(-1)]
    if o3 < n:
        right.append(t.arange(o3, n, device=dev))
    return (t.cat(left) if len(left) > 1 else left[0], t.cat(right) if len(right) > 1 else right[0])


def _resized(t, a, n_keep, n_new):
    """Length-n_new tensor holding a[:n_keep]: a view of a's own storage when that is large
    enough (the particle arrays are kept with some headroom), else a new allocation with
    ~6 % of headroom."""
    cap = a.untyped_storage().nbytes() // a.element_size() - a.storage_offset()
    if cap >= n_new:
        return t.empty(0, dtype=a.dtype, device=a.device).set_(a.untyped_storage(), a.storage_offset(),
                                                               (n_new,))
    b = t.empty(n_new + n_new // 16 + 1024, dtype=a.dtype, device=a.device)[:n_new]
    b[:n_keep] = a[:n_keep]
    return b


def _compact_and_append(t, arrs, n, idx_l, idx_r, recv_l, recv_r):
    """Remove the particles idx_l, idx_r from the length-n arrays `arrs` and append the rows of
    recv_l, recv_r (one row per array); returns (new arrays, new length).

    Compaction in O(number of movers): the holes left in the first n - n_leave slots are filled
    with the survivors of the last n_leave slots, the arrivals are appended; nothing else is
    copied (the reference rebuilds every array as from-left | stayed | from-right, :289-417;
    only the order of the particles differs, and they are re-sorted before the next deposit)."""
    dev = arrs[0].device
    n_rl, n_rr = recv_l.shape[1], recv_r.shape[1]
    n_leave = int(idx_l.numel() + idx_r.numel())
    m = n - n_leave
    n_new = m + n_rl + n_rr
    src = dst = None
    if n_leave:
        leave = t.cat((idx_l, idx_r))
        in_tail = leave >= m
        tail_free = t.ones(n - m, dtype=t.bool, device=dev)
        tail_free[leave[in_tail] - m] = False
        src = m + t.nonzero(tail_free).reshape(-1)       # survivors sitting in the tail
        dst = leave[~in_tail]                             # holes in the head
    if dev.type == 'cuda':
        lib, st = _capi.lib(), _capi.stream()
        if src is not None and src.numel():
            _capi.check(lib.fb_handover_move(src.numel(), _capi.ptr(src), _capi.ptr(dst), len(arrs),
                                             _capi.ptr_array(arrs), st), 'fb_handover_move')
        out = [_resized(t, a, m, n_new) for a in arrs]
        for buf, first in ((recv_l, m), (recv_r, m + n_rl)):
            if buf.shape[1]:
                assert buf.stride(1) == 1
                _capi.check(lib.fb_handover_append(buf.shape[1], first, len(out), _capi.ptr_array(out),
                                                   _capi.ptr(buf), buf.stride(0), st), 'fb_handover_append')
        return out, n_new
    out = []
    for i, a in enumerate(arrs):
        if src is not None and src.numel():
            a[dst] = a.index_select(0, src)
        b = _resized(t, a, m, n_new)
        if n_rl:
            b[m:m + n_rl] = recv_l[i]
        if n_rr:
            b[m + n_rl:n_new] = recv_r[i]
        out.append(b)
    return out, n_new


HEADER = 8          # FB_HANDOVER_HEADER (include/fbpic_amd.h)
_CAP0 = 16384       # initial capacity (particles) of a hand-over message: 1 MiB; grows on demand


class _Link(object):
    """Persistent message buffers of one neighbour link: capacity `cap` particles in both
    directions.  Both ends of a link see the same two counts after every exchange, so they grow
    the capacity by the same rule at the same time without talking about it."""

    def __init__(self, t, dev, present):
        self.present = present
        self.cap = _CAP0
        self.send = self.recv = None
        if present:
            self._alloc(t, dev)

    def _alloc(self, t, dev):
        n = HEADER + len(_STATE) * self.cap
        self.send = t.zeros(n, dtype=t.float64, device=dev)
        self.recv = t.zeros(n, dtype=t.float64, device=dev)

    def grow_for(self, t, dev, count):
        if count > self.cap:
            self.cap = 1 << int(2 * count - 1).bit_length()
            if self.present:
                self._alloc(t, dev)


def _device_state(t, species, comm, dev):
    st = getattr(species, '_handover', None)
    if st is None or st['dev'] != dev:
        st = {'dev': dev,
              'left': _Link(t, dev, comm.left_proc is not None),
              'right': _Link(t, dev, comm.right_proc is not None),
              'counts': t.zeros(8, dtype=t.int64, device=dev),
              'counts_host': t.zeros(8, dtype=t.int64).pin_memory(),
              'idx': None, 'ws': None}
        species._handover = st
    return st


def _exchange_on_device(comm, species, fld, time):
    """Hand-over of device-resident particles; see the module docstring."""
    t = _capi.torch()
    lib, p, pa, st_ = _capi.lib(), _capi.ptr, _capi.ptr_array, _capi.stream()
    dev = species.z.device
    st = _device_state(t, species, comm, dev)
    L, R = st['left'], st['right']
    g0 = fld.interp[0]
    ng = comm.n_guard
    zbox_min = g0.zmin + ng * g0.dz
    zbox_max = g0.zmax - ng * g0.dz
    n = species.Ntot
    nattr = len(_STATE)
    arrs = [getattr(species, k) for k in _STATE]
    # index lists: a quarter of the particles may leave per side before this gives up
    idx_cap = max(2 * max(L.cap, R.cap), n // 4 + 1024)
    if st['idx'] is None or st['idx'].shape[1] < idx_cap:
        st['idx'] = t.empty((2, idx_cap), dtype=t.int32, device=dev)
    idx_cap = st['idx'].shape[1]
    # cell-sorted arrays with a valid prefix sum: only the cell rows next to the two box edges
    # are compared (zbox_min lies in cell row ng, zbox_max in row Nz - ng; one row of margin on
    # each side absorbs the rounding of the two different expressions); otherwise all of z
    use_prefix = bool(species.sorted and getattr(species, '_prefix_valid', False)
                      and getattr(species, 'use_bin_sort', False) and n > 0)
    cuts = (-1, -1, -1, -1)
    if use_prefix:
        Nz, Nr = fld.Nz, fld.Nr
        shift = species.prefix_sum_shift            # window moves since the sort
        rows = [min(max(r, 0), Nz) for r in (ng + shift - 1, ng + shift + 2,
                                              Nz - ng + shift - 1, Nz - ng + shift + 2)]
        cuts = tuple(r * (Nr + 1) - 1 for r in rows)      # -1 = offset 0
    _capi.check(lib.fb_handover_select_pack(
        n, p(species.z), p(species.prefix_sum) if use_prefix else None, cuts[0], cuts[1], cuts[2], cuts[3],
        zbox_min, zbox_max, nattr, pa(arrs), L.cap, R.cap, idx_cap, p(L.send), p(R.send),
        p(st['idx'][0]), p(st['idx'][1]), p(st['counts']), st_), 'fb_handover_select_pack')
    # ONE exchange of the two fixed-size messages; nothing has been read back so far
    comm._handover_caps = (L.cap, R.cap)       # (profiling tools that stand in for the neighbour)
    comm.exchange_domains(L.send, R.send, L.recv, R.recv)
    comm._handover_caps = None
    _capi.check(lib.fb_handover_recv_counts(p(L.recv), p(R.recv), p(st['counts']), st_),
                'fb_handover_recv_counts')
    st['counts_host'].copy_(st['counts'], non_blocking=True)
    t.cuda.current_stream().synchronize()             # the one host read of a hand-over
    n_sl, n_sr, n_rl, n_rr = [int(v) for v in st['counts_host'][:4].tolist()]
    if max(n_sl, n_sr) > idx_cap:
        raise _capi.BackendError('particle hand-over: %d / %d particles leave the slab of rank %d at '
                                 'once (more than a quarter of its %d particles)'
                                 % (n_sl, n_sr, comm.rank, n))
    # payload of the two sides: views into the received messages ...
    parts = []        # (buffer of rows, row stride, count) in append order: from-left, from-right
    caps = (L.cap, R.cap)
    over = [max(0, c - cap) if link.present else 0
            for c, cap, link in zip((n_sl, n_sr, n_rl, n_rr), caps + caps, (L, R, L, R))]
    if L.present and n_rl:
        parts.append([L.recv[HEADER:], L.cap, min(n_rl, L.cap), 'left'])
    if R.present and n_rr:
        parts.append([R.recv[HEADER:], R.cap, min(n_rr, R.cap), 'right'])
    if any(over):
        # ... plus, when a message was too small, the remainder in a second, exactly sized one
        # (both ends of the link read the same count in the header, so both post it)
        def rest_out(side, link, nsel, extra):
            if not link.present or not extra:
                return None
            idx = st['idx'][side][link.cap:nsel].to(t.int64)
            buf = t.empty((nattr, extra), dtype=t.float64, device=dev)
            _capi.check(lib.fb_handover_pack(extra, p(idx), nattr, pa(arrs), p(buf), buf.stride(0), st_),
                        'fb_handover_pack')
            return buf
        s_l, s_r = rest_out(0, L, n_sl, over[0]), rest_out(1, R, n_sr, over[1])
        r_l = t.empty((nattr, over[2]), dtype=t.float64, device=dev) if (L.present and over[2]) else None
        r_r = t.empty((nattr, over[3]), dtype=t.float64, device=dev) if (R.present and over[3]) else None
        comm.exchange_domains(s_l, s_r, r_l, r_r, skip_empty=True)
        if r_l is not None:
            parts.append([r_l, r_l.stride(0), over[2], 'left'])
        if r_r is not None:
            parts.append([r_r, r_r.stride(0), over[3], 'right'])
    # plasma uncovered by the moving window enters through the right edge of the last rank
    # (boundary_communicator.py:803-808)
    if (comm.moving_win is not None) and (comm.rank == comm.size - 1) and species.continuous_injection:
        new = t.from_numpy(species.generate_continuously_injected_particles(time)).to(dev)
        if new.shape[1]:
            parts.append([new, new.stride(0), new.shape[1], 'injected'])
    if not L.present:
        n_rl = 0
    if not R.present:
        n_rr = 0
    n_in = sum(q[2] for q in parts)
    if n_sl + n_sr == 0 and n_in == 0:
        return                       # nobody crossed a boundary: arrays (and their sort) stay
    # compaction: the leavers' slots are given to survivors of the tail
    n_leave = n_sl + n_sr
    if n_leave:
        need = int(lib.fb_handover_workspace_bytes(n_leave))
        if st['ws'] is None or st['ws'].shape[0] < need:
            st['ws'] = t.empty(2 * need, dtype=t.uint8, device=dev)
        _capi.check(lib.fb_handover_compact(n, n_sl, p(st['idx'][0]), n_sr, p(st['idx'][1]), nattr,
                                            pa(arrs), p(st['ws']), st['ws'].shape[0], st_),
                    'fb_handover_compact')
    m = n - n_leave
    n_new = m + n_in
    out = [_resized(t, a, m, n_new) for a in arrs]
    # periodic wrap of the hand-over across the ends of the global box
    Ltot = comm._Nz_global_domain * comm.dz
    first = m
    iz = _STATE.index('z')
    for buf, stride, cnt, origin in parts:
        shift = 0.
        if origin == 'right' and comm.right_proc == 0:
            shift = Ltot
        elif origin == 'left' and comm.left_proc == comm.size - 1:
            shift = -Ltot
        _capi.check(lib.fb_handover_append_shift(cnt, first, nattr, pa(out), p(buf), stride,
                                                 iz if shift != 0. else -1, shift, st_),
                    'fb_handover_append_shift')
        first += cnt
    for k, b in zip(_STATE, out):
        setattr(species, k, b)
    species.Ntot = n_new
    for k in _FIELDS:
        # E, B on the particles: re-sized only.  Nothing reads them before the next gather
        # writes them (inside step() every read follows a gather that stores; the reference
        # zeroes them here, :289-417) - except for a neutral species, which never gathers
        f = getattr(species, k, None)
        if hasattr(f, 'untyped_storage') and f.device == dev:
            f = _resized(t, f, 0, n_new)
        else:
            f = t.zeros(n_new, dtype=t.float64, device=dev)
        if species.q == 0:
            f.zero_()
        setattr(species, k, f)
    # The arrays stay cell-sorted except for the few particles that were moved / appended: the
    # deposition and the gather work on runs of equal cells and do not need more (any order is
    # correct), and the next fused pass re-sorts everything anyway.  Only the per-cell prefix
    # sum is no longer exact.
    nearly_sorted = bool(species.sorted and (n_leave + n_in) * 16 < max(n_new, 1))
    moved = species._moved_since_sort
    species.on_particle_number_changed()
    species._prefix_valid = False
    if nearly_sorted:
        species.sorted = True
        species._moved_since_sort = moved
    # capacities for the next hand-over (same rule, same numbers on both ends of a link)
    L.grow_for(t, dev, max(n_sl, n_rl))
    R.grow_for(t, dev, max(n_sr, n_rr))


def exchange_particles_between_ranks(comm, species, fld, time):
    if species.z.is_cuda and os.environ.get('FBPIC_AMD_HANDOVER', 'device') == 'device':
        return _exchange_on_device(comm, species, fld, time)
    return _exchange_with_tensor_ops(comm, species, fld, time)


def _exchange_with_tensor_ops(comm, species, fld, time):
    t = _capi.torch()
    _prime_device_ops(t, species.z.device)
    g0 = fld.interp[0]
    ng = comm.n_guard
    zbox_min = g0.zmin + ng * g0.dz
    zbox_max = g0.zmax - ng * g0.dz
    dev = species.z.device
    idx_l, idx_r = _leaving_indices(t, species, fld, ng, zbox_min, zbox_max)
    arrs = [getattr(species, k) for k in _STATE]
    n = species.Ntot

    def pack(idx, proc):
        if proc is None or idx.numel() == 0:
            return t.empty((len(_STATE), 0), dtype=t.float64, device=dev)
        if dev.type == 'cuda':
            # every attribute in one launch (the library's hand-over kernels)
            buf = t.empty((len(_STATE), idx.numel()), dtype=t.float64, device=dev)
            _capi.check(_capi.lib().fb_handover_pack(idx.numel(), _capi.ptr(idx), len(arrs),
                                                     _capi.ptr_array(arrs), _capi.ptr(buf),
                                                     buf.stride(0), _capi.stream()), 'fb_handover_pack')
            return buf
        return t.stack([a.index_select(0, idx) for a in arrs]).contiguous()
    send_l = pack(idx_l, comm.left_proc)
    send_r = pack(idx_r, comm.right_proc)
    # 1) counts, 2) payloads (boundary_communicator.py:782-801)
    n_sl = t.tensor([send_l.shape[1]], dtype=t.int64, device=dev)
    n_sr = t.tensor([send_r.shape[1]], dtype=t.int64, device=dev)
    n_rl = t.zeros(1, dtype=t.int64, device=dev)
    n_rr = t.zeros(1, dtype=t.int64, device=dev)
    comm.exchange_domains(n_sl, n_sr, n_rl, n_rr)
    n_rl, n_rr = int(n_rl.item()), int(n_rr.item())
    recv_l = t.empty((len(_STATE), n_rl), dtype=t.float64, device=dev)
    recv_r = t.empty((len(_STATE), n_rr), dtype=t.float64, device=dev)
    comm.exchange_domains(send_l, send_r, recv_l, recv_r, skip_empty=True)
    # plasma uncovered by the moving window enters through the right edge of the last rank
    # (boundary_communicator.py:803-808)
    if (comm.moving_win is not None) and (comm.rank == comm.size - 1) \
            and species.continuous_injection:
        new = species.generate_continuously_injected_particles(time)
        recv_r = t.from_numpy(new).to(dev)
        n_rr = recv_r.shape[1]
    # periodic wrap of the hand-over across the ends of the global box
    Ltot = comm._Nz_global_domain * comm.dz
    if comm.right_proc == 0 and n_rr:
        recv_r[2] += Ltot
    if comm.left_proc == comm.size - 1 and n_rl:
        recv_l[2] -= Ltot
    if idx_l.numel() + idx_r.numel() == 0 and n_rl == 0 and n_rr == 0:
        return                       # nobody crossed a boundary: arrays (and their sort) stay
    new_arrs, n_new = _compact_and_append(t, arrs, n, idx_l, idx_r, recv_l, recv_r)
    for k, b in zip(_STATE, new_arrs):
        setattr(species, k, b)
    species.Ntot = n_new
    for k in _FIELDS:
        f = getattr(species, k, None)
        if hasattr(f, 'untyped_storage') and f.device == dev:
            f = _resized(t, f, 0, n_new)
        else:
            f = t.empty(n_new, dtype=t.float64, device=dev)
        f.zero_()
        setattr(species, k, f)
    species.sorted = False
    species.on_particle_number_changed()
    if hasattr(species, '_prefix_valid'):
        species._prefix_valid = False
