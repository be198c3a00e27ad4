"""Particle hand-over between neighbouring z-slabs (one rank per GPU).

Semantics of fbpic/boundaries/particle_buffer_handling.py:17-172 (remove_outside_particles)
and :289-417 (add_buffers_to_particles) + boundary_communicator.py:750-826: particles whose
z left the local *physical* range [zmin + ng dz, zmax - ng dz] are removed and sent to the
left / right neighbour (dropped at an open end); received particles are appended as
(from-left | stayed | from-right); particles that wrapped around the periodic box are
shifted by +-L.  ONE ownership rule on every rank and every backend: the z comparison of the
reference's CPU path (the parity target); device tensors only test the few cell rows next to
the box edges (`_select_leaving`).  Runs once every `exchange_period` (~14) steps.
"""
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


def exchange_particles_between_ranks(comm, species, fld, time):
    t = _capi.torch()
    _prime_device_ops(t, species.z.device)
    if species.z.is_cuda and not getattr(species, '_handover_pool_primed', False):
        # message-sized blocks for the caching allocator (send / receive payloads and index
        # tensors of a hand-over): requested from the driver now, not by the first real hand-over
        species._handover_pool_primed = True
        m = max(species.Ntot // 24, 4096)
        blocks = [t.empty((len(_STATE), m), dtype=t.float64, device=species.z.device) for _ in range(4)]
        blocks += [t.empty(2 * m, dtype=t.int64, device=species.z.device) for _ in range(4)]
        del blocks
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
