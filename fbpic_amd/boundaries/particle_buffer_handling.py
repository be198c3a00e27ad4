"""Particle hand-over between neighbouring z-slabs (one rank per GPU).

Semantics of fbpic/boundaries/particle_buffer_handling.py:17-172 (remove_outside_particles)
and :289-417 (add_buffers_to_particles) + boundary_communicator.py:750-826: particles whose
z left the local *physical* range [zmin + ng dz, zmax - ng dz] are removed and sent to the
left / right neighbour (dropped at an open end); received particles are appended as
(from-left | stayed | from-right); particles that wrapped around the periodic box are
shifted by +-L.  ONE ownership rule on every rank and every backend: the z comparison of the
reference's CPU path (the parity target).  Runs once every `exchange_period` (~14) steps.

Selection, packing and compaction are library kernels (csrc/handover.hip) and every count stays
on the device until the payload has been posted: one launch selects + packs both sides into
fixed-capacity messages whose header carries the count, ONE exchange, one host read of the four
counts, then compaction and append; a message that was too small is followed by an exactly
sized remainder.  The protocol is written once (`exchange_particles_between_ranks`); its five
data movements dispatch on the tensors: device tensors -> the library kernels (the product),
host tensors -> the same movements as tensor operations (the second branch of `_select_pack`,
`_recv_counts`, `_pack_rest`, `_compact`, `_append`), which exist for the gloo tests of the
transport logic on CPU (tests/test_multirank_cpu.py) and nothing else.
"""
from scipy.constants import c
from .. import _capi

_STATE = ('x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w')     # reference buffer order
_FIELDS = ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz')


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


HEADER = 8          # FB_HANDOVER_HEADER (include/fbpic_amd.h)
# one-pass iterations since the sort up to which the hand-over scans home rows only (see the device path)
HOME_SCAN_MAX_AGE = 8
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
              'counts_host': (t.zeros(8, dtype=t.int64).pin_memory() if dev.type == 'cuda'
                              else t.zeros(8, dtype=t.int64)),
              'idx': None, 'ws': None}
        species._handover = st
    return st




# ---------------------------------------------------------------------------------------------
# The five data movements of a hand-over.  `arrs`: the 8 state arrays; `idx`: int32[2, idx_cap];
# `counts`: int64[8] ([0] n_left, [1] n_right, [2] received from the left, [3] from the right).
def _select_pack(t, n, z, prefix, cuts, zbox_min, zbox_max, arrs, L, R, idx, counts):
    nattr = len(arrs)
    if z.is_cuda:
        lib, p = _capi.lib(), _capi.ptr
        _capi.check(lib.fb_handover_select_pack(
            n, p(z), p(prefix), cuts[0], cuts[1], cuts[2], cuts[3], zbox_min, zbox_max, nattr,
            _capi.ptr_array(arrs), L.cap, R.cap, idx.shape[1], p(L.send), p(R.send), p(idx[0]), p(idx[1]),
            p(counts), _capi.stream()), 'fb_handover_select_pack')
        return
    counts.zero_()
    for side, link, sel in ((0, L, t.nonzero(z[:n] < zbox_min).reshape(-1)),
                            (1, R, t.nonzero(z[:n] > zbox_max).reshape(-1))):
        cnt = int(sel.numel())
        counts[side] = cnt
        k = min(cnt, idx.shape[1])
        idx[side, :k] = sel[:k].to(t.int32)
        if link.present:
            link.send[0] = float(cnt)
            k = min(cnt, link.cap)
            for a in range(nattr):
                link.send[HEADER + a * link.cap:HEADER + a * link.cap + k] = arrs[a][sel[:k]]


def _recv_counts(t, L, R, counts):
    if counts.is_cuda:
        _capi.check(_capi.lib().fb_handover_recv_counts(_capi.ptr(L.recv), _capi.ptr(R.recv),
                                                        _capi.ptr(counts), _capi.stream()),
                    'fb_handover_recv_counts')
        return
    counts[2] = int(L.recv[0].item()) if L.present else 0
    counts[3] = int(R.recv[0].item()) if R.present else 0


def _pack_rest(t, arrs, idx32, dev):
    """Rows of the particles idx32 (the leavers a fixed-size message could not hold)."""
    nattr, n = len(arrs), int(idx32.numel())
    idx = idx32.to(t.int64)
    if dev.type == 'cuda':
        buf = t.empty((nattr, n), dtype=t.float64, device=dev)
        _capi.check(_capi.lib().fb_handover_pack(n, _capi.ptr(idx), nattr, _capi.ptr_array(arrs),
                                                 _capi.ptr(buf), buf.stride(0), _capi.stream()),
                    'fb_handover_pack')
        return buf
    return t.stack([a.index_select(0, idx) for a in arrs]).contiguous()


def _compact(t, n, n_l, n_r, idx, arrs, st):
    """Remove the n_l + n_r listed particles from the length-n arrays in O(movers): the holes
    below the new length are filled with the survivors above it (any order)."""
    n_leave = n_l + n_r
    if not n_leave:
        return
    if arrs[0].is_cuda:
        lib = _capi.lib()
        need = int(lib.fb_handover_workspace_bytes(n_leave))
        if st['ws'] is None or st['ws'].shape[0] < need:
            st['ws'] = t.empty(2 * need, dtype=t.uint8, device=arrs[0].device)
        _capi.check(lib.fb_handover_compact(n, n_l, _capi.ptr(idx[0]), n_r, _capi.ptr(idx[1]), len(arrs),
                                            _capi.ptr_array(arrs), _capi.ptr(st['ws']), st['ws'].shape[0],
                                            _capi.stream()), 'fb_handover_compact')
        return
    m = n - n_leave
    leave = t.cat((idx[0, :n_l], idx[1, :n_r])).to(t.int64)
    in_tail = leave >= m
    tail_free = t.ones(n - m, dtype=t.bool)
    tail_free[leave[in_tail] - m] = False
    src = m + t.nonzero(tail_free).reshape(-1)
    dst = leave[~in_tail]
    for a in arrs:
        a[dst] = a.index_select(0, src)


def _append(t, out, buf, stride, cnt, first, shift_attr, shift):
    """out[k][first : first + cnt] = row k of buf (rows `stride` doubles apart), + shift on one row."""
    if out[0].is_cuda:
        _capi.check(_capi.lib().fb_handover_append_shift(cnt, first, len(out), _capi.ptr_array(out),
                                                         _capi.ptr(buf), stride, shift_attr, shift,
                                                         _capi.stream()), 'fb_handover_append_shift')
        return
    flat = buf.reshape(-1)
    for k, a in enumerate(out):
        a[first:first + cnt] = flat[k * stride:k * stride + cnt] + (shift if k == shift_attr else 0.)

def exchange_particles_between_ranks(comm, species, fld, time):
    """One hand-over of `species` with the two z neighbours; see the module docstring."""
    finish_particle_handover(comm, species, fld, time, begin_particle_handover(comm, species, fld))


def begin_particle_handover(comm, species, fld):
    """First half of a hand-over: selection + packing, the exchange of the two fixed-size messages and the
    REQUEST of the one host read (the four counts; an event marks it in the stream).  Reads the particle
    arrays, changes none of them.  Simulation.step posts it right behind the particle pass of the iteration
    BEFORE a hand-over iteration: the host then waits for that event only, and prepares compaction and append
    (~300 us of host work, during which the stream used to be empty) while the field kernels of that
    iteration are still queued.  Returns the context `finish_particle_handover` needs."""
    t = _capi.torch()
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
    use_prefix = bool(dev.type == 'cuda' and getattr(species, 'sorted', False)
                      and getattr(species, '_prefix_valid', False)
                      and getattr(species, 'use_bin_sort', False) and n > 0)
    # Round 6: after one-pass iterations the arrays are in HOME order (the order of the last sort, whose
    # prefix sum still describes it: a one-pass iteration permutes nothing), not in cell order - the scan
    # used to fall back to all of z there (150 us per hand-over at C2).  A particle moves less than one
    # cell per step (c dt <= dz), so `since` iterations after the sort every leaver still sits in a home
    # row within `since` + 1 rows of the rows compared for a fresh sort.
    margin = 0
    if (not use_prefix) and dev.type == 'cuda' and n > 0 and getattr(species, '_home_valid', False) \
            and getattr(species, '_prefix_valid', False) and getattr(species, 'use_bin_sort', False) \
            and getattr(species, '_cycle_since_sort', None) is not None \
            and species._cycle_since_sort <= HOME_SCAN_MAX_AGE and comm.dz >= c * abs(getattr(species, 'dt', 0.)) * 0.999:
        use_prefix = True
        margin = int(species._cycle_since_sort) + 1
    cuts = (-1, -1, -1, -1)
    if use_prefix:
        Nz, Nr = fld.Nz, fld.Nr
        shift = getattr(species, 'prefix_sum_shift', 0)      # window moves since the sort
        rows = [min(max(r, 0), Nz) for r in (ng + shift - 1 - margin, ng + shift + 2 + margin,
                                              Nz - ng + shift - 1 - margin, Nz - ng + shift + 2 + margin)]
        cuts = tuple(r * (Nr + 1) - 1 for r in rows)      # -1 = offset 0
    _select_pack(t, n, species.z, species.prefix_sum if use_prefix else None, cuts, zbox_min, zbox_max,
                 arrs, L, R, st['idx'], st['counts'])
    # ONE exchange of the two fixed-size messages; nothing has been read back so far
    comm._handover_caps = (L.cap, R.cap)       # (profiling tools that stand in for the neighbour)
    comm.exchange_domains(L.send, R.send, L.recv, R.recv)
    comm._handover_caps = None
    _recv_counts(t, L, R, st['counts'])
    st['counts_host'].copy_(st['counts'], non_blocking=True)
    ev = None
    if dev.type == 'cuda':
        ev = st.get('event')
        if ev is None:
            ev = st['event'] = t.cuda.Event()
        ev.record(t.cuda.current_stream())
    return {'st': st, 'n': n, 'arrs': arrs, 'idx_cap': idx_cap, 'dev': dev, 'event': ev,
            'z_ptr': species.z.data_ptr()}


def finish_particle_handover(comm, species, fld, time, ctx):
    """Second half: the host read of the counts, then compaction, append and the bookkeeping of the
    species.  `ctx`: what `begin_particle_handover` returned for the same species - whose arrays must not
    have changed since."""
    t = _capi.torch()
    st, n, arrs, idx_cap, dev = ctx['st'], ctx['n'], ctx['arrs'], ctx['idx_cap'], ctx['dev']
    if species.Ntot != n or species.z.data_ptr() != ctx['z_ptr']:
        raise _capi.BackendError('particle hand-over: the particle arrays changed between the two halves '
                                 'of a hand-over (the messages of the first half are already exchanged)')
    L, R = st['left'], st['right']
    nattr = len(_STATE)
    if ctx['event'] is not None:
        ctx['event'].synchronize()                    # the one host read of a hand-over
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
            return _pack_rest(t, arrs, st['idx'][side][link.cap:nsel], dev)
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
    if (comm.moving_win is not None) and (comm.rank == comm.size - 1) \
            and getattr(species, 'continuous_injection', False):
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
    _compact(t, n, n_sl, n_sr, st['idx'], arrs, st)
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
        _append(t, out, buf, stride, cnt, first, iz if shift != 0. else -1, shift)
        first += cnt
    for k, b in zip(_STATE, out):
        setattr(species, k, b)
    species.Ntot = n_new
    for k in _FIELDS:
        # E, B on the particles: inside step() re-sized only - every read follows a gather that
        # stores them.  The reference zeroes them here (:289-417): so does a direct call of
        # comm.exchange_particles, and a neutral species (which never gathers) always
        f = getattr(species, k, None)
        if hasattr(f, 'untyped_storage') and f.device == dev:
            f = _resized(t, f, 0, n_new)
        else:
            f = t.zeros(n_new, dtype=t.float64, device=dev)
        if getattr(species, 'q', 1) == 0 or dev.type != 'cuda' or not getattr(species, '_in_step', False):
            f.zero_()
        setattr(species, k, f)
    # The arrays stay cell-sorted except for the few particles that were moved / appended: the
    # deposition and the gather work on runs of equal cells and do not need more (any order is
    # correct), and the next fused pass re-sorts everything anyway.  Only the per-cell prefix
    # sum is no longer exact.
    few_movers = (n_leave + n_in) * 16 < max(n_new, 1)
    nearly_sorted = bool(getattr(species, 'sorted', False) and few_movers)
    moved = getattr(species, '_moved_since_sort', 0.)
    home = getattr(species, '_home_valid', False) and species.cell_idx is not None
    home_ptr = species.cell_idx.data_ptr() if home else None
    species.on_particle_number_changed()
    species._prefix_valid = False
    if nearly_sorted:
        species.sorted = True
        species._moved_since_sort = moved
    if few_movers:
        # The home cells of the one-pass cycle (Particles.cycle) stay usable as well: a particle
        # that was moved into a hole or appended meets some other particle's (or no) home cell
        # and is simply treated as one that has left it - any content gives the same result.
        # (Decided by the share of movers, not by `sorted`: a one-pass iteration clears that flag
        # - the arrays are in home order, not in cell order - and every hand-over that followed one
        # used to force a sorting iteration.)
        species._home_valid = bool(home and species.cell_idx is not None
                                   and species.cell_idx.data_ptr() == home_ptr)
    # capacities for the next hand-over (same rule, same numbers on both ends of a link)
    L.grow_for(t, dev, max(n_sl, n_rl))
    R.grow_for(t, dev, max(n_sr, n_rr))


