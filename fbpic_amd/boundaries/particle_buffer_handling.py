"""Particle hand-over between neighbouring z-slabs (one rank per GPU).

Semantics of fbpic/boundaries/particle_buffer_handling.py:17-172 (remove_outside_particles)
and :289-417 (add_buffers_to_particles) + boundary_communicator.py:750-826: particles whose
z left the local *physical* range [zmin + ng dz, zmax - ng dz] are removed and sent to the
left / right neighbour (dropped at an open end); received particles are appended as
(from-left | stayed | from-right); particles that wrapped around the periodic box are
shifted by +-L.  The selection is expressed with device-agnostic tensor operations
(boolean masks on the SoA tensors) so that the same code runs on RCCL/GPU tensors and on
gloo/CPU tensors in the tests; it runs once every `exchange_period` (~14) steps.
"""
from .. import _capi

_STATE = ('x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w')     # reference buffer order
_FIELDS = ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz')


_PRIMED = set()


def _prime_device_ops(t, dev):
    """Run the tensor operations of the hand-over once on a few dummy values.  The device code
    of an operation is loaded the first time it runs (tens of ms); without this, that cost
    lands on the first step in which a particle really crosses a boundary - typically
    `exchange_period` steps into a run - instead of on the first (warm-up) step."""
    if dev in _PRIMED or dev.type != 'cuda':
        return
    _PRIMED.add(dev)
    a = t.arange(16, dtype=t.float64, device=dev)
    b = t.stack([a[2:5], a[3:6]]).contiguous()
    c = t.cat((b[0], a[1:9], b[1])).contiguous()
    b[1] += 1.
    m = a > 3.
    t.cat((a[m], c[~m[:14]])).sum().item()
    t.zeros(4, dtype=t.float64, device=dev)
    t.tensor([3], dtype=t.int64, device=dev).item()


def exchange_particles_between_ranks(comm, species, fld, time):
    t = _capi.torch()
    _prime_device_ops(t, species.z.device)
    g0 = fld.interp[0]
    ng = comm.n_guard
    zbox_min = g0.zmin + ng * g0.dz
    zbox_max = g0.zmax - ng * g0.dz
    z = species.z
    dev = z.device
    cut = (z.is_cuda and comm.left_proc is not None and comm.right_proc is not None
           and getattr(species, 'use_bin_sort', False) and species.Ntot > 0)
    if cut:
        # Device path between two neighbours, as the reference's GPU path
        # (particle_buffer_handling.py:177-236): the particles are cell-sorted, so the ones in
        # the guard cells are the two ends of the arrays - cut at two prefix-sum offsets (one
        # host read) instead of building three boolean selections of every attribute.  (Cell
        # based: a particle changes owner half a cell later than with the z comparison of the
        # reference's CPU path, which the ranks next to an open end keep using.)
        if not species.sorted:
            species.sort_particles(fld=fld)
            species.sorted = True
        Nz, Nr = fld.Nz, fld.Nr
        iz_min = max(ng + species.prefix_sum_shift, 0)
        iz_max = min(Nz - ng + species.prefix_sum_shift + 1, Nz)
        ps = species.prefix_sum
        ends = t.stack((ps[max(iz_min * (Nr + 1) - 1, 0)], ps[iz_max * (Nr + 1) - 1])).tolist()
        i_min = int(ends[0]) if iz_min * (Nr + 1) - 1 >= 0 else 0
        i_max = int(ends[1])
        sel_l, sel_r, stay = slice(0, i_min), slice(i_max, species.Ntot), slice(i_min, i_max)
        nothing_leaves = (i_min == 0 and i_max == species.Ntot)
    else:
        sel_l = z < zbox_min
        sel_r = z > zbox_max
        stay = ~(sel_l | sel_r)
        nothing_leaves = False
    arrs = [getattr(species, k) for k in _STATE]

    def pack(sel, proc):
        if proc is None:
            return t.empty((len(_STATE), 0), dtype=t.float64, device=dev)
        return t.stack([a[sel] for a in arrs]).contiguous()
    send_l = pack(sel_l, comm.left_proc)
    send_r = pack(sel_r, comm.right_proc)
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
    if nothing_leaves and n_rl == 0 and n_rr == 0:
        return                       # nobody crossed a boundary: arrays (and their sort) stay
    for i, k in enumerate(_STATE):
        setattr(species, k, t.cat((recv_l[i], arrs[i][stay], recv_r[i])).contiguous())
    species.Ntot = int(species.x.shape[0])
    for k in _FIELDS:
        setattr(species, k, t.zeros(species.Ntot, dtype=t.float64, device=dev))
    species.sorted = False
    species.on_particle_number_changed()
