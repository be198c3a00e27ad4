"""N > 1 path on CPU: two ranks over the gloo backend (world_size 2).  Exercises the
z-slab decomposition arithmetic, the guard-cell exchange ('replace' for E/B, 'add' for
J/rho), and the particle hand-over incl. the periodic wrap -- the same code that runs on
RCCL with one rank per MI355X (torch tensors, backend-agnostic)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from scipy.constants import c

NZ, NR, NM, NG = 48, 6, 2, 4
DZ = 0.5e-6


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Grid:
    pass


def _make_comm(boundary):
    from fbpic_amd.boundaries.boundary_communicator import BoundaryCommunicator
    return BoundaryCommunicator(NZ, 0., NZ * DZ, NR, NR * DZ, NM, DZ / c, None, False,
                                {'z': boundary, 'r': 'reflective'}, 8, NG, {'z': 8, 'r': 4},
                                1., None, 1, True)


def _global_field(seed, nz):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(nz, NR)) + 1j * rng.normal(size=(nz, NR))


def _worker(rank, world, port, boundary, q):
    try:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
        comm = _make_comm(boundary)
        assert (comm.rank, comm.size) == (rank, world)
        Nz_l, iz0 = comm.get_Nz_and_iz(local=True, with_damp=True, with_guard=True, rank=rank)
        zmin_l, zmax_l, Nz_chk = comm.divide_into_domain()
        assert Nz_chk == Nz_l and abs((zmax_l - zmin_l) / DZ - Nz_l) < 1e-9
        # ---------------- 'replace': guards must receive the neighbour's valid cells
        Nz_phys = NZ // world
        names = ('Er', 'Et', 'Ez')
        interp = []
        for m in range(NM):
            g = _Grid()
            for i, k in enumerate(names):
                G = _global_field(10 * m + i, NZ)
                idx = (np.arange(Nz_l) + iz0) % NZ if boundary == 'periodic' else None
                if boundary == 'periodic':
                    loc = G[idx].copy()
                    loc[:NG] = 99.; loc[-NG:] = 99.          # garbage in the guard cells
                    setattr(g, k, torch.from_numpy(loc))
                    setattr(g, k + '_expect', G[idx])
            interp.append(g)
        if boundary == 'periodic':
            comm.exchange_fields(interp, 'E', 'replace')
            for g in interp:
                for k in names:
                    assert np.array_equal(getattr(g, k).numpy(), getattr(g, k + '_expect')), k
        # ---------------- 'EB' = E and B in one message per neighbour (what step() uses)
        names6 = ('Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz')
        if boundary == 'periodic':
            interp = []
            for m in range(NM):
                g = _Grid()
                for i, k in enumerate(names6):
                    G = _global_field(300 + 10 * m + i, NZ)
                    idx = (np.arange(Nz_l) + iz0) % NZ
                    loc = G[idx].copy()
                    loc[:NG] = -7.; loc[-NG:] = -7.
                    setattr(g, k, torch.from_numpy(loc))
                    setattr(g, k + '_expect', G[idx])
                interp.append(g)
            comm.exchange_fields(interp, 'EB', 'replace')
            for g in interp:
                for k in names6:
                    assert np.array_equal(getattr(g, k).numpy(), getattr(g, k + '_expect')), k
        # ---------------- 'add': overlapping [0,2ng) / [Nz-2ng,Nz) regions are summed
        def local_J(r, m, i, nz):
            return _global_field(1000 + 100 * r + 10 * m + i, nz)
        Nz_of = [comm.get_Nz_and_iz(True, True, True, rank=r)[0] for r in range(world)]
        interp = []
        for m in range(NM):
            g = _Grid()
            for i, k in enumerate(('Jr', 'Jt', 'Jz')):
                setattr(g, k, torch.from_numpy(local_J(rank, m, i, Nz_l).copy()))
            interp.append(g)
        comm.exchange_fields(interp, 'J', 'add')
        for m in range(NM):
            for i, k in enumerate(('Jr', 'Jt', 'Jz')):
                exp = local_J(rank, m, i, Nz_l).copy()
                if comm.left_proc is not None:
                    exp[:2 * NG] += local_J(comm.left_proc, m, i, Nz_of[comm.left_proc])[-2 * NG:]
                if comm.right_proc is not None:
                    exp[-2 * NG:] += local_J(comm.right_proc, m, i, Nz_of[comm.right_proc])[:2 * NG]
                assert np.allclose(getattr(interp[m], k).numpy(), exp, rtol=0, atol=1e-14), k
        # ---------------- gather of the physical cells on rank 0 (diagnostics)
        for with_damp in (False, True):
            Nz_g, iz_g = comm.get_Nz_and_iz(local=False, with_damp=with_damp, with_guard=False)
            G = _global_field(4242, Nz_g)
            idx = np.arange(Nz_l) + iz0 - iz_g          # my rows in the global array
            ok = (idx >= 0) & (idx < Nz_g)
            loc = np.full((Nz_l, NR), 1e30 + 0j)
            loc[ok] = G[idx[ok]]                        # guard rows that wrap: garbage
            got = comm.gather_grid_array(torch.from_numpy(loc) if with_damp else loc,
                                         with_damp=with_damp)
            if rank == 0:
                assert got.shape == (Nz_g, NR) and np.array_equal(got, G), with_damp
            else:
                assert got is None
        # ---------------- particle hand-over (fixed-size messages with the count in the header;
        # a capacity of 16 particles makes every link overflow: the remainder message and the
        # growth of the link capacities are exercised too)
        from fbpic_amd.boundaries import particle_buffer_handling as pbh
        from fbpic_amd.boundaries.particle_buffer_handling import exchange_particles_between_ranks
        pbh._CAP0 = 16

        class _Sp:
            def on_particle_number_changed(self):
                self.changed = True
        sp = _Sp()
        rng = np.random.default_rng(77 + rank)
        n = 500
        zlo, zhi = zmin_l + NG * DZ, zmax_l - NG * DZ          # local physical range
        z = rng.uniform(zlo - 1.5 * DZ, zhi + 1.5 * DZ, n)
        ids = rank * 10000 + np.arange(n, dtype=np.float64)
        for k in ('x', 'y', 'ux', 'uy', 'uz', 'inv_gamma'):
            setattr(sp, k, torch.from_numpy(rng.normal(size=n)))
        sp.z = torch.from_numpy(z.copy())
        sp.w = torch.from_numpy(ids.copy())
        sp.Ntot = n

        class _F:
            pass
        fld = _F()
        g0 = _Grid()
        g0.zmin, g0.zmax, g0.dz = zmin_l, zmax_l, DZ
        fld.interp = [g0]
        exchange_particles_between_ranks(comm, sp, fld, 0.)
        links = [sp._handover[k] for k in ('left', 'right')]
        assert all((not l.present) or l.cap >= 32 for l in links), [l.cap for l in links]
        znew = sp.z.numpy()
        assert sp.Ntot == znew.size == sp.Ex.shape[0] and sp.changed
        assert np.all(znew >= zlo - 1e-12) and np.all(znew <= zhi + 1e-12)
        # global conservation: every particle that had a neighbour to go to still exists once
        mine = torch.zeros(world * 10000, dtype=torch.float64)
        mine[sp.w.numpy().astype(int)] = 1.
        dist.all_reduce(mine)
        expected_lost = 0
        if boundary == 'open':
            # particles leaving through an open end are dropped
            tot = torch.tensor([float(((z < zlo) & (comm.left_proc is None)).sum()
                                      + ((z > zhi) & (comm.right_proc is None)).sum())])
            dist.all_reduce(tot)
            expected_lost = int(tot.item())
        assert int(mine.sum().item()) == world * n - expected_lost
        assert mine.max().item() == 1.
        # positions are unchanged modulo the box length
        L = NZ * DZ
        own = (sp.w.numpy().astype(int) // 10000) == rank
        z0 = z[(sp.w.numpy()[own].astype(int)) % 10000]
        assert np.allclose((znew[own] - z0 + L / 2) % L - L / 2, 0., atol=1e-18)
        q.put((rank, 'ok'))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize('boundary,world', [('periodic', 2), ('open', 2), ('periodic', 3),
                                            ('open', 3)])
def test_two_rank_exchange_gloo(boundary, world):
    """world = 2: both neighbours of a periodic ring are the same peer (message order matters);
    world = 3: distinct neighbours, and for an open chain one interior rank."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, boundary, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d:\n%s' % (rank, msg)


def test_decomposition_arithmetic_single_process():
    """get_Nz_and_iz / get_zmin_zmax for every rank of a 4-rank open-boundary split,
    evaluated without a process group (rank passed explicitly)."""
    from fbpic_amd.boundaries.boundary_communicator import BoundaryCommunicator
    comm = BoundaryCommunicator(1026, -1e-6, -1e-6 + 1026 * DZ, 8, 8 * DZ, 2, DZ / c, None, False,
                                {'z': 'open', 'r': 'reflective'}, 32, None, {'z': 64, 'r': 32},
                                1., None, None, False)
    assert comm.n_guard == 63 and comm.n_inject == 31       # ng = reach + 1
    assert comm.exchange_period == int((63 / 2 - 3) / 2.)
    comm.size = 4       # decomposition arithmetic only
    per = 1026 // 4
    tot = 0
    for r in range(4):
        Nz, iz = comm.get_Nz_and_iz(local=True, with_damp=False, with_guard=False, rank=r)
        assert iz == r * per and Nz == per + (2 if r == 3 else 0)
        tot += Nz
        Nzg, izg = comm.get_Nz_and_iz(local=True, with_damp=True, with_guard=True, rank=r)
        extra = (64 + 31 if r in (0, 3) else 0)
        assert Nzg == Nz + 2 * 63 + extra and izg == iz - 63 - (64 + 31 if r == 0 else 0)
    assert tot == 1026
    d = comm.generate_damp_array(63, 64, 31)
    assert d.shape == (158,) and np.all(d[:94] == 0.) and np.all(d[126:] == 1.)
    assert np.all(np.diff(d[94:126]) > 0)


def test_rccl_handshake_logic(monkeypatch):
    """Start-up handshake of the in-library transport (boundary_communicator._rccl_handshake),
    with the exchange replaced by a stand-in peer: correct routing keeps the transport, a peer
    that swaps its sides or a rank without communicator makes every rank fall back."""
    from fbpic_amd import _capi
    from fbpic_amd.boundaries import boundary_communicator as bc

    class FakeDist:
        def __init__(self, others_ok=True):
            self.others_ok = others_ok

        def all_gather_object(self, out, obj):
            out[:] = [obj] + [self.others_ok] * (len(out) - 1)

    class Comm:            # the attributes the handshake reads
        rank, size, left_proc, right_proc = 1, 4, 0, 2
        transport = 'rccl'
        _rccl_handshake = bc.BoundaryCommunicator._rccl_handshake

    monkeypatch.setattr(_capi, 'require_device', lambda: torch.device('cpu'))

    def peer(swap):
        def exchange(self, send_l, send_r, recv_l, recv_r):
            # the left neighbour's send-to-right arrives from the left, and vice versa
            recv_l.copy_(torch.tensor([float(self.left_proc), 0. if swap else 1.], dtype=torch.float64))
            recv_r.copy_(torch.tensor([float(self.right_proc), 1. if swap else 0.], dtype=torch.float64))
        return exchange

    for swap, others_ok, init_error, expect in ((False, True, None, 'rccl'), (True, True, None, 'torch'),
                                                (False, False, None, 'torch'), (False, True, 'no librccl', 'torch')):
        comm = Comm()
        comm._exchange_rccl = peer(swap).__get__(comm)
        monkeypatch.setattr(bc, '_dist', lambda ok=others_ok: FakeDist(ok))
        comm._rccl_handshake(init_error=init_error)
        assert comm.transport == expect, (swap, others_ok, init_error)


def _direct_worker(rank, world, port, boundary, q):
    """exchange_fields on a z-major slab whose field group is the WHOLE slab (the scratch slab
    that holds the z-real E, B of a decomposed step): the guard rows are sent from and received
    into the rows themselves - contiguous blocks, row padding included - without message
    buffers (boundary_communicator.exchange_fields, 'replace').  Torch tensors on the CPU stand
    in for the device slab; the path contains no kernel."""
    try:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
        comm = _make_comm(boundary)
        Nz_l, iz0 = comm.get_Nz_and_iz(local=True, with_damp=True, with_guard=True, rank=rank)
        nf, pad = 6 * NM, 8
        rs = nf * NR + pad
        base = torch.zeros(Nz_l * rs, dtype=torch.complex128)
        slab = base.as_strided((Nz_l, nf, NR), (rs, NR, 1))
        # global reference: field f at global cell row g holds (g + 1) + 100 f (+ i)
        gz = (np.arange(Nz_l) + iz0)
        if boundary == 'periodic':
            gz = gz % NZ
        vals = (gz[:, None, None] + 1.) + 100. * np.arange(nf)[None, :, None] + 1j * np.arange(NR)[None, None, :]
        slab.copy_(torch.from_numpy(vals))
        truth = slab.clone()
        # spoil my guard rows: the exchange must restore them from the neighbours
        ng = comm.n_guard
        if comm.left_proc is not None:
            slab[:ng] = -7.
        if comm.right_proc is not None:
            slab[Nz_l - ng:] = -7.

        class Owner:
            Nm = NM
            data_is_on_gpu = True
            d_interp = None

            @staticmethod
            def _group(fieldtype):
                return 0, 0, nf, True
        grids = []
        for m in range(NM):
            g = _Grid()
            g._owner = Owner
            g.Er = slab[:, 6 * m, :]
            grids.append(g)
        comm.exchange_fields(grids, 'EB', 'replace', slab=slab)
        assert torch.equal(slab, truth), 'guard rows not restored on rank %d' % rank
        # (and the padding between the rows was left alone: it is never read)
        q.put((rank, 'ok'))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize('boundary,world', [('periodic', 2), ('periodic', 3), ('open', 3)])
def test_direct_row_exchange_gloo(boundary, world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_direct_worker, args=(r, world, port, boundary, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d:\n%s' % (rank, msg)


def _store_worker(rank, world, port, q):
    try:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
        from fbpic_amd.boundaries.boundary_communicator import _host_side_all_gather
        out = _host_side_all_gather(dist, ('host%d' % (rank % 2), rank))
        assert out == [('host%d' % (r % 2), r) for r in range(world)], out
        q.put((rank, 'ok'))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_host_side_all_gather_gloo():
    """The (host, gpu) pairs of the one-rank-per-GPU check travel over the rendezvous store, not
    over a device communicator (which could itself hang when two ranks share a GPU)."""
    world, port = 3, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_store_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d:\n%s' % (rank, msg)


def test_handover_link_capacity_rule_is_symmetric():
    """Both ends of a neighbour link see the same two counts after a hand-over and grow the
    capacity of the fixed-size message by the same rule (particle_buffer_handling._Link)."""
    from fbpic_amd.boundaries.particle_buffer_handling import _Link, _CAP0
    cpu = torch.device('cpu')
    a, b = _Link(torch, cpu, True), _Link(torch, cpu, True)      # my right link / the neighbour's left link
    assert a.cap == b.cap == _CAP0 and a.send.numel() == 8 + 8 * _CAP0
    for n_out, n_in in ((10, 0), (_CAP0, 3), (_CAP0 + 1, 7), (5, 300000), (1, 1)):
        a.grow_for(torch, cpu, max(n_out, n_in))          # I sent n_out, received n_in
        b.grow_for(torch, cpu, max(n_in, n_out))          # the neighbour sent n_in, received n_out
        assert a.cap == b.cap and a.cap >= max(n_out, n_in) and a.recv.numel() == 8 + 8 * a.cap
    assert a.cap == 1 << 20                                # 2 x 300000 rounded up to a power of two


def test_rows_untouched_by_the_EB_exchange():
    from fbpic_amd.boundaries.boundary_communicator import BoundaryCommunicator
    comm = BoundaryCommunicator(1024, 0., 1024 * DZ, 8, 8 * DZ, 2, DZ / c, None, False,
                                {'z': 'open', 'r': 'reflective'}, 32, 64, {'z': 64, 'r': 32},
                                1., None, None, False)
    comm.size = 4
    Nz = 256 + 128
    comm.left_proc, comm.right_proc = None, 1            # first rank: damp + inject cells on the left
    assert comm.rows_untouched_by_EB_exchange(Nz + 96) == (64 + 64 + 32, Nz + 96 - 64)
    comm.left_proc, comm.right_proc = 0, 2               # interior rank
    assert comm.rows_untouched_by_EB_exchange(Nz) == (64, Nz - 64)
    comm.left_proc, comm.right_proc = 2, None            # last rank
    assert comm.rows_untouched_by_EB_exchange(Nz + 96) == (64, Nz + 96 - 160)
