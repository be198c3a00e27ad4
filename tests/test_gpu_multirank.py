"""End-to-end z-domain decomposition on the GPU: the same global simulation run on 1 rank
and on 2 ranks (two processes, gloo transport staged through the host because the test box
has a single GPU; on a multi-GPU node the transport is RCCL, same code path otherwise).
The decomposed run must reproduce the single-domain fields in the physical region and the
same global particle set: the finite-order PSATD stencil (n_order) makes the solver local,
guard cells of width stencil_reach+1 carry the rest (reference docs:
docs/source/overview/parallelisation.rst:126-174)."""
import os
import socket
import tempfile
import numpy as np
import pytest
import torch.multiprocessing as mp
from scipy.constants import c

pytestmark = pytest.mark.gpu

NZ, NR, NM = 256, 32, 2
DZ = 0.2e-6
N_ORDER, N_GUARD = 8, 32
NSTEP = 9            # exchange_period = int((32/2-3)/2) = 6 -> includes one particle hand-over


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_particles(shape):
    import helpers
    sim = helpers.uniform_plasma_sim(NZ, NR, NM, (2, 2, 8), shape, seed=5, u_th=0.1, n_order=N_ORDER)
    s = sim.ptcl[0]
    return np.array([getattr(s, k) for k in helpers.PTCL])


def _run(rank, world, port, shape, outdir, correct):
    import torch
    import torch.distributed as dist
    import helpers
    from fbpic_amd.main import Simulation
    P = _global_particles(shape)      # the GLOBAL particle set (built before the ranks exist)
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
    zmax = NZ * DZ
    sim = Simulation(NZ, zmax, NR, NR * DZ, NM, DZ / c, n_order=N_ORDER, n_guard=N_GUARD,
                     particle_shape=shape)
    zlo, zhi = sim.comm.get_zmin_zmax(local=True, with_damp=False, with_guard=False, rank=rank)
    sel = (P[2] >= zlo) & (P[2] < zhi)
    sp = sim.add_new_species(q=-1.602176634e-19, m=9.1093837139e-31)
    helpers.set_species_state(sp, P[:, sel])
    sim.step(NSTEP, correct_currents=correct)
    ng = sim.comm.n_guard
    sl = slice(ng, sim.fld.Nz - ng) if ng else slice(None)
    out = {}
    for m in range(NM):
        for k in helpers.INTERP:
            out['%s_%d' % (k, m)] = getattr(sim.fld.interp[m], k)[sl]
    for k in helpers.PTCL[:8]:
        out['p_' + k] = getattr(sp, k)
    np.savez(os.path.join(outdir, 'w%d_r%d.npz' % (world, rank)), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _worker(rank, world, port, shape, outdir, correct, q):
    try:
        _run(rank, world, port, shape, outdir, correct)
        q.put((rank, 'ok'))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize('shape,correct,tol,nranks', [('linear', False, 1e-9, 2),
                                                      ('cubic', False, 1e-9, 2),
                                                      ('linear', False, 1e-9, 4)])
def test_two_ranks_reproduce_single_domain(shape, correct, tol, nranks):
    """(nranks = 4: every rank has two distinct neighbours, as on the 4- and 8-GPU runs.)
    Without current correction every operation is local within the stencil reach, so
    the decomposed run must agree with the single domain to rounding.  (With the curl-free
    correction each rank inverts a Laplacian on its own guard-padded box before the J guard
    exchange, reference main.py:530-538: the decomposed scheme then differs from the single
    domain at the percent level BY CONSTRUCTION, in the reference too.  That path is pinned
    rank by rank against the reference itself running decomposed:
    tests/test_gpu_multirank_golden.py, 2e-11.)"""
    import helpers
    outdir = tempfile.mkdtemp()
    ctx = mp.get_context('spawn')
    for world in (1, nranks):
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, shape, outdir, correct, q))
                 for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in range(world)]
        for p in procs:
            p.join(60)
        for rank, msg in res:
            assert msg == 'ok', 'world %d rank %d:\n%s' % (world, rank, msg)
    one = np.load(os.path.join(outdir, 'w1_r0.npz'))
    two = [np.load(os.path.join(outdir, 'w%d_r%d.npz' % (nranks, r))) for r in range(nranks)]
    for m in range(NM):
        for k in helpers.INTERP:
            key = '%s_%d' % (k, m)
            ref = one[key]
            got = np.concatenate([t[key] for t in two], axis=0)
            assert got.shape == ref.shape
            grp = [kk for kk in helpers.INTERP if kk[0] == k[0]]
            scale = max(np.abs(one['%s_%d' % (kk, mm)]).max() for kk in grp for mm in range(NM))
            if scale > 0:
                err = np.abs(got - ref).max() / scale
                assert err < tol, (key, err)
    # global particle set (order differs: compare sorted by (w, x, y, z))
    ref = np.array([one['p_' + k] for k in helpers.PTCL[:8]])
    got = np.concatenate([np.array([t['p_' + k] for k in helpers.PTCL[:8]]) for t in two], axis=1)
    assert got.shape == ref.shape
    assert np.array_equal(np.sort(got[7]), np.sort(ref[7]))     # no particle lost or duplicated
    if correct:
        return      # percent-level field differences reorder near-degenerate particles
    L = NZ * DZ
    ref[2] %= L
    got[2] %= L
    o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
    o2 = np.lexsort((got[2], got[1], got[0], got[7]))
    for j, k in enumerate(helpers.PTCL[:8]):
        d = np.abs(got[j][o2] - ref[j][o1])
        if k == 'z':
            d = np.minimum(d, L - d)
        assert d.max() < max(tol * 1e-2, 1e-9) * max(np.abs(ref[j]).max(), 1e-300), k
