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
from conftest import achieved
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


def _run(rank, world, port, shape, outdir, correct, fuse=True, tag='', n_guard=N_GUARD):
    import torch
    import torch.distributed as dist
    import helpers
    from fbpic_amd.main import Simulation
    P = _global_particles(shape)      # the GLOBAL particle set (built before the ranks exist)
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
    zmax = NZ * DZ
    sim = Simulation(NZ, zmax, NR, NR * DZ, NM, DZ / c, n_order=N_ORDER, n_guard=n_guard,
                     particle_shape=shape)
    zlo, zhi = sim.comm.get_zmin_zmax(local=True, with_damp=False, with_guard=False, rank=rank)
    sel = (P[2] >= zlo) & (P[2] < zhi)
    sp = sim.add_new_species(q=-1.602176634e-19, m=9.1093837139e-31)
    helpers.set_species_state(sp, P[:, sel])
    if not fuse:
        # the forward Hankel transform of J, rho and the curl-free correction as separate launches
        # (default on a decomposed domain: ONE launch, fb_spect_cycle_standard with correct_currents = 2)
        sim.fld.fuse_spectral_cycle = False
    if tag == '_late':
        sim.early_handover = False      # the whole hand-over at its place in the iteration
    sim.step(NSTEP, correct_currents=correct)
    ng = sim.comm.n_guard
    sl = slice(ng, sim.fld.Nz - ng) if ng else slice(None)
    if tag:
        sl = slice(None)            # the whole local grid, guard cells included
        out_launches = sim.fld.spect_cycle_launches
    out = {}
    for m in range(NM):
        for k in helpers.INTERP:
            out['%s_%d' % (k, m)] = getattr(sim.fld.interp[m], k)[sl]
    for k in helpers.PTCL[:8]:
        out['p_' + k] = getattr(sp, k)
    if tag:
        out['launches'] = out_launches
    # hand-overs whose first half was posted behind the particle pass of the iteration before
    # (Simulation.early_handover: on by default; FBPIC_AMD_EARLY_HANDOVER=0 in the environment of a rank)
    out['early_handovers'] = getattr(sim.comm, 'early_handovers', 0)
    np.savez(os.path.join(outdir, 'w%d_r%d%s.npz' % (world, rank, tag)), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _worker(rank, world, port, shape, outdir, correct, q, fuse=True, tag='', n_guard=N_GUARD):
    try:
        _run(rank, world, port, shape, outdir, correct, fuse, tag, n_guard)
        q.put((rank, 'ok'))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize('shape,correct,tol,nranks', [('linear', False, 1e-13, 2),
                                                      ('cubic', False, 1e-13, 2),
                                                      ('linear', False, 1e-13, 4)])
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
    import atexit
    import shutil
    atexit.register(shutil.rmtree, outdir, ignore_errors=True)      # (/tmp is RAM on the test boxes)
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
    # the hand-over inside the call (iteration 6 of 9) took the early form on every rank, none on one rank
    assert int(one['early_handovers']) == 0 and all(int(t['early_handovers']) == 1 for t in two)
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
                achieved(None, err, tol, 'fields')
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
        achieved(None, d.max() / max(np.abs(ref[j]).max(), 1e-300), 1e-14, 'particles')      # measured 4.3e-16


def _run_restart(rank, world, port, shape, outdir, correct):
    """6 steps with a checkpoint after 3 (one file per rank), then a NEW decomposed Simulation
    filled from that checkpoint and stepped 3 more times."""
    import torch.distributed as dist
    import helpers
    from fbpic_amd.main import Simulation
    from fbpic_amd.openpmd_diag import set_periodic_checkpoint, restart_from_checkpoint
    P = _global_particles(shape)
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    zmax = NZ * DZ

    def build(content):
        sim = Simulation(NZ, zmax, NR, NR * DZ, NM, DZ / c, n_order=N_ORDER, n_guard=N_GUARD,
                         particle_shape=shape)
        zlo, zhi = sim.comm.get_zmin_zmax(local=True, with_damp=False, with_guard=False, rank=rank)
        sel = (P[2] >= zlo) & (P[2] < zhi)
        sp = sim.add_new_species(q=-1.602176634e-19, m=9.1093837139e-31)
        Q = P[:, sel].copy()
        if not content:                    # something else entirely: the checkpoint must replace it
            Q = Q[:, ::2]
            Q[3:6] *= -3.
        helpers.set_species_state(sp, Q)
        return sim, sp
    ckdir = os.path.join(outdir, 'ck')
    a, spa = build(True)
    set_periodic_checkpoint(a, 3, checkpoint_dir=ckdir)
    a.step(3, correct_currents=correct)
    a.step(3, correct_currents=correct)
    dist.barrier()
    b, spb = build(False)
    assert restart_from_checkpoint(b, 3, checkpoint_dir=ckdir) == 3 and b.iteration == 3
    b.step(3, correct_currents=correct)
    out = {}
    for tag, sim, sp in (('a', a, spa), ('b', b, spb)):
        for m in range(NM):
            for k in helpers.INTERP:
                out['%s_%s_%d' % (tag, k, m)] = getattr(sim.fld.interp[m], k)
        for k in helpers.PTCL[:8]:
            out['%s_p_%s' % (tag, k)] = getattr(sp, k)
    np.savez(os.path.join(outdir, 'restart_r%d.npz' % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def _restart_worker(rank, world, port, shape, outdir, correct, q):
    try:
        _run_restart(rank, world, port, shape, outdir, correct)
        q.put((rank, 'ok'))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_rank_checkpoint_restart():
    """Restart of a decomposed run (reference: tests/test_example_docs_scripts.py:40-51 with
    test_checkpoint_dir=True on 2 MPI ranks; checkpoint_restart.py:36-37 writes / reads one file
    per rank): run 6 == run 3 + restart on the same 2 ranks + run 3, every rank's whole local
    grids (guard cells included) and particles, with the curl-free correction on."""
    import helpers
    outdir = tempfile.mkdtemp()
    import atexit
    import shutil
    atexit.register(shutil.rmtree, outdir, ignore_errors=True)      # (/tmp is RAM on the test boxes)
    ctx = mp.get_context('spawn')
    port = _free_port()
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_restart_worker, args=(r, world, port, 'linear', outdir, True, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d:\n%s' % (rank, msg)
    assert sorted(os.listdir(os.path.join(outdir, 'ck', 'npz'))) == [
        'checkpoint%08d_rank%d.npz' % (it, r) for it in (3, 6) for r in range(world)]
    for r in range(world):
        d = np.load(os.path.join(outdir, 'restart_r%d.npz' % r))
        for m in range(NM):
            for k in helpers.INTERP:
                grp = [kk for kk in helpers.INTERP if kk[0] == k[0]]
                scale = max(np.abs(d['a_%s_%d' % (kk, mm)]).max() for kk in grp for mm in range(NM))
                if scale > 0:
                    achieved(None, np.abs(d['b_%s_%d' % (k, m)] - d['a_%s_%d' % (k, m)]).max() / scale,
                             1e-12, 'fields')          # measured 1.0e-13
        A = np.array([d['a_p_' + k] for k in helpers.PTCL[:8]])
        B = np.array([d['b_p_' + k] for k in helpers.PTCL[:8]])
        assert A.shape == B.shape
        o1 = np.lexsort((A[2], A[1], A[0], A[7]))
        o2 = np.lexsort((B[2], B[1], B[0], B[7]))
        for j in range(8):
            achieved(None, np.abs(B[j][o2] - A[j][o1]).max() / max(np.abs(A[j]).max(), 1e-300), 1e-14,
                     'particles')


def test_decomposed_correction_deferral_equals_separate_launches():
    """on a decomposed domain the forward Hankel transform of J, rho_next and the
    curl-free correction run as ONE launch in front of the J guard exchange (Simulation._hankel_deferral
    -> 'correct', fb_spect_cycle_standard with correct_currents = 2).  The full-size C4 test cannot see
    a wrong fused launch (Nr = 256 > 128: it does not run there; its bounds are percent-level anyway).
    Here, with Nr = 32: the same two ranks with the deferral and with the separate launches
    (Fields.fuse_spectral_cycle = False), whole local grids incl. guard cells and every particle;
    reference order that must hold: fbpic/main.py:530-557.
    Bound of the fields: 5e-13.  The two runs differ by the summation order of the Hankel products AND by the
    order of their own deposition atomics (not reproducible run to run); the curl-free correction multiplies
    that rounding by ~Nz / 2 pi x n e c / |J|.  Measured over six executions on
    five boxes: 4.3e-14, 9.0e-14, 9.0e-14, 9.9e-14, 9.9e-14, 1.35e-13 - a bound of 1e-13 sat inside the spread.
    A wrong fused launch (a swapped matrix, a missing filter factor) shows at 1e-3 or worse."""
    import helpers
    outdir = tempfile.mkdtemp()
    import atexit
    import shutil
    atexit.register(shutil.rmtree, outdir, ignore_errors=True)
    ctx = mp.get_context('spawn')
    world = 2
    for fuse, tag in ((True, '_fused'), (False, '_sep')):
        port = _free_port()
        q = ctx.Queue()
        # (n_guard = 64: 128 + 2 x 64 = 256 local rows, a length of the LDS z-FFT - the fused launch needs it)
        procs = [ctx.Process(target=_worker, args=(r, world, port, 'linear', outdir, True, q, fuse, tag, 64))
                 for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in range(world)]
        for p in procs:
            p.join(60)
        for rank, msg in res:
            assert msg == 'ok', 'rank %d (%s):\n%s' % (rank, tag, msg)
    for r in range(world):
        a = np.load(os.path.join(outdir, 'w2_r%d_fused.npz' % r))
        b = np.load(os.path.join(outdir, 'w2_r%d_sep.npz' % r))
        assert int(a['launches']) >= NSTEP - 1 and int(b['launches']) == 0      # the launch under test did run
        for m in range(NM):
            for k in helpers.INTERP:
                key = '%s_%d' % (k, m)
                grp = [kk for kk in helpers.INTERP if kk[0] == k[0]]
                scale = max(np.abs(b['%s_%d' % (kk, mm)]).max() for kk in grp for mm in range(NM))
                if scale > 0:
                    achieved(None, np.abs(a[key] - b[key]).max() / scale, 5e-13, 'fields, whole local grid')
        pa = np.array([a['p_' + k] for k in helpers.PTCL[:8]])
        pb = np.array([b['p_' + k] for k in helpers.PTCL[:8]])
        assert pa.shape == pb.shape
        o1 = np.lexsort((pa[2], pa[1], pa[0], pa[7]))
        o2 = np.lexsort((pb[2], pb[1], pb[0], pb[7]))
        for j, k in enumerate(helpers.PTCL[:8]):
            achieved(None, np.abs(pa[j][o1] - pb[j][o2]).max() / max(np.abs(pb[j]).max(), 1e-300), 1e-13, 'particles')


def test_early_handover_equals_handover_in_place():
    """The first half of a particle hand-over (selection, packing, the two messages, the request of the host
    read) posted behind the particle pass of the iteration BEFORE (Simulation.early_handover, the default on
    a decomposed domain without moving window) against the whole hand-over at its place in the iteration
    (reference order, fbpic/main.py:435-446, boundary_communicator.py:750-826): the same particles leave and
    arrive - whole local grids incl. guard cells to the rounding of the deposition
    atomics' order, every particle, the same particle counts."""
    import helpers
    outdir = tempfile.mkdtemp()
    import atexit
    import shutil
    atexit.register(shutil.rmtree, outdir, ignore_errors=True)
    ctx = mp.get_context('spawn')
    world = 2
    for tag in ('_early', '_late'):
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, 'linear', outdir, False, q, True, tag))
                 for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in range(world)]
        for p in procs:
            p.join(60)
        for rank, msg in res:
            assert msg == 'ok', 'rank %d (%s):\n%s' % (rank, tag, msg)
    for r in range(world):
        a = np.load(os.path.join(outdir, 'w2_r%d_early.npz' % r))
        b = np.load(os.path.join(outdir, 'w2_r%d_late.npz' % r))
        assert int(a['early_handovers']) == 1 and int(b['early_handovers']) == 0
        for m in range(NM):
            for k in helpers.INTERP:
                key = '%s_%d' % (k, m)
                grp = [kk for kk in helpers.INTERP if kk[0] == k[0]]
                scale = max(np.abs(b['%s_%d' % (kk, mm)]).max() for kk in grp for mm in range(NM))
                if scale > 0:
                    achieved(None, np.abs(a[key] - b[key]).max() / scale, 1e-13, 'fields, whole local grid')
        pa = np.array([a['p_' + k] for k in helpers.PTCL[:8]])
        pb = np.array([b['p_' + k] for k in helpers.PTCL[:8]])
        assert pa.shape == pb.shape                      # the same number of particles on the rank
        # (the ORDER of the leavers in a message, and with it of the arrivals and of the survivors that fill
        # the holes, is that of the selection's atomics - it differs between two executions of either form)
        o1 = np.lexsort((pa[2], pa[1], pa[0], pa[7]))
        o2 = np.lexsort((pb[2], pb[1], pb[0], pb[7]))
        for j in range(8):
            achieved(None, np.abs(pa[j][o1] - pb[j][o2]).max() / max(np.abs(pb[j]).max(), 1e-300), 1e-13,
                     'particles')
