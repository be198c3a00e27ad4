"""BASELINE config C4 at its own size: the laser-wakefield run of docs/source/example_input/
lwfa_script.py on a 4096 x 256 grid (Nm = 2, 16 ppc, open z, moving window, continuous
injection, a0 = 4 laser), n_order = 32, cut into 8 z-slabs (reference:
tests/test_example_docs_scripts.py:40-51 runs that script on several MPI ranks).

The test box has one GPU, so the 8 ranks are 8 processes sharing it (gloo transport staged
through the host; on an 8-GPU node the same code path moves device buffers with RCCL, see
test_decomposed_on_real_gpus).  Without current correction every operation is local within the
stencil reach, so the decomposed run must reproduce the single-domain run (= C3 with the same
finite-order solver) in the physical region to rounding, with exactly the same global particle
set: 16 steps cover one particle hand-over between all neighbours, 16 moves of the window and
one injection of new plasma on the last rank.  (With the curl-free correction the decomposed
SCHEME differs from the single domain by construction, in the reference too; that path is
pinned rank by rank against the reference running decomposed - tests/test_gpu_multirank_golden.py,
`mr_lwfa_lin_2r` being this configuration in miniature.)"""
import json
import os
import socket
import subprocess
import sys
import tempfile
import numpy as np
import pytest
import torch.multiprocessing as mp
from scipy.constants import c
from conftest import achieved, ROOT

pytestmark = pytest.mark.gpu

NZ, NR, NM = 4096, 256, 2
ZMIN, ZMAX, RMAX = -10.e-6, 30.e-6, 20.e-6
NSTEP = 16
FIELDS = ['Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz', 'Jr', 'Jt', 'Jz', 'rho']
PTCL = ['x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    from fbpic_amd.main import Simulation
    ramp_start, ramp_length = 5.e-6, 10.e-6

    def dens_func(z, r):
        n = np.ones_like(z)
        n = np.where(z < ramp_start + ramp_length, (z - ramp_start) / ramp_length, n)
        return np.where(z < ramp_start, 0., n)
    np.random.seed(0)
    dt = (ZMAX - ZMIN) / NZ / c
    return Simulation(NZ, ZMAX, NR, RMAX, NM, dt, zmin=ZMIN, p_zmin=ramp_start, p_zmax=500.e-6,
                      p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24,
                      dens_func=dens_func, n_order=32, n_guard=64, particle_shape='linear',
                      boundaries={'z': 'open', 'r': 'reflective'})


def _run(rank, world, port, outdir, correct=False):
    import time
    t_start = time.time()
    tlog = os.path.join(ROOT, 'gpurun_out', 'c4_timing')
    os.makedirs(tlog, exist_ok=True)

    def mark(what):
        with open(os.path.join(tlog, 'w%d_r%d.log' % (world, rank)), 'a') as f:
            f.write('%8.1f s  %s\n' % (time.time() - t_start, what))
    import torch
    torch.set_num_threads(2)
    import torch.distributed as dist
    import helpers
    mark('imports done')
    from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
    # the GLOBAL initial plasma (lattice + one np.random stream), built once by the parent
    P = np.load(os.path.join(outdir, 'global_particles.npy'), mmap_mode='r')
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
    sim = _build()
    mark('Simulation built (local grid %d rows)' % sim.fld.Nz)
    zlo, zhi = sim.comm.get_zmin_zmax(local=True, with_damp=False, with_guard=False, rank=rank)
    if rank == world - 1:
        zhi = np.inf
    z = np.asarray(P[2])
    sel = (z >= zlo) & (z < zhi)
    helpers.set_species_state(sim.ptcl[0], np.asarray(P[:, sel]))
    add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
    sim.set_moving_window(v=c)
    mark('particles selected, laser added')
    np.random.seed(12345)              # the angles of the injected plasma: same draws in both runs
    sim.step(NSTEP, correct_currents=correct)
    mark('%d steps done' % NSTEP)
    Nz_phys, iz0 = sim.comm.get_Nz_and_iz(local=True, with_damp=False, with_guard=False, rank=rank)
    _, iz_arr = sim.comm.get_Nz_and_iz(local=True, with_damp=True, with_guard=True, rank=rank)
    sl = slice(iz0 - iz_arr, iz0 - iz_arr + Nz_phys)
    out = {'zmin': sim.fld.interp[0].zmin + (iz0 - iz_arr) * sim.fld.interp[0].dz, 'Nz_local': sim.fld.Nz,
           'n_guard': sim.comm.n_guard, 'exchange_period': sim.comm.exchange_period}
    for m in range(NM):
        for k in FIELDS:
            out['%s_%d' % (k, m)] = getattr(sim.fld.interp[m], k)[sl]
    for k in PTCL:
        out['p_' + k] = getattr(sim.ptcl[0], k)
    np.savez(os.path.join(outdir, 'w%d_r%d.npz' % (world, rank)), **out)
    mark('results written')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _worker(rank, world, port, outdir, q, correct=False):
    try:
        # a rank that is still running after 75 s writes where it is (a hang of one rank
        # stalls all the others in their next exchange)
        import faulthandler
        tlog = os.path.join(ROOT, 'gpurun_out', 'c4_timing')
        os.makedirs(tlog, exist_ok=True)
        # (appended, not truncated: the retry that follows a lost rank used to overwrite what the
        # lost rank had dumped - round 5 caught a SIGABRT of one rank and found its dump empty)
        stack = open(os.path.join(tlog, 'w%d_r%d_stack.log' % (world, rank)), 'a')
        stack.write('---- pid %d, correct=%s\n' % (os.getpid(), correct))
        stack.flush()
        faulthandler.enable(file=stack)          # and on a fatal signal (SIGSEGV, SIGABRT, SIGBUS)
        faulthandler.dump_traceback_later(75, repeat=False, file=stack)
        # what the runtime libraries print before they abort (a GPU memory access fault of the HSA
        # runtime, an uncaught C++ exception of the transport) goes to the rank's own file
        err = open(os.path.join(tlog, 'w%d_r%d_stderr.log' % (world, rank)), 'a')
        err.write('---- pid %d, correct=%s\n' % (os.getpid(), correct))
        err.flush()
        os.dup2(err.fileno(), 2)
        _run(rank, world, port, outdir, correct)
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, 'ok'))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def _cgroup(name):
    for base in ('/sys/fs/cgroup', '/sys/fs/cgroup/memory'):
        try:
            with open(os.path.join(base, name)) as fh:
                return fh.read().strip().replace('\n', '; ')
        except OSError:
            continue
    return 'n/a'


def _loss_report(world, before):
    """What can still be learnt after rank processes vanished without a word: the cgroup's OOM
    counters (a process killed by the memory controller leaves no Python trace), its peak,
    the kernel log if readable, who holds GPU memory."""
    lines = ['memory.events before: ' + before, 'memory.events after:  ' + _cgroup('memory.events'),
             'memory.max: ' + _cgroup('memory.max'), 'memory.peak: ' + _cgroup('memory.peak'),
             'memory.current: ' + _cgroup('memory.current'), 'pids.max: ' + _cgroup('pids.max'),
             'pids.peak: ' + _cgroup('pids.peak')]
    for cmd in (['dmesg'], ['rocm-smi', '--showpids', '--showmemuse']):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            lines.append('$ ' + ' '.join(cmd))
            lines += (r.stdout or r.stderr).splitlines()[-25:]
        except Exception as exc:        # pragma: no cover
            lines.append('$ %s: %r' % (' '.join(cmd), exc))
    path = os.path.join(ROOT, 'gpurun_out', 'c4_timing', 'loss_report_w%d.txt' % world)
    with open(path, 'a') as f:
        f.write('\n'.join(lines) + '\n' + '-' * 60 + '\n')
    return path


def _launch(world, outdir, correct=False):
    # The ranks are processes of ONE host here: keep their BLAS / OpenMP pools small.  The GPU
    # boxes show 256 logical CPUs but grant a quota of 16 cores; eight processes with one
    # 256-thread pool each (scipy pinv of the Hankel matrices, NumPy, torch) spend their time
    # being throttled (measured on the test box: 13 minutes for this test, against 28 s with two
    # threads each).
    for var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[var] = '2'
    ctx = mp.get_context('spawn')
    port = _free_port()
    q = ctx.Queue()
    events_before = _cgroup('memory.events')
    procs = [ctx.Process(target=_worker, args=(r, world, port, outdir, q, correct)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=100))
    except queue.Empty:
        pass
    for p in procs:
        p.join(5 if len(res) < world else 120)
        if p.is_alive():
            p.kill()
    lost = len(res) < world
    if lost:
        exits = ['r%d: exitcode %r' % (r, p.exitcode) for r, p in enumerate(procs)]
        rep = _loss_report(world, events_before)
        with open(rep, 'a') as f:
            f.write('exit codes (negative = killed by that signal): ' + ', '.join(exits) + '\n')
    transport = ('Connection closed by peer', 'Connection reset by peer', 'Socket closed',
                 'Connection refused', 'Broken pipe')
    for rank, msg in res:
        if msg != 'ok' and (lost or any(k in msg for k in transport)):
            # a peer died: this rank only reports the broken connection (kept for the record)
            with open(os.path.join(ROOT, 'gpurun_out', 'c4_timing', 'lost_w%d_r%d.txt' % (world, rank)), 'a') as f:
                f.write(msg + '\n')
            lost = True
            continue
        assert msg == 'ok', 'world %d rank %d:\n%s' % (world, rank, msg)
    return not lost


def _launch_once(world, outdir, correct=False):
    """No retry any more.  Rounds 3-5 lost rank processes in ~1 run of 15-20 of this test (one rank
    ending with SIGABRT, the others with a closed connection) and repeated the launch.  Round 5 found
    the cause with the per-rank stderr files kept above - the message header of the particle hand-over
    could be written before all waves had counted their leavers (csrc/handover.hip,
    k_handover_select_pack), the receiver then posted a remainder message 64 particles short and gloo
    aborted that rank - fixed it, and ran 30 executions without a loss (profiles/r05_c4_stress.txt).
    A lost rank now fails the test, with what the ranks said under gpurun_out/c4_timing/."""
    assert _launch(world, outdir, correct), (
        '%d rank processes disappeared without reporting: see gpurun_out/c4_timing/loss_report_w%d.txt, '
        'w%d_r*_stderr.log, w%d_r*_stack.log' % (world, world, world, world))


@pytest.mark.parametrize('correct', [False, True])
def test_c4_lwfa_4096x256_on_8_slabs_reproduces_the_single_domain(correct):
    """correct = False: every operation is local within the stencil reach - the 8 slabs reproduce the
    single domain to rounding.  correct = True: the run as docs/source/example_input/lwfa_script.py
    makes it (`sim.step` with its default curl-free correction, reference main.py:530-538): every
    rank inverts the Laplacian of ITS slab before the J guard exchange, so the decomposed scheme
    differs from the single domain by construction (in the reference too; rank-by-rank parity with
    the reference is pinned in miniature by tests/test_gpu_multirank_golden.py, mr_lwfa_lin_8r) -
    here the full-size run must stay within the scheme difference of the single domain, with
    exactly the same particle set."""
    import helpers
    import atexit
    import shutil
    # (/tmp is RAM on the test boxes: the 3-6 GB this test writes there count against the
    # container's memory until they are removed - memory.peak grew by that much per run)
    outdir = tempfile.mkdtemp()
    atexit.register(shutil.rmtree, outdir, ignore_errors=True)
    world = 8
    glob = _build()
    np.save(os.path.join(outdir, 'global_particles.npy'),
            np.array([getattr(glob.ptcl[0], k) for k in helpers.PTCL]))
    del glob
    _launch_once(1, outdir, correct=correct)
    _launch_once(world, outdir, correct=correct)
    one = np.load(os.path.join(outdir, 'w1_r0.npz'))
    parts = [np.load(os.path.join(outdir, 'w%d_r%d.npz' % (world, r))) for r in range(world)]
    # local grids: 512 physical cells each + 2 x n_guard cells (+ 64 damp and n_guard / 2 inject
    # cells at the two ends); the 16 steps include a particle hand-over between all neighbours
    # (n_guard = 64 >= stencil reach of n_order 32 + 1 = 63: FFT-friendly lengths 640 / 736 / 4416)
    ng = int(one['n_guard'])
    assert all(int(p['n_guard']) == ng for p in parts) and ng == 64
    end = 512 + 2 * ng + 64 + ng // 2
    assert [int(p['Nz_local']) for p in parts] == [end] + [512 + 2 * ng] * 6 + [end]
    assert int(one['Nz_local']) == 4096 + 2 * ng + 2 * (64 + ng // 2)
    assert int(parts[0]['exchange_period']) < NSTEP
    assert abs(float(parts[0]['zmin']) - float(one['zmin'])) < 1e-12 * (ZMAX - ZMIN)   # same window motion
    for grp in ('E', 'B', 'J', 'r'):
        keys = ['%s_%d' % (k, m) for k in FIELDS if k[0] == grp for m in range(NM)]
        scale = max(np.abs(one[k]).max() for k in keys)
        assert scale > 0
        for k in keys:
            got = np.concatenate([p[k] for p in parts], axis=0)
            assert got.shape == one[k].shape == (NZ, NR)
            # (uncorrected: measured <= 6e-14; corrected: the scheme difference, measured ~1e-3 ... 1e-2
            # on J, far less on E, B after 16 steps)
            achieved(None, np.abs(got - one[k]).max() / scale, 5e-2 if correct else 1e-12, 'fields ' + grp)
    ref = np.array([one['p_' + k] for k in PTCL])
    got = np.concatenate([np.array([p['p_' + k] for k in PTCL]) for p in parts], axis=1)
    assert got.shape == ref.shape and ref.shape[1] > 9.0e6       # nobody lost, duplicated or mis-injected
    assert np.array_equal(np.sort(got[7]), np.sort(ref[7]))
    if correct:
        # (fields that differ at 1e-7 move the particles by as much: pairing them through a sort on
        # (w, x, y, z) is no longer stable - theta and -theta of a lattice ring share w and x - so the
        # two particle sets are compared attribute by attribute as sorted distributions)
        for j, k in enumerate(PTCL):
            achieved(None, np.abs(np.sort(got[j]) - np.sort(ref[j])).max() / max(np.abs(ref[j]).max(), 1e-300),
                     1e-3, 'particle distributions')
    else:
        o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
        o2 = np.lexsort((got[2], got[1], got[0], got[7]))
        for j, k in enumerate(PTCL):
            achieved(None, np.abs(got[j][o2] - ref[j][o1]).max() / max(np.abs(ref[j]).max(), 1e-300),
                     2.5e-12, 'particles')          # measured 2.5e-13
    del one, parts
    shutil.rmtree(outdir, ignore_errors=True)


def test_bench_strong_scaling_dry_run_on_8_ranks():
    """`bench.py --gpus 8 --scaling strong` (the fixed 1024 x 128 box of BASELINE.json cut into 8
    slabs of 128 + 2 x 64 rows) launched as the driver launches it, the 8 ranks sharing the GPU
    over gloo: the decomposed bench path runs end to end and prints its one JSON line."""
    env = dict(os.environ, FBPIC_AMD_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--scaling', 'strong', '--steps', '4',
           '--warmup', '2', '--no-kernel-timing', '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['scaling'] == 'strong' and out['steps'] == 4
    assert out['config']['particles'] == 4194304 and out['value'] > 0
    assert 'x8' in out['config']['parallelism']
    # the line carries the other scaling too (BASELINE's metric names the fixed box, the driver's
    # contract fixed work per GPU): here 8 slabs of 1024 rows
    other = out['other_scaling']
    assert other['scaling'] == 'weak' and other['particles'] == 8 * 4194304 and other['value'] > 0
    assert other['rows_per_rank'] == 1024
