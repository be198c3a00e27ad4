#!/usr/bin/env python3
"""Benchmark of the FBPIC per-step PIC cycle on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: uniform plasma, Nz x Nr = 1024 x 128, Nm = 2,
32 particles per cell (p_nz, p_nr, p_nt = 2, 4, 4 -> 4 194 304 macroparticles per 1024
cells in z), dz = dr = 0.2 um, dt = dz/c, z-periodic, linear shape, curl-free correction,
filtered currents, thermal momenta N(0, 0.01^2).  A "step" is one full PIC cycle
(Simulation.step: sort, deposit rho/J, transforms, gather, Vay push, PSATD solve).
With N > 1 ranks the domain is decomposed in z (one rank per GPU, RCCL guard-cell
exchange).  --scaling weak (default): every rank owns a 1024-cell slab (global Nz = 1024 * N,
"C2 per rank x N"); --scaling strong: the fixed 1024 x 128 box of BASELINE.json is cut into N
slabs (n_order = 32, n_guard = 64: 128 + 2 x 64 local rows at N = 8).

Inside the timed region Simulation.step runs its default MI355X orchestration, which differs
from the reference's launch sequence in three sanctioned ways (results
equal within the stated tolerances, named in config.workload): rho_prev is not re-deposited
after the first step of a call on a single periodic domain, the iFFT/FFT round trip of E, B
on a single periodic domain is skipped, and the gathered E, B are written to the particle
arrays only on the last step.  --reference-sequence turns all of that (and the kernel
fusions) off: every operation of fbpic/main.py:346-586 is launched on its own.

Prints ONE JSON line on rank 0.  `value` = macroparticle updates per second over all
ranks with all data resident in HBM.  `roofline` is for the kernel with the largest
share of device time, measured with HIP events on the launch stream in an instrumented
pass after the timed region.  `cpu_baseline` times the CPU oracle (C/OpenMP restatement
of the reference's Numba CPU path + NumPy FFT/dot, `kind: port`) on the same workload.
"""
import argparse
import json
import os
import sys
import time

# The CPU baseline runs in a process of its own (`--cpu-baseline-only`, started by the default run): its
# OpenMP threads are bound to cores (SURVEY.md 8d recipe; docs/source/overview/parallelisation.rst:47-53 of
# the reference) - libgomp reads the two variables when it is first loaded (with `import torch`), and it
# pins the INITIAL thread as well, so in the GPU process they would cut `available_cores()` to one core and
# put NumPy's and scipy's thread pools on it (a first version of this did: `cores: 2`, transforms slower on
# n threads than on one).
_CPU_LEG = '--cpu-baseline-only' in sys.argv
_AFFINITY_AT_START = os.sched_getaffinity(0)
if _CPU_LEG:
    os.environ.setdefault('OMP_PROC_BIND', 'close')
    os.environ.setdefault('OMP_PLACES', 'cores')
    os.environ.setdefault('OMP_WAIT_POLICY', 'passive')     # (the team sleeps through the NumPy phases)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (datasheet)

# ALGORITHMIC work per launch, strictly as SURVEY.md 8d counts it: the union rule (8 B x distinct
# per-particle arrays read + distinct arrays written by the operations the kernel fuses).
# name -> (bound, function(args) -> bytes|flops)
WORK = {
    'fb_push_x': ('hbm', lambda a: 80.0 * a[0]),
    'fb_push_p': ('hbm', lambda a: 112.0 * a[0]),
    'fb_gather': ('hbm', lambda a: 72.0 * a[2]),
    # fused gather+push_p+push_x: 56 B read + 56 B written (+ 48 B when E,B are stored for an
    # observer: last iteration of a step() call), union rule of SURVEY.md 8d
    'fb_gather_push': ('hbm', lambda a: (160.0 if a[19] is not None else 112.0) * a[2]),
    # the first pass of the two-pass step: the same 112 B (the 8 B of cell + rank it also writes
    # are sort bytes: OVERHEAD below)
    'fb_gather_push_rank_next': ('hbm', lambda a: (160.0 if a[19] is not None else 112.0) * a[2]),
    # the one-pass cycle: gather + push_p + push_x + deposit J + push_x + deposit rho with every
    # attribute read once (x, y, z, ux, uy, uz, inv_gamma, w: 64 B) and written once (all but w:
    # 56 B) - the union rule of SURVEY.md 8d over the six operations; + 48 B when E, B are stored
    'fb_gather_push_deposit_J_rho': ('hbm', lambda a: (168.0 if a[21] is not None else 120.0) * a[2]),
    'fb_deposit_rho': ('hbm', lambda a: 32.0 * a[2]),
    'fb_deposit_J': ('hbm', lambda a: 64.0 * a[2]),
    'fb_deposit_J_rank_next': ('hbm', lambda a: 64.0 * a[2]),
    # push_x folded into the counting sort: the push_x bytes
    'fb_push_x_bin_sort_particles': ('hbm', lambda a: 80.0 * a[0]),
    # destination-ordered push_x + sort + rho deposition: x,y,z,ux,uy,uz,inv_gamma,w read, x,y,z
    # written = 64 R + 24 W ("two-pass fused lower bound 120 + (64 R + 24 W)")
    'fb_push_x_sort_deposit_rho': ('hbm', lambda a: 88.0 * a[0]),
    # the same pass with the J deposition riding along: same distinct arrays
    'fb_push_x_sort_deposit_J_rho': ('hbm', lambda a: 88.0 * a[0]),
    'fb_zfft': ('hbm', lambda a: 32.0 * a[0] * a[1]),
    'fb_shift_periodic': ('hbm', lambda a: 8.0 * a[0]),
    'fb_hankel': ('mfma', lambda a: 4.0 * a[7] * a[8] * a[8] * a[0]),
    # forward products of J (p, m, z) + rho and inverse products of E, B (p, m, z) of every mode:
    # 10 Nm transforms in one launch (the cell-local solver step between them is not counted)
    'fb_spect_cycle_standard': ('mfma', lambda a: 4.0 * a[19] * a[20] * a[20] * 10 * a[0]),
    # (p | m) formed in the operand load: same GEMM work per job
    'fb_hankel_rt_to_pm_scaled': ('mfma', lambda a: 4.0 * a[12] * a[13] * a[13] * a[0]),
    'fb_hankel_scaled': ('mfma', lambda a: 4.0 * a[10] * a[11] * a[11] * a[0]),
    # pair jobs carry two products
    'fb_hankel_pm_to_rt': ('mfma', lambda a: 4.0 * a[10] * a[11] * a[11]
                           * sum(2 if a[2][j] else 1 for j in range(a[0]))),
}
# Sort / permutation bytes the same launches move ("implementation overhead, not algorithmic
# work - report separately"): reported as `overhead_bytes` per launch and in
# `frac_incl_overhead`, never in `frac`.
OVERHEAD = {
    'fb_gather_push_deposit_J_rho': lambda a: 4.0 * a[2],      # home cell of every particle, read
    'fb_gather_push_rank_next': lambda a: 8.0 * a[2],          # cell + rank of the next sort, written
    'fb_deposit_J_rank_next': lambda a: 8.0 * a[2],
    # the sort pass re-writes the 5 arrays the push does not change (40 B) + reads cell and rank
    # (8 B) + writes sorted cell and permutation (8 B) + reads w (8 B)
    'fb_push_x_bin_sort_particles': lambda a: 64.0 * a[0],
    # permutation index read (4 B) + ux,uy,uz,inv_gamma,w re-written at the sorted slot (40 B)
    'fb_push_x_sort_deposit_rho': lambda a: 44.0 * a[0],
    'fb_push_x_sort_deposit_J_rho': lambda a: 44.0 * a[0],
    'fb_cell_index': lambda a: 32.0 * a[0],
    'fb_sort_by_cell': lambda a: 16.0 * a[0],
    'fb_permute': lambda a: (16.0 * a[2] + 4.0) * a[0],
    'fb_bin_sort_particles': lambda a: 144.0 * a[0],
}


def bench_c3(args, torch, world, rank):
    """BASELINE configs[2]: docs/source/example_input/lwfa_script.py at 4096 x 256, Nm = 2, 16 ppc,
    open z boundary with damping, moving window at c, continuous injection, a0 = 4 laser; the
    density ramp starts inside the initial box so that the window is full of plasma (> 9 M
    macroparticles) from the first timed step."""
    import numpy as np
    from scipy.constants import c
    from fbpic_amd import _capi
    from fbpic_amd.main import Simulation, GpuMemoryManager
    from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
    assert world == 1, 'C4 (the decomposed run) is launched like the default bench: --gpus N'
    zmin, zmax, rmax = -10.e-6, 30.e-6, 20.e-6
    Nz, Nr, Nm = 4096, 256, 2
    dt = (zmax - zmin) / Nz / c
    ramp_start, ramp_length = 5.e-6, 10.e-6

    def dens_func(z, r):
        n = np.ones_like(z)
        n = np.where(z < ramp_start + ramp_length, (z - ramp_start) / ramp_length, n)
        return np.where(z < ramp_start, 0., n)
    np.random.seed(0)
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dt, zmin=zmin, p_zmin=ramp_start, p_zmax=500.e-6,
                     p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24,
                     dens_func=dens_func, n_order=-1, particle_shape='linear',
                     boundaries={'z': 'open', 'r': 'reflective'})
    add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
    sim.set_moving_window(v=c)
    apply_policy(sim, args.policy)
    # the warm-up covers the first particle hand-over / injection (every `exchange_period` steps):
    # its buffers grow once (allocations), which is start-up cost, not the steady state
    warm = max(args.warmup, sim.comm.exchange_period + 2)
    with GpuMemoryManager(sim):
        sim.step(warm)
        finish_outputs(sim)     # the warm-up is the timed call's twin (first-use FFT plans included)
        torch.cuda.synchronize()
        n0 = sum(s.Ntot for s in sim.ptcl)
        t0 = time.perf_counter()
        sim.step(args.steps)
        finish_outputs(sim)
        torch.cuda.synchronize()
        dt_wall = time.perf_counter() - t0
        n1 = sum(s.Ntot for s in sim.ptcl)
        kern = None
        if not args.no_kernel_timing:
            _capi.enable_timing()
            sim.step(10)
            kern = _capi.collect_timing()
    npart = 0.5 * (n0 + n1)
    out = {
        'metric': 'particle-updates/sec', 'value': npart * args.steps / dt_wall,
        'unit': 'particle-updates/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': warm,
        'warmup_requested': args.warmup,
        'ms_per_step': 1e3 * dt_wall / args.steps, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'C3 laser-wakefield 4096x256 Nm=2 16 ppc linear shape, open z (n_guard 64, '
                               'n_damp 64, n_inject 32: %d local rows), moving window at c, continuous '
                               'injection, a0=4 Gaussian laser, standard PSATD n_order=-1; window '
                               'filled with plasma' % sim.fld.Nz,
                   'particles': int(npart), 'particles_start': n0, 'particles_end': n1,
                   'parallelism': 'z-slab x1', 'sequence': 'fused'},
    }
    if kern:
        ceil = measured_ceilings(torch)
        out['roofline'], out['kernels'] = roofline(kern, ceil, 'c3')
        out['measured_ceilings'] = ceil
    print(json.dumps(out))


_TWO_PASS = ('one-pass MI355X sequence: gather+push_p+push_x+J deposit+push_x+rho deposit of an iteration '
             'in ONE pass over the particles, whose arrays are re-sorted every 4th iteration only - '
             'that iteration runs the two-pass sequence (gather+push(+rank) | J deposit+push_x+sort+rho '
             'deposit) - both kinds inside the timed region (extra.particle_passes); forward Hankel of '
             'J,rho + PSATD step + inverse Hankel of E,B one launch; the timed call continues from the device state of '
             'the warm-up call (nobody touched the tensors in between), so its first iteration is an '
             'interior one; J, rho on the interpolation grid and E,B on the particles, which step() defers '
             'to their first read, ARE read inside the timed region; sanctioned skips inside the timed '
             'region: rho_prev re-deposit, the diagnostics-only J deposit at the first step of '
             'a call (no diagnostic is registered), ')
SEQUENCE_NOTE = {
    False: _TWO_PASS + 'identity iFFT/FFT of E,B on the single periodic domain, gathered E,B '
                       'stored on the last step only',
    # z-slab decomposition: the guard exchange of E,B needs the interpolation grid each step
    'decomposed': _TWO_PASS + 'gathered E,B stored on the last step only (the E,B round trip '
                              'through the interpolation grid is NOT skipped: its guard cells are exchanged)',
    True: "reference sequence (fbpic/main.py:346-586): every operation its own launch, rho_prev "
          "re-deposited every step, E,B stored by every gather",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--Nz', type=int, default=1024)
    ap.add_argument('--Nr', type=int, default=128)
    ap.add_argument('--Nm', type=int, default=2)
    ap.add_argument('--shape', default='linear')
    ap.add_argument('--ppc', default='2,4,4')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='(internal) build the workload, time the CPU oracle on it, print its JSON object')
    ap.add_argument('--cpu-steps', type=int, default=10)   # ~10 s of CPU work on 16 cores
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-side-legs', action='store_true',
                    help='skip the uncarried-call and reference-sequence legs behind the headline')
    ap.add_argument('--resort-fragmentation', type=float, default=None,
                    help='override Particles.resort_fragmentation (adaptive sort policy)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--config', choices=('C2', 'C3', 'C5'), default='C2',
                    help='BASELINE.json configuration: C2 = configs[1] (the one the metric is quoted '
                         'on), C5 = configs[4] (2048x512 Nm=4 cubic 64 ppc), C3 = configs[2] (laser-'
                         'wakefield 4096x256, moving window, window filled with plasma)')
    ap.add_argument('--policy', default='',
                    help='sort-policy attributes of every species, e.g. bad=2.0,stray=0.5,period=3,suspend=0 '
                         '(cycle_bad_limit, cycle_stray_limit, cycle_sort_period, cycle_suspend_iterations)')
    ap.add_argument('--reference-sequence', action='store_true',
                    help="the reference's launch sequence: no fusion, rho_prev re-deposited every step")
    return ap.parse_args()


def set_reference_sequence(sm, on):
    """The reference's launch sequence (every operation of fbpic/main.py:346-586 its own launch, rho_prev
    re-deposited every step, identity FFT round trip kept) on or off."""
    sm.redeposit_rho_prev_every_step = bool(on)
    sm.fuse_gather_push = not on
    sm.prerank_in_deposit = not on
    sm.reference_sequence = bool(on)
    for sp in sm.ptcl:
        sp.fuse_sort_deposit_rho = not on


def apply_policy(sm, spec):
    names = {'bad': 'cycle_bad_limit', 'stray': 'cycle_stray_limit', 'period': 'cycle_sort_period',
             'suspend': 'cycle_suspend_iterations', 'homesort': 'record_home_in_rho_sort'}
    for item in filter(None, spec.split(',')):
        k, v = item.split('=')
        for sp in sm.ptcl:
            setattr(sp, names[k], int(v) if k in ('period', 'suspend', 'homesort') else float(v))


def config_name(args, ppc, world):
    """Which entry of BASELINE.json's `configs` the workload is (C2 = configs[1], the one the
    metric is quoted on; C5 = configs[4]); anything else is a custom size.  A weak-scaling run
    on N > 1 ranks is N slabs of that size, not the configuration itself."""
    key = (args.Nz, args.Nr, args.Nm, ppc[0] * ppc[1] * ppc[2], args.shape)
    name = {(1024, 128, 2, 32, 'linear'): 'C2', (2048, 512, 4, 64, 'cubic'): 'C5'}.get(key, 'custom')
    if world > 1 and args.scaling == 'weak':
        return '%s-per-rank x%d' % (name, world)
    return name


def cpu_baseline_in_subprocess(args):
    """The cpu_baseline object of the line, measured by `bench.py --cpu-baseline-only` in a process of its
    own (bound OpenMP threads, no GPU runtime in it) on the same workload."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--Nz', str(args.Nz), '--Nr', str(args.Nr),
           '--Nm', str(args.Nm), '--shape', args.shape, '--ppc', args.ppc, '--cpu-steps', str(args.cpu_steps)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        return json.loads(r.stdout.strip().split('\n')[-1])
    except Exception as exc:            # the GPU line stands on its own
        return {'error': repr(exc)[:300]}


def main():
    args = parse()
    if args.cpu_baseline_only:
        import helpers
        ppc = tuple(int(v) for v in args.ppc.split(','))
        sm = helpers.uniform_plasma_sim(args.Nz, args.Nr, args.Nm, ppc, args.shape, seed=0, n_order=-1)
        print(json.dumps(cpu_baseline(sm, args)))
        return
    import numpy as np
    import torch
    import helpers
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # 'nccl' is RCCL on ROCm (xGMI).  FBPIC_AMD_DIST_BACKEND=gloo exists only to smoke-test
        # the multi-rank path on a single-GPU box (host-staged transport).
        backend = os.environ.get('FBPIC_AMD_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node == --gpus'
    from fbpic_amd import _capi
    from fbpic_amd.main import GpuMemoryManager
    if args.config == 'C5':
        args.Nz, args.Nr, args.Nm, args.shape, args.ppc = 2048, 512, 4, 'cubic', '2,2,16'
    ppc = tuple(int(v) for v in args.ppc.split(','))
    n_order = -1 if world == 1 else 32
    if args.config == 'C3':
        return bench_c3(args, torch, world, rank)
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build(scaling):
        # weak scaling: every rank owns args.Nz cells; strong: the args.Nz cells are divided.
        # The Simulation is given the global box either way.
        Nz_g = args.Nz * world if scaling == 'weak' else args.Nz
        sm = helpers.uniform_plasma_sim(Nz_g, args.Nr, args.Nm, ppc, args.shape, seed=0,
                                        n_order=n_order, n_guard=(None if world == 1 else 64))
        if args.reference_sequence:
            set_reference_sequence(sm, True)
        apply_policy(sm, args.policy)
        if args.resort_fragmentation is not None:
            for sp in sm.ptcl:
                sp.resort_fragmentation = args.resort_fragmentation
        n_loc = sum(s_.Ntot for s_ in sm.ptcl)
        n_tot = n_loc
        if world > 1:
            tcount = torch.tensor([n_loc], dtype=torch.int64,
                                  device=('cuda' if dist.get_backend() == 'nccl' else 'cpu'))
            dist.all_reduce(tcount)
            n_tot = int(tcount.item())
        return sm, Nz_g, n_tot

    def max_over_ranks(seconds):
        if world > 1:
            tt = torch.tensor([seconds], dtype=torch.float64,
                              device=('cuda' if dist.get_backend() == 'nccl' else 'cpu'))
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return seconds

    sim, Nz_global, n_total = build(args.scaling)
    cpu_base = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        cpu_base = cpu_baseline_in_subprocess(args)
        if 'value' not in cpu_base:
            # (the leg must not cost the line its baseline: measured in this process instead, threads unbound)
            failed = cpu_base
            cpu_base = cpu_baseline(sim, args)
            cpu_base['subprocess_error'] = failed.get('error')

    # decomposed run: the warm-up also covers the first particle hand-over between the ranks
    # (every `exchange_period` steps; its first execution pays one-time start-up costs, ~3 ms)
    warm = args.warmup if world == 1 else max(args.warmup, sim.comm.exchange_period + 2)
    # The ceilings of this box (fp64 triad, 4096^3 dgemm) are measured BEFORE the timed region, on
    # every run whatever --no-kernel-timing says (extra.preheat): ~1 s of full load brings the GPU out
    # of its idle clock state, which a 5-step warm-up (2 ms) does not - extra.repeat_ms_per_step of
    # round-3-style lines fell 0.48 -> 0.45 -> 0.44 ms over three back-to-back timed calls of the
    # same kernels.
    with GpuMemoryManager(sim):
        # (inside the block: after the host -> device copies, nothing but the warm-up steps between
        # this load and the timed region)
        _sysfs_card()           # (resolves the device's sysfs directory once: a rocm-smi process)
        ceil = measured_ceilings(torch)
        sim.step(warm)
        finish_outputs(sim)     # the warm-up is the timed call's twin
        barrier()
        clocks_before = device_clocks()
        t0 = time.perf_counter()
        sim.step(args.steps)
        finish_outputs(sim)
        barrier()
        dt_wall = time.perf_counter() - t0
        clocks_first = device_clocks()
        # the same timed call twice more, back to back (the headline stays the first): tells a slow
        # box / clock state from a slow build when the line moves between runs of unchanged kernels
        repeats = [1e3 * dt_wall / args.steps]
        for _ in range(2 if world == 1 else 0):
            barrier()
            t1 = time.perf_counter()
            sim.step(args.steps)
            finish_outputs(sim)
            barrier()
            repeats.append(1e3 * (time.perf_counter() - t1) / args.steps)
        clocks_after = device_clocks()
        kern = None
        if not args.no_kernel_timing:
            _capi.enable_timing()
            sim.step(10)     # enough launches for a stable mean (durations vary ~15% per step)
            kern = _capi.collect_timing()
        # Two more figures next to the headline, same state, same steps, AFTER everything the line's
        # own numbers come from (extra.per_call_prologue_ms, extra.reference_sequence_ms_per_step):
        #  * what one step() call costs when the state is NOT carried from the previous call - the
        #    reference's per-call prologue (main.py:403-459: E, B exchange + transform, particle
        #    exchange, rho_prev deposit) - against the carried call;
        #  * the reference's own launch sequence on these kernels.
        side = {}
        if world == 1 and not args.reference_sequence and not args.no_side_legs:
            def timed_call():
                barrier()
                ts = time.perf_counter()
                sim.step(args.steps)
                finish_outputs(sim)
                barrier()
                return 1e3 * (time.perf_counter() - ts)
            carried = timed_call()
            sim.carry_state_between_calls = False
            timed_call()
            uncarried = timed_call()
            sim.carry_state_between_calls = True
            side['per_call_prologue_ms'] = uncarried - carried
            side['uncarried_call_ms_per_step'] = uncarried / args.steps
            side['carried_call_ms_per_step'] = carried / args.steps
            set_reference_sequence(sim, True)
            sim.carry_state_between_calls = False
            timed_call()
            side['reference_sequence_ms_per_step'] = timed_call() / args.steps
            set_reference_sequence(sim, False)
            sim.carry_state_between_calls = True
    dt_wall = max_over_ranks(dt_wall)
    passes = {'one_pass': sum(s_.cycle_passes for s_ in sim.ptcl),
              # sorts that record home cells: the sorting (two-pass) iterations + the sort in front of a call's
              # / hand-over's rho_prev deposition
              'sorts': sum(s_.cycle_sorts for s_ in sim.ptcl),
              'sort_period': sim.ptcl[0].cycle_sort_period if sim.ptcl else None,
              'last_stray_fraction': max((s_.cycle_last_stray_fraction or 0.) for s_ in sim.ptcl)
              if sim.ptcl else None}
    # N > 1: BASELINE.json's metric names the FIXED 1024 x 128 box on 1 -> 8 GPUs (strong scaling),
    # the driver's contract asks for per-GPU work that stays fixed (weak): the line carries both -
    # `value` / `scaling` are the run --scaling selects (default weak), `other_scaling` the other
    # one, same steps, same barriers, max over ranks.
    other = None
    if world > 1 and not args.reference_sequence:
        oscal = 'strong' if args.scaling == 'weak' else 'weak'
        sim = None
        try:
            sim2, Nz2, n2 = build(oscal)
            warm2 = max(args.warmup, sim2.comm.exchange_period + 2)
            with GpuMemoryManager(sim2):
                sim2.step(warm2)
                finish_outputs(sim2)
                barrier()
                t2 = time.perf_counter()
                sim2.step(args.steps)
                finish_outputs(sim2)
                barrier()
                dt2 = max_over_ranks(time.perf_counter() - t2)
            other = {'scaling': oscal, 'value': n2 * args.steps / dt2, 'unit': 'particle-updates/s',
                     'ms_per_step': 1e3 * dt2 / args.steps, 'steps': args.steps, 'warmup': warm2,
                     'particles': n2, 'global_grid': [Nz2, args.Nr],
                     'rows_per_rank': Nz2 // world, 'guard_rows_per_side': 64}
        except Exception as exc:          # the headline run above stands on its own
            other = {'scaling': oscal, 'error': repr(exc)[:300]}
    if rank != 0:
        return
    value = n_total * args.steps / dt_wall
    out = {
        'metric': 'particle-updates/sec', 'value': value, 'unit': 'particle-updates/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': warm, 'warmup_requested': args.warmup,
        'ms_per_step': 1e3 * dt_wall / args.steps, 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'ns_per_particle_step': 1e9 * dt_wall / (args.steps * n_total) * world,
        'config': {'workload': '%s uniform plasma %dx%d Nm=%d %d ppc %s shape, z-periodic, '
                               'standard PSATD n_order=%d, curl-free correction, filtered; %s'
                               % (config_name(args, ppc, world), Nz_global, args.Nr, args.Nm,
                                  ppc[0] * ppc[1] * ppc[2], args.shape, n_order,
                                  SEQUENCE_NOTE[True if args.reference_sequence else ('decomposed' if world > 1 else False)]),
                   'particles': n_total, 'parallelism': 'z-slab x%d' % world,
                   'sequence': 'reference' if args.reference_sequence else 'fused'},
    }
    if other:
        out['other_scaling'] = other
    out['extra'] = {'repeat_ms_per_step': repeats, 'device': device_identity(torch),
                    'preheat': 'fp64 triad + 4096^3 dgemm (measured_ceilings, ~1 s) before the warm-up, '
                               'whatever --no-kernel-timing says',
                    'clocks_before': clocks_before, 'clocks_after_headline_call': clocks_first,
                    'clocks_after': clocks_after,
                    'particle_passes': passes,
                    'build': _capi.lib().fb_build_info().decode()}
    out['extra'].update(side)
    if kern:
        out['roofline'], out['kernels'] = roofline(kern, ceil, {'C2': True, 'C5': 'c5'}.get(config_name(args, ppc, world), False))
        dp_issue_floor(out['roofline'], kern, clocks_first or clocks_after, out['extra']['device'],
                       config_name(args, ppc, world))
        tb = out['roofline'].get('traffic_box')
        if out['roofline'].get('traffic') is not None:
            same = bool(tb) and tb.get('unique_id_suffix') == out['extra']['device'].get('unique_id_suffix')
            out['roofline']['traffic_stale'] = not same     # counters of another run on another box of the pool
    out['measured_ceilings'] = ceil
    if cpu_base:
        out['cpu_baseline'] = cpu_base
    print(json.dumps(out))


def _rocm_smi(*flags):
    """Best-effort `rocm-smi <flags> --json` of the first device ({} when the tool is missing)."""
    import subprocess
    try:
        r = subprocess.run(['rocm-smi', *flags, '--json'], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        return d.get('card0', next(iter(d.values()))) if isinstance(d, dict) and d else {}
    except Exception:
        return {}


_SYSFS_CARD = []


def _sysfs_card():
    """/sys/class/drm/cardN/device of the GPU rocm-smi calls card0 (matched by unique id: the host's
    other GPUs are visible in sysfs too); None when it cannot be resolved."""
    if not _SYSFS_CARD and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        _SYSFS_CARD.append(None)        # (rocm-smi's card0 is not this rank's GPU: no clocks reported)
    if not _SYSFS_CARD:
        import glob
        def as_int(text):
            try:
                return int(str(text).strip(), 16)
            except ValueError:
                return None
        uid = as_int(_rocm_smi('--showuniqueid').get('Unique ID', ''))
        found = None
        for d in glob.glob('/sys/class/drm/card*/device'):
            try:
                u = as_int(open(d + '/unique_id').read())
            except OSError:
                continue
            if uid is not None and u == uid:      # (as numbers: one of the two drops leading zeros)
                found = d
        _SYSFS_CARD.append(found)
    return _SYSFS_CARD[0]


def device_clocks():
    """Current sclk / mclk (MHz) from the starred line of pp_dpm_sclk / pp_dpm_mclk in sysfs: a file
    read, so it can sit right behind a timed call without leaving the GPU idle (a rocm-smi process
    in that place costs the next call its clock state). {} when sysfs is not reachable."""
    d = _sysfs_card()
    out = {}
    if d is None:
        return out
    for name in ('sclk', 'mclk'):
        try:
            for line in open('%s/pp_dpm_%s' % (d, name)):
                if '*' in line:
                    out[name + '_MHz'] = int(''.join(ch for ch in line.split(':')[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
    return out


def device_identity(torch):
    p = torch.cuda.get_device_properties(torch.cuda.current_device())
    ident = {'name': p.name, 'compute_units': p.multi_processor_count,
             'hip': getattr(torch.version, 'hip', None)}
    d = _rocm_smi('--showuniqueid', '--showserial')
    for k, v in d.items():
        if 'unique' in k.lower():
            ident['unique_id_suffix'] = str(v)[-6:]
        elif 'serial' in k.lower():
            ident['serial_suffix'] = str(v)[-6:]
    return ident


def finish_outputs(sim):
    """Everything a step() call of the reference leaves behind (main.py:572-586: J and rho back on
    the interpolation grid; the gathered E, B on the particles) is produced inside the timed
    region: Simulation.step defers these to their first read, so read them."""
    sim.fld.materialize_sources()
    for sp in sim.ptcl:
        if sp.q != 0:
            sp.Ex


def measured_ceilings(torch):
    """Ceilings of THIS box, measured next to the run (SURVEY.md 8d asks for the fraction of
    both the datasheet and the measured ceiling): fp64 stream triad a = b + s*c over 3 x 256 MB
    and a 4096^3 fp64 GEMM (rocBLAS through torch.mm)."""
    n = 32 * 1024 * 1024
    a = torch.empty(n, dtype=torch.float64, device='cuda')
    b = torch.ones_like(a)
    c = torch.ones_like(a)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps
    t_triad = timed(lambda: torch.add(b, c, alpha=1.5, out=a), 20)
    del a, b, c
    m = 4096
    x = torch.randn(m, m, dtype=torch.float64, device='cuda')
    y = torch.randn(m, m, dtype=torch.float64, device='cuda')
    z = torch.empty_like(x)
    t_gemm = timed(lambda: torch.mm(x, y, out=z), 5)
    return {'triad_GBs': 3 * 8 * n / t_triad / 1e9, 'dgemm_4096_TFLOPs': 2. * m**3 / t_gemm / 1e12}


def roofline(kern, ceil=None, profiled_workload=True):
    """Per entry point: launches, mean device ms, achieved GB/s or TFLOP/s; the roofline
    object describes the entry point with the largest total device time (gather+push when it
    is within 5 % of it)."""
    table = {}
    for name, recs in kern.items():
        ms = [r[0] for r in recs]
        tot = sum(ms)
        ent = {'launches': len(ms), 'mean_ms': tot / max(len(ms), 1), 'total_ms': tot}
        over = sum(OVERHEAD[name](r[1]) for r in recs) if name in OVERHEAD else 0.
        if name in WORK:
            bound, fn = WORK[name]
            work = sum(fn(r[1]) for r in recs)
            if bound == 'hbm':
                ent.update(bound='hbm', achieved=work / (tot * 1e-3) / 1e9, unit='GB/s',
                           peak=HBM_PEAK_GBS, algorithmic_bytes=work / len(ms))
                if over:
                    ent['overhead_bytes'] = over / len(ms)
                    ent['frac_incl_overhead'] = (work + over) / (tot * 1e-3) / 1e9 / HBM_PEAK_GBS
            else:
                ent.update(bound='mfma', achieved=work / (tot * 1e-3) / 1e12, unit='TFLOP/s',
                           peak=FP64_MFMA_PEAK_TFLOPS)
            ent['frac'] = ent['achieved'] / ent['peak']
            if ceil:
                meas = ceil['triad_GBs'] if bound == 'hbm' else ceil['dgemm_4096_TFLOPs']
                ent['frac_of_measured'] = ent['achieved'] / meas
        elif over:
            # pure sort launches: no algorithmic work, only overhead bytes
            ent.update(overhead_bytes=over / len(ms), overhead_GBs=over / (tot * 1e-3) / 1e9)
        table[name] = ent
    cand = sorted((n for n in table if 'frac' in table[n]), key=lambda n: -table[n]['total_ms'])
    dom = cand[0]
    # When gather+push is within 5 % of the longest entry point (the unfused sequence: gather,
    # J deposition and sort take the same time to within box-to-box noise), it is the one
    # described: it is the kernel the target names (">= 70 % of the HBM roofline on
    # gather/push"), and the line stays comparable run to run.  In the two-pass sequence the
    # deposition pass is the longest and is the one described.  All are in `kernels` either way.
    for n in ('fb_gather_push_rank_next', 'fb_gather_push', 'fb_gather'):
        if n in table and 'frac' in table[n] and \
                table[n]['total_ms'] >= 0.95 * table[cand[0]]['total_ms']:
            dom = n
            break
    d = table[dom]
    roof = {'kernel': dom, 'bound': d['bound'], 'achieved': d['achieved'], 'peak': d['peak'],
            'unit': d['unit'], 'frac': d['frac'], 'traffic': None,
            'mean_launch_ms': d['mean_ms']}
    for k in ('frac_of_measured', 'algorithmic_bytes', 'overhead_bytes', 'frac_incl_overhead'):
        if k in d:
            roof[k] = d[k]
    if profiled_workload:   # tag of the PMC passes under profiles/ for this workload ('' = C2)
        roof.update(pmc_traffic(dom, '' if profiled_workload is True else profiled_workload))
    # the other particle pass of the two-pass step next to it (the target names gather/push)
    for n in ('fb_gather_push_rank_next', 'fb_gather_push'):
        if n in table and n != dom and 'frac' in table[n]:
            roof['gather_push'] = {k: table[n][k] for k in
                                   ('achieved', 'frac', 'frac_of_measured', 'algorithmic_bytes',
                                    'overhead_bytes', 'frac_incl_overhead', 'mean_ms') if k in table[n]}
            if profiled_workload:
                roof['gather_push'].update(pmc_traffic(n, '' if profiled_workload is True else profiled_workload))
            break
    # the Hankel GEMM (the MFMA-bound kernel of the path) next to it: all launches together
    hk = [table[n] for n in table if n.startswith(('fb_hankel', 'fb_spect_cycle')) and 'achieved' in table[n]]
    if hk:
        tot_ms = sum(e['total_ms'] for e in hk)
        flops = sum(e['achieved'] * 1e12 * e['total_ms'] * 1e-3 for e in hk)
        roof['hankel'] = {'bound': 'mfma', 'achieved': flops / (tot_ms * 1e-3) / 1e12,
                          'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': flops / (tot_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                          'launches': sum(e['launches'] for e in hk)}
        if ceil:
            roof['hankel']['frac_of_measured'] = roof['hankel']['achieved'] / ceil['dgemm_4096_TFLOPs']
    compact = {n: {k: (round(v, 5) if isinstance(v, float) else v) for k, v in e.items()
                   if k in ('launches', 'mean_ms', 'achieved', 'unit', 'frac', 'frac_of_measured',
                            'frac_incl_overhead', 'overhead_GBs')}
               for n, e in table.items()}
    return roof, compact


def dp_issue_floor(roof, kern, clocks, device, config=None):
    """The second, compute-side floor of the dominant particle kernel.  On gfx950 fp64 VALU instructions and
    fp64 MFMAs share ONE pipe per SIMD (tools/overlap_probe.hip): a wave64 VALU
    instruction occupies it for 4 cycles, v_mfma_f64_4x4x4 for 16.  With the instruction counts per chunk of
    64 particles from the SQ counter passes committed under profiles/ (SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES
    per wave-chunk: profiles/r06_dp_issue_counts.json), a launch cannot be shorter than
        chunks x (4 VALU + MFMA cycles) / (CUs x 4 SIMDs) / sclk.
    `frac_of_dp_floor` = that floor / the measured mean launch: the kernel is bound by this pipe and by
    the latency chain of its waves, not by HBM - `frac` (of the 8 TB/s roof) is kept because the metric's
    target names it."""
    try:
        counts = json.load(open(os.path.join(ROOT, 'profiles', 'r06_dp_issue_counts.json')))
    except (OSError, ValueError):
        return
    sclk = (clocks or {}).get('sclk_MHz')
    cus = device.get('compute_units', 256)
    cfg = None if config is None else str(config).split('-per-rank')[0]      # ('C2-per-rank x8': N slabs of C2)

    def price(target, entry, mean_ms):
        # (the counts of an entry point belong to ONE template instance: those of another configuration's are not used)
        c = counts.get(entry)
        if not c or entry not in kern or not mean_ms or c.get('config', cfg) != cfg:
            return
        recs = kern[entry]
        npart = sum(r[1][c.get('n_arg', 2)] for r in recs) / len(recs)
        chunks = (npart + 63) // 64
        cyc = chunks * (4.0 * c['valu_per_chunk'] + c['mfma_cycles_per_chunk']) / (cus * 4)
        for label, mhz in (('', sclk), ('_at_2400MHz', 2400)):
            if mhz:
                us = cyc / mhz
                target['dp_issue_floor_us' + label] = us
                target['frac_of_dp_floor' + label] = us / (1e3 * mean_ms)
        target['dp_issue_counts'] = c
    price(roof, roof['kernel'], roof.get('mean_launch_ms'))
    # C5: the first pass of its two (roofline.gather_push) next to the dominant second one
    if isinstance(roof.get('gather_push'), dict):
        price(roof['gather_push'], 'fb_gather_push_rank_next', roof['gather_push'].get('mean_ms'))


# entry point -> substrings identifying its dominant device kernel in the rocprofv3 summaries
_KERNEL_OF = {
    'fb_gather_push': ('k_gather<',), 'fb_gather': ('k_gather<',),
    'fb_gather_push_rank_next': ('k_gather',),
    'fb_push_x_sort_deposit_J_rho': ('k_perm_deposit_J_rho',),
    'fb_gather_push_deposit_J_rho': ('k_cycle_linear<',),
    'fb_deposit_J_rank_next': ('k_deposit<', ', 3, ', 'true, true>'),
    'fb_deposit_J': ('k_deposit<', ', 3, '), 'fb_deposit_rho': ('k_deposit<', ', 1, '),
    'fb_push_x_bin_sort_particles': ('k_scatter<true>',), 'fb_push_x': ('k_push_x',),
    'fb_push_x_sort_deposit_rho': ('k_deposit<', ', 1, ', 'false, true>'),
    'fb_push_p': ('k_push_p',),
}


def pmc_traffic(entry, tag=''):
    """HBM-side bytes per launch of the dominant kernel from the PMC passes committed under
    profiles/ (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of this same
    command: counters cannot be collected inside the timed run).  `tag`: '' for the default (C2)
    command, 'c5' / 'c3' for the passes of --config C5 / C3 (profiles/rNN_vM_<tag>_pmc_*.csv).
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated here on
    fb_push_x: 2 x FETCH_SIZE = 56 B / particle exactly); both counters are in KiB."""
    import csv
    import glob
    import re
    keys = _KERNEL_OF.get(entry)
    mid = ('_%s' % tag) if tag else ''
    pat = re.compile(r'^r\d+_v\d+%s_pmc_(fetch|write)_size\.csv$' % mid)
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_*_size.csv'))
                   if pat.match(os.path.basename(f)))
    fetch = [os.path.join(ROOT, 'profiles', f) for f in names if 'fetch' in f]
    write = [os.path.join(ROOT, 'profiles', f) for f in names if 'write' in f]
    if not keys or not fetch or not write:
        return {}

    def pick(path):
        best = None
        for row in csv.DictReader(open(path)):
            if all(k in row['kernel'] for k in keys):
                if best is None or int(row['dispatches']) > int(best['dispatches']):
                    best = row
        return best
    f, w = pick(fetch[-1]), pick(write[-1])
    if f is None or w is None:
        return {}
    out = {'traffic': (2.0 * float(f['avg_value']) + float(w['avg_value'])) * 1024.0,
           'traffic_source': '%s + %s (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B, per launch of %s)'
                             % (os.path.basename(fetch[-1]), os.path.basename(write[-1]),
                                f['kernel'])}
    # the box the counter passes ran on (tools/profile_round.sh writes it next to the CSVs): a static
    # figure of another run, usually of another box of the pool than this line's
    box = fetch[-1].replace('_pmc_fetch_size.csv', '_pmc_box.json')
    try:
        out['traffic_box'] = json.load(open(box))
    except (OSError, ValueError):
        out['traffic_box'] = None
    return out


def available_cores():
    """Cores this process may really use: CPU affinity capped by the cgroup CPU quota (the GPU
    boxes expose 256 logical CPUs but grant a quota of 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(sim, args):
    """CPU oracle (port of the reference's Numba-threaded CPU path) on the same workload:
    `cpu_steps` full PIC cycles at full size, one thread per available core, bound to cores
    (OMP_PROC_BIND / OMP_PLACES, top of this file), the C kernels compiled for this host
    (-O3 -march=native, oracle.build_native), the z-FFT on a thread pool as the reference plans
    FFTW with threads = nthreads (fourier.py:59-96), np.dot on the BLAS threads NumPy brings."""
    from oracle import oracle as orc
    mask = _AFFINITY_AT_START
    try:
        os.sched_setaffinity(0, mask)      # (whatever an import has pinned the initial thread to)
    except OSError:
        pass
    native = orc.build_native()
    if native:
        orc.use_library(native)
    nthreads = min(orc.max_threads(), available_cores())
    # libgomp pins the INITIAL thread too when it binds its team (first parallel region): every thread
    # pool created later by that thread - NumPy's BLAS, scipy.fft's workers - would inherit a one-core
    # mask and the transforms would run slower on n threads than on one (measured: 0.40 against 0.19 s
    # per step).  Start the team now, then give the initial thread its mask back; the workers stay pinned.
    orc.set_threads(nthreads)
    orc.push_x(*[__import__('numpy').zeros(4096) for _ in range(7)], 0.)
    try:
        os.sched_setaffinity(0, mask)
    except OSError:
        pass
    orc.FFT_WORKERS = nthreads
    o = orc.from_sim(sim, nthreads=nthreads)
    o.step(1)                       # warm-up (thread pools, FFT plans)
    o.phase_seconds.clear()
    t0 = time.perf_counter()
    o.step(args.cpu_steps)
    dt = time.perf_counter() - t0
    phases = {k: round(v, 3) for k, v in sorted(o.phase_seconds.items())}
    n = sum(s['x'].size for s in o.species)
    # single-thread figure: one more step of the same state on one core
    orc.set_threads(1)
    orc.FFT_WORKERS = 1
    o.nthreads = 1
    o.glob = [g[:1] for g in o.glob]
    o.phase_seconds.clear()
    t1 = time.perf_counter()
    o.step(1)
    dt1 = time.perf_counter() - t1
    phases1 = {k: round(v, 3) for k, v in sorted(o.phase_seconds.items())}
    return {'value': n * args.cpu_steps / dt, 'unit': 'particle-updates/s', 'cores': nthreads,
            'single_thread': n / dt1, 'cpu_model': cpu_model(), 'kind': 'port',
            'sample': '%d full PIC steps of the same %dx%d Nm=%d %d-particle workload '
                      '(after 1 warm-up step), C/OpenMP oracle + NumPy FFT/dot'
                      % (args.cpu_steps, sim.fld.Nz, sim.fld.Nr, sim.fld.Nm, n),
            'seconds': dt,
            'build': ('gcc -O3 -march=native -ffp-contract=off -fopenmp (built on this host)' if native
                      else 'checker build (oracle/Makefile flags): no compiler on this host'),
            'omp': {k: os.environ.get(k) for k in ('OMP_PROC_BIND', 'OMP_PLACES', 'OMP_WAIT_POLICY')},
            'fft_threads': nthreads,
            'phase_seconds': phases, 'phase_seconds_single_thread_step': phases1,
            'note': 'phases: particle kernels = gather, push, deposit (thread-private grids), reduce (their '
                    'sum + divide by volume), erase; transforms = z-FFT (threaded pocketfft) + Hankel np.dot; '
                    'field kernels = filter, correction, push_eb'}


if __name__ == '__main__':
    main()
