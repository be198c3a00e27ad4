#!/usr/bin/env python3
"""Where do the 14 minutes of tests/test_gpu_c4_golden.py go?  The same 8 processes with time stamps of rank 0."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch.multiprocessing as mp


def run(rank, world, port):
    t0 = time.time()
    import torch
    torch.set_num_threads(2)
    import torch.distributed as dist
    from scipy.constants import c
    from fbpic_amd.main import Simulation
    from fbpic_amd.lpa_utils.laser import add_laser_pulse, GaussianLaser
    from conftest import golden
    import test_gpu_c4_golden as T
    g = golden(T.NAME)
    def say(what):
        if rank in (0, 7):
            print('rank %d %7.1f s  %s' % (rank, time.time() - t0, what), flush=True)
    say('imports')
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    say('process group')
    np.random.seed(0)
    sim = Simulation(int(g['Nz']), float(g['zmax']), int(g['Nr']), float(g['rmax']), 2, float(g['dt']), zmin=float(g['zmin']),
                     p_zmin=float(g['zmin']), p_zmax=float(g['zmax']), p_rmin=0., p_rmax=18.e-6, p_nz=2, p_nr=2, p_nt=4, n_e=4.e24,
                     dens_func=T._dens_func(g), n_order=int(g['n_order']), n_guard=int(g['n_guard']), particle_shape='linear',
                     boundaries={'z': 'open', 'r': 'reflective'})
    say('Simulation built')
    add_laser_pulse(sim, GaussianLaser(a0=4., waist=5.e-6, tau=16.e-15, z0=15.e-6))
    say('laser')
    sim.set_moving_window(v=c)
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        sim.step(1)
        say('step %d' % i)
    if rank == 0:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); sim.step(1); pr.disable()
        pstats.Stats(pr).sort_stats('cumtime').print_stats(25)
    else:
        sim.step(1)
    dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    from test_gpu_multirank_golden import _free_port
    ctx = mp.get_context('spawn')
    port = _free_port()
    ps = [ctx.Process(target=run, args=(r, 8, port)) for r in range(8)]
    [p.start() for p in ps]; [p.join() for p in ps]
