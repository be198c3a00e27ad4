#!/usr/bin/env python3
"""Where does the time of the FIRST particle hand-overs of a decomposed run go?  (loopback, one GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch, helpers
import loopback_multirank as lb
from fbpic_amd.main import GpuMemoryManager
from fbpic_amd.boundaries import boundary_communicator as bc
from fbpic_amd.boundaries import particle_buffer_handling as pbh
lb.install_loopback(bc, torch)
orig = pbh.exchange_particles_between_ranks
log = []


def timed(comm, species, fld, t_):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n0 = species.Ntot
    if len(log) == 1:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        orig(comm, species, fld, t_)
        torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(14)
    else:
        orig(comm, species, fld, t_)
    torch.cuda.synchronize(); log.append((1e3 * (time.perf_counter() - t0), n0, species.Ntot))
pbh.exchange_particles_between_ranks = timed
bc.exchange_particles_between_ranks = timed if hasattr(bc, 'exchange_particles_between_ranks') else None
sim = helpers.uniform_plasma_sim(2048, 128, 2, (2, 4, 4), 'linear', seed=0, n_order=32, n_guard=64)
with GpuMemoryManager(sim):
    ts = []
    for blk in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); sim.step(14); torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0) / 14)
print('ms/step per block of 14 steps:', ' '.join('%.3f' % v for v in ts))
print('hand-overs (ms, n before, n after):', log)
