#!/usr/bin/env python3
"""Cost of the particle hand-over step of the decomposed run (loopback transport on one GPU):
per-step wall time (synchronised) around an exchange iteration and a host profile of that step.
    python tools/exchange_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch, helpers
import loopback_multirank as lb
from fbpic_amd.main import GpuMemoryManager
from fbpic_amd.boundaries import boundary_communicator as bc
bc._dist = lambda: lb.FakeDist


def loopback(self, send_left, send_right, recv_left, recv_right, skip_empty=False):
    L_local = self._Nz_global_domain * self.dz / 2
    for recv, send in ((recv_left, send_right), (recv_right, send_left)):
        if recv is None or send is None or recv.numel() == 0:
            continue
        recv.copy_(send)
        if recv.dim() == 2 and recv.shape[0] == 8 and recv.dtype == torch.float64:
            recv[2] += L_local
bc.BoundaryCommunicator.exchange_domains = loopback
sim = helpers.uniform_plasma_sim(2048, 128, 2, (2, 4, 4), 'linear', seed=0, n_order=32, n_guard=64)
P = sim.comm.exchange_period
with GpuMemoryManager(sim):
    sim.step(2 * P + 2); torch.cuda.synchronize()
    # one call per step: each call pays the first-step extras (rho_prev deposit), so compare
    # the exchange step against its neighbours, all measured the same way
    ts = []
    for i in range(2 * P):
        t0 = time.perf_counter(); sim.step(1); torch.cuda.synchronize(); ts.append((sim.iteration, 1e3 * (time.perf_counter() - t0)))
    print('exchange_period', P, ' per-call ms:', ' '.join('%d:%.2f' % t for t in ts))
    # steps inside one call (no per-call extras): P steps with exactly one exchange vs P-1 without
    while sim.iteration % P != 1:
        sim.step(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sim.step(P - 1); torch.cuda.synchronize(); t_no = time.perf_counter() - t0
    t0 = time.perf_counter(); sim.step(P); torch.cuda.synchronize(); t_ex = time.perf_counter() - t0
    print('%d steps without exchange: %.3f ms/step; %d steps with one: %.3f ms/step -> exchange step costs %.2f ms more'
          % (P - 1, 1e3 * t_no / (P - 1), P, 1e3 * t_ex / P, 1e3 * (t_ex - t_no * P / (P - 1))))
    import cProfile, pstats
    while sim.iteration % P != P - 1:
        sim.step(1)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); sim.step(2); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('cumtime').print_stats(45)
