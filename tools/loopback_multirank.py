"""Profiling aid for the DECOMPOSED-domain step on a one-GPU box (8-GPU nodes are only
available to the round-end driver): rank 0 of a fake 2-rank periodic ring whose neighbour
is an identical copy of itself - every guard-cell / particle message is answered with this
rank's own outgoing message of the opposite side.  All kernels, packs/unpacks and host-side
work of the multi-rank path run; only the RCCL transport is replaced by a device copy.
Prints ms/step next to the single-domain step of the same per-rank size.

    python tools/loopback_multirank.py [--steps 30] [--Nz 1024]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class FakeDist:
    is_available = staticmethod(lambda: True)
    is_initialized = staticmethod(lambda: True)
    get_rank = staticmethod(lambda: 0)
    get_world_size = staticmethod(lambda: 2)
    get_backend = staticmethod(lambda: 'nccl')

    @staticmethod
    def all_gather_object(out, obj):        # the (host, gpu) uniqueness check of the communicator
        out[:] = [(obj[0], i) for i in range(len(out))]


def install_loopback(bc, torch):
    """Rank 0 of a fake 2-rank periodic ring: every message is answered with this rank's own
    outgoing message of the opposite side (device copy instead of the RCCL transport)."""
    bc._dist = lambda: FakeDist

    def loopback(self, send_left, send_right, recv_left, recv_right, skip_empty=False):
        L_local = self._Nz_global_domain * self.dz / 2
        for recv, send in ((recv_left, send_right), (recv_right, send_left)):
            if recv is None or send is None or recv.numel() == 0:
                continue
            recv.copy_(send)
            caps = getattr(self, '_handover_caps', None)
            if caps is not None and recv.dim() == 1:     # fixed-size hand-over message
                cap = (recv.numel() - 8) // 8
                recv[8 + 2 * cap:8 + 3 * cap] += L_local     # z row: re-enter on the other side
            elif recv.dim() == 2 and recv.shape[0] == 8 and recv.dtype == torch.float64:
                recv[2] += L_local      # particle payload: re-enter on the other side
    bc.BoundaryCommunicator.exchange_domains = loopback


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--Nz', type=int, default=1024)
    ap.add_argument('--Nr', type=int, default=128)
    ap.add_argument('--single', action='store_true', help='time the single-domain step instead')
    a = ap.parse_args()
    import torch
    import helpers
    from fbpic_amd.boundaries import boundary_communicator as bc
    from fbpic_amd.main import GpuMemoryManager

    if not a.single:
        install_loopback(bc, torch)
    world = 1 if a.single else 2
    sim = helpers.uniform_plasma_sim(a.Nz * world, a.Nr, 2, (2, 4, 4), 'linear', seed=0,
                                     n_order=(-1 if a.single else 32),
                                     n_guard=(None if a.single else 64))
    n = sum(s.Ntot for s in sim.ptcl)
    with GpuMemoryManager(sim):
        sim.step(a.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.step(a.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print('%s: Nz_local %d, %d particles, exchange_period %d: %.4f ms/step, %.3e updates/s per rank'
          % ('single domain' if a.single else 'decomposed (loopback)', sim.fld.Nz, n,
             sim.comm.exchange_period, 1e3 * dt / a.steps, n * a.steps / dt))
    print('   particle passes: one-pass %d, sorting two-pass %d' % (sum(s.cycle_passes for s in sim.ptcl),
                                                                      sum(s.cycle_sorts for s in sim.ptcl)))


if __name__ == '__main__':
    main()
