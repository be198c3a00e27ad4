#!/usr/bin/env python3
"""Measured ceilings of the box the benchmark runs on (SURVEY.md 8d asks for the fraction of
both the datasheet and the measured ceilings): fp64 stream triad through torch (HBM), fp64
GEMM through torch.matmul (rocBLAS / hipBLASLt: MFMA), and the CPU oracle at 1 / 64 / all
threads on the C2 workload."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def gpu():
    import torch
    out = {}
    n = 1 << 27                                    # 3 x 1 GiB of fp64
    a = torch.empty(n, dtype=torch.float64, device='cuda')
    b = torch.rand(n, dtype=torch.float64, device='cuda')
    c = torch.rand(n, dtype=torch.float64, device='cuda')
    for _ in range(3):
        torch.add(b, c, alpha=1.5, out=a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.add(b, c, alpha=1.5, out=a)
    e1.record(); torch.cuda.synchronize()
    out['triad_GBps'] = 20 * 3 * 8 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b, c
    for m in (4096, 8192):
        x = torch.rand((m, m), dtype=torch.float64, device='cuda')
        y = torch.rand((m, m), dtype=torch.float64, device='cuda')
        z = torch.empty_like(x)
        for _ in range(2):
            torch.matmul(x, y, out=z)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            torch.matmul(x, y, out=z)
        e1.record(); torch.cuda.synchronize()
        out['dgemm_%d_TFLOPs' % m] = 5 * 2.0 * m**3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return out


def cpu(threads_list):
    import helpers
    from oracle import oracle as orc
    sim = helpers.uniform_plasma_sim(1024, 128, 2, (2, 4, 4), 'linear', seed=0)
    res = {}
    for nt in threads_list:
        o = orc.from_sim(sim, nthreads=nt)
        steps = 1 if nt == 1 else 3
        if nt > 1:
            o.step(1)
        t0 = time.perf_counter()
        o.step(steps)
        dt = time.perf_counter() - t0
        res['oracle_%d_threads_updates_per_s' % nt] = sum(s['x'].size for s in o.species) * steps / dt
    return res


if __name__ == '__main__':
    out = {}
    if '--no-gpu' not in sys.argv:
        out.update(gpu())
    if '--cpu' in sys.argv:
        from oracle import oracle as orc
        import bench
        out['available_cores'] = bench.available_cores()
        out.update(cpu(sorted(set([1, bench.available_cores(), 64]))))
    print(json.dumps(out))
