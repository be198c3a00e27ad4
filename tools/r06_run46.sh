#!/bin/bash
# round 6, GPU run 46: counter read-back of the one-pass iterations 'adaptive' (first pass of an order always, later ones only
# while the limits are within reach) against 'always' (FBPIC_AMD_CYCLE_MEASURE): parity tests, C2 / C3 bench alternating
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run46; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_cycle.py tests/test_gpu_lwfa.py tests/test_gpu_configs.py -q -x -m gpu > $O/t.log 2>&1; echo "tests rc $?"; tail -2 $O/t.log
for e in adaptive always adaptive always adaptive always; do
  FBPIC_AMD_CYCLE_MEASURE=$e python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 $e', round(d['ms_per_step'],4), [round(x,4) for x in d['extra'].get('repeat_ms_per_step',[])], d['extra']['particle_passes'])" | tee -a $O/bench_ab.txt
done
for e in adaptive always; do
  FBPIC_AMD_CYCLE_MEASURE=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 driver command $e', round(d['ms_per_step'],4), [round(x,4) for x in d['extra'].get('repeat_ms_per_step',[])])" | tee -a $O/bench_ab.txt
  FBPIC_AMD_CYCLE_MEASURE=$e python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 $e', round(d['ms_per_step'],4), d['extra'].get('particle_passes'))" | tee -a $O/bench_ab.txt
  FBPIC_AMD_CYCLE_MEASURE=$e python tools/loopback_multirank.py --steps 56 --warmup 16 2>&1 | grep -v amdgpu.ids | sed "s/^/$e /" | tee -a $O/bench_ab.txt
done
