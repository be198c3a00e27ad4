#!/bin/bash
# round 6, GPU run 49: regroup threshold (FBPIC_AMD_CYCLE_REGROUP) and sort period (FBPIC_AMD_SORT_PERIOD) scans on the
# final kernels: C2 steady state (--steps 40 --warmup 20), C2 driver command (--steps 20 --warmup 5), C3
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run49; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(d['ms_per_step'],4), [round(x,4) for x in d.get('extra',{}).get('repeat_ms_per_step',[])], d.get('extra',{}).get('particle_passes'))"; }
for at in 12 4 6 8 3 12; do
  FBPIC_AMD_CYCLE_REGROUP=$at python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | line "C2 40/20 regroup_at=$at" | tee -a $O/scan.txt
done
for at in 12 6 4; do
  FBPIC_AMD_CYCLE_REGROUP=$at python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | line "C2 20/5 regroup_at=$at" | tee -a $O/scan.txt
done
for p in 3 2 4 5 3; do
  FBPIC_AMD_SORT_PERIOD=$p python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | line "C2 40/20 sort_period=$p" | tee -a $O/scan.txt
  FBPIC_AMD_SORT_PERIOD=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | line "C2 20/5 sort_period=$p" | tee -a $O/scan.txt
done
for at in 12 8 16 24; do
  FBPIC_AMD_CYCLE_REGROUP=$at python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | line "C3 regroup_at=$at" | tee -a $O/scan.txt
done
for p in 3 2 4 6; do
  FBPIC_AMD_SORT_PERIOD=$p python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | line "C3 sort_period=$p" | tee -a $O/scan.txt
done
