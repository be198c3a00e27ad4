#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run11; mkdir -p $O
for at in 64 24 16 12 8; do
  echo "== regroup at $at" >> $O/knock.txt
  KNOCK_REPS=8 FBPIC_AMD_CYCLE_REGROUP=$at timeout 300 python tools/cycle_knock.py >> $O/knock.txt 2>&1
done
for at in 64 16 12; do
  echo "== C2 bench regroup at $at" >> $O/c2.txt
  FBPIC_AMD_CYCLE_REGROUP=$at timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs > $O/c2_$at.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/c2_$at.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['extra']['repeat_ms_per_step'], round(d['kernels']['fb_gather_push_deposit_J_rho']['mean_ms'],4))" >> $O/c2.txt 2>&1
done
grep -v amdgpu.ids $O/knock.txt; cat $O/c2.txt
