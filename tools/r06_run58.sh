#!/bin/bash
# round 6, GPU run 58: range length per wave of the particle kernels at C5 (FBPIC_AMD_WAVE_FACTOR k: k x 16384 waves, i.e. 64 / k chunks
# per wave) and at C3
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run58; mkdir -p $O
for k in 1 2 4 8 1 4; do
  FBPIC_AMD_WAVE_FACTOR=$k python bench.py --config C5 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C5 wave_factor=$k', round(d['ms_per_step'],3), [round(x,3) for x in d['extra']['repeat_ms_per_step']], k['fb_gather_push_rank_next']['mean_ms'], k['fb_push_x_sort_deposit_J_rho']['mean_ms'])" | tee -a $O/scan.txt
done
for k in 1 2 4 1; do
  FBPIC_AMD_WAVE_FACTOR=$k python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 wave_factor=$k', round(d['ms_per_step'],4))" | tee -a $O/scan.txt
done
