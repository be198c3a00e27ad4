#!/bin/bash
# round 6, GPU run 63: graded ranges at C3 - long ranges (FBPIC_AMD_CYCLE_CPW) x tail (FBPIC_AMD_CYCLE_TAIL); default now "8,2"
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run63; mkdir -p $O
for c in "0 8,2" "6 8,2" "4 8,2" "0 4,2" "6 4,2" "0 4,4" "0 0" "0 8,2"; do set -- $c
  FBPIC_AMD_CYCLE_CPW=$1 FBPIC_AMD_CYCLE_TAIL=$2 python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 cpw=$1 (0 = 10) tail=$2', round(d['ms_per_step'],4), d['kernels'].get('fb_gather_push_deposit_J_rho',{}).get('mean_ms'))" | tee -a $O/scan.txt
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 20/5 default', round(d['ms_per_step'],4), [round(x,4) for x in d['extra']['repeat_ms_per_step']], d['kernels']['fb_gather_push_deposit_J_rho']['mean_ms'])" | tee -a $O/scan.txt
