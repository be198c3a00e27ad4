#!/bin/bash
# round 6, GPU run 35: chunks per wave of the one-pass kernel in the LATTICE phase of the C2 workload (the driver's
# steps 6 - 25): waves of identical work run in discrete rounds - 16 384 waves on 3 072 slots are 5.33 rounds, i.e. 6
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run35; mkdir -p $O
for cpw in 4 2 1 3 4 2 1 6; do
  FBPIC_AMD_CYCLE_CPW=$cpw timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cpw $cpw', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], 'one-pass(late)', d['kernels']['fb_gather_push_deposit_J_rho']['mean_ms'])" | tee -a $O/cpw.txt
done
