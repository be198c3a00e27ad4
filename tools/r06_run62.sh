#!/bin/bash
# round 6, GPU run 62: graded ranges of the one-pass kernel - the waves that start last walk short ranges (FBPIC_AMD_CYCLE_TAIL =
# "<1 / share of an XCD's chunks>,<chunks per short wave>", 0 = off = every wave 4 chunks): parity tests, C2 / C3 bench alternating
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run62; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_cycle.py tests/test_gpu_fullsize_oracle.py -q -x -m gpu > $O/t.log 2>&1; echo "tests rc $? $(grep -E 'passed|failed' $O/t.log | tail -1)" | tee -a $O/scan.txt
c2() { python bench.py --steps $2 --warmup $3 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 $2/$3 tail=$1', round(d['ms_per_step'],4), [round(x,4) for x in d['extra']['repeat_ms_per_step']], d['kernels']['fb_gather_push_deposit_J_rho']['mean_ms'])" | tee -a $O/scan.txt; }
for t in 0 8,1 8,2 4,1 4,2 16,1 0 8,1; do FBPIC_AMD_CYCLE_TAIL=$t c2 $t 40 20; done
for t in 0 8,1 4,2 0 8,1; do FBPIC_AMD_CYCLE_TAIL=$t c2 $t 20 5; done
for t in 0 8,1 8,2 4,2 0; do
  FBPIC_AMD_CYCLE_TAIL=$t python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 tail=$t', round(d['ms_per_step'],4), d['kernels'].get('fb_gather_push_deposit_J_rho',{}).get('mean_ms'))" | tee -a $O/scan.txt
done
