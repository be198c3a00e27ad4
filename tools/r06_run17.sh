#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run17; mkdir -p $O
FBPIC_AMD_CYCLE_PAIRS=1 FBPIC_AMD_CYCLE_REGROUP=2 timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_lwfa.py tests/test_gpu_cycle.py -q --tb=short -k "not regrouped_in_a_process" > $O/t_pairs.log 2>&1; echo "pairs tests rc $?" >> $O/summary.txt
for pr in 0 1 0 1; do FBPIC_AMD_CYCLE_PAIRS=$pr timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/c3_$pr.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c3_$pr.json').read().strip().split('\n')[-1]); print('C3 pairs $pr', d['ms_per_step'], round(d['kernels']['fb_gather_push_deposit_J_rho']['mean_ms'],4))"; done
for pr in 0 1; do FBPIC_AMD_CYCLE_PAIRS=$pr timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs > $O/c2_$pr.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c2_$pr.json').read().strip().split('\n')[-1]); print('C2 pairs $pr', d['ms_per_step'], d['extra']['repeat_ms_per_step'], round(d['kernels']['fb_gather_push_deposit_J_rho']['mean_ms'],4))"; done
grep -E "passed|failed|^FAILED" $O/t_pairs.log | tail -5; cat $O/summary.txt
