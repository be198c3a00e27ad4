#!/bin/bash
# round 6, GPU run 59: ranges of at most 8 (cubic gather + push + rank) / 16 (sorting deposition pass) chunks per wave as the default,
# against the old cap of 64 (FBPIC_AMD_CPW_CAP=64): C5 alternating, C2 unchanged?, C3 with shorter ranges of the one-pass kernel;
# tests that run the kernels at the big sizes
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run59; mkdir -p $O
c5() { python bench.py --config C5 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C5 $1', round(d['ms_per_step'],3), [round(x,3) for x in d['extra']['repeat_ms_per_step']], k['fb_gather_push_rank_next']['mean_ms'], k['fb_push_x_sort_deposit_J_rho']['mean_ms'])" | tee -a $O/scan.txt; }
c5 "caps 8/16 (default)"; FBPIC_AMD_CPW_CAP=64 c5 "cap 64 (before)"; c5 "caps 8/16 (default)"; FBPIC_AMD_CPW_CAP=64 c5 "cap 64 (before)"
FBPIC_AMD_CPW_CAP=4 c5 "cap 4"
python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2', round(d['ms_per_step'],4), [round(x,4) for x in d['extra']['repeat_ms_per_step']])" | tee -a $O/scan.txt
for w in 0 6 4 3 0; do
  FBPIC_AMD_CYCLE_CPW=$w python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 cycle_cpw=$w (0 = default 10)', round(d['ms_per_step'],4))" | tee -a $O/scan.txt
done
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_kernels.py -q -m gpu > $O/t.log 2>&1; echo "tests rc $? $(grep -E 'passed|failed' $O/t.log | tail -1)" | tee -a $O/scan.txt
