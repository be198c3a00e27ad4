#!/bin/bash
# round 6, GPU run 65: graded ranges also in the ranking form of the one-pass kernel and in the sorting deposition passes
# (shared helper graded_ranges / wave_range, fb_common.h): full GPU suite, then default against FBPIC_AMD_CYCLE_TAIL=0 (plain cut)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_v10; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "gpu suite rc $? : $(grep -E 'passed|failed' $O/t_all.log | tail -1)" | tee $O/summary.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('$1', round(d['ms_per_step'],4), [round(x,4) for x in d.get('extra',{}).get('repeat_ms_per_step',[])], {n: round(k[n]['mean_ms'],4) for n in ('fb_gather_push_deposit_J_rho','fb_gather_push_rank_next_home','fb_push_x_sort_deposit_J_rho','fb_gather_push_rank_next') if n in k})" | tee -a $O/scan.txt; }
for t in "" 0 "" 0; do FBPIC_AMD_CYCLE_TAIL=$t python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-side-legs 2>/dev/null | line "C2 40/20 tail='$t'"; done
for t in "" 0 "" 0; do FBPIC_AMD_CYCLE_TAIL=$t python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | line "C3 tail='$t'"; done
for t in "" 0; do FBPIC_AMD_CYCLE_TAIL=$t python bench.py --config C5 --no-cpu-baseline --no-side-legs 2>/dev/null | line "C5 tail='$t'"; done
