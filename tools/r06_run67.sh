#!/bin/bash
# round 6, GPU run 67: C5, graded ranges of the sorting deposition pass (default) against the plain cut, same box
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_v10; mkdir -p $O
line() { LBL="$1" python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print(os.environ['LBL'], round(d['ms_per_step'],4), [round(x,4) for x in d.get('extra',{}).get('repeat_ms_per_step',[])], {n: round(k[n]['mean_ms'],4) for n in ('fb_push_x_sort_deposit_J_rho','fb_gather_push_rank_next') if n in k})" | tee -a $O/scan_ab.txt; }
for t in graded plain graded; do
  if [ $t = plain ]; then export FBPIC_AMD_CYCLE_TAIL=0; else unset FBPIC_AMD_CYCLE_TAIL; fi
  python bench.py --config C5 --no-cpu-baseline --no-side-legs 2>/dev/null | line "C5 $t"
done
