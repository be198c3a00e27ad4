#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_onepass.py -q --tb=short -k "regrouped" > $O/t_regroup.log 2>&1; echo "regroup subprocess test rc $?" >> $O/summary.txt
for per in 3 4 5 6; do
  timeout 400 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-side-legs --policy "period=$per" > $O/c2_p$per.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/c2_p$per.json').read().strip().split('\n')[-1]); print('period $per', d['ms_per_step'], d['extra']['repeat_ms_per_step'], d['extra']['particle_passes'])" >> $O/c2.txt 2>&1
done
tail -3 $O/t_regroup.log; cat $O/c2.txt; cat $O/summary.txt
