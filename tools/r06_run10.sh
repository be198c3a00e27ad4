#!/bin/bash
# round 6, GPU run 10: in-wave regrouping of chunks full of strays (FBPIC_AMD_CYCLE_REGROUP = threshold)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run10; mkdir -p $O
FBPIC_AMD_CYCLE_REGROUP=3 timeout 900 python -m pytest tests/test_gpu_onepass.py -q --tb=short -k "not cubic and not -3]" > $O/t_onepass_regroup.log 2>&1; echo "onepass regroup rc $?" >> $O/summary.txt
FBPIC_AMD_CYCLE_REGROUP=3 timeout 600 python -m pytest tests/test_gpu_lwfa.py tests/test_gpu_cycle.py -q --tb=short > $O/t_lwfa_regroup.log 2>&1; echo "lwfa+cycle regroup rc $?" >> $O/summary.txt
for at in 64 20 12 8; do
  echo "== regroup at $at" >> $O/knock.txt
  KNOCK_REPS=8 FBPIC_AMD_CYCLE_REGROUP=$at timeout 300 python tools/cycle_knock.py >> $O/knock.txt 2>&1
done
for cfg in "64 " "8 bad=2.0,stray=2.0" "12 bad=2.0,stray=2.0" "8 bad=2.0,stray=2.0,period=6" "16 bad=2.0,stray=2.0" "8 "; do
  set -- $cfg
  echo "== C3 regroup at $1 policy '$2'" >> $O/c3.txt
  FBPIC_AMD_CYCLE_REGROUP=$1 timeout 400 python bench.py --config C3 --no-cpu-baseline --policy "$2" > $O/c3_$1_$(echo $2 | tr '=,.' '___').json 2>/dev/null
  python -c "
import json,glob; d=json.loads(open('$O/c3_$1_$(echo $2 | tr '=,.' '___').json').read().strip().split('\n')[-1]); print(d['ms_per_step'], {k:(round(v['mean_ms'],3),v['launches']) for k,v in d['kernels'].items() if v['mean_ms']*v['launches']>0.3})" >> $O/c3.txt 2>&1
done
tail -4 $O/t_onepass_regroup.log $O/t_lwfa_regroup.log; grep -v amdgpu.ids $O/knock.txt; cat $O/c3.txt; cat $O/summary.txt
