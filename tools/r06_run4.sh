#!/bin/bash
# round 6, GPU run 4: persistent waves, everything dealt out statically (no atomics at all)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run4; mkdir -p $O
for cfg in "4 12 100 0" "2 12 100 0" "8 12 100 0" "1 12 100 0" "4 11 100 0" "4 13 100 0" "3 12 100 0" "4 12 100 2" "6 12 100 0"; do
  set -- $cfg
  echo "== range $1 waves/CU $2 static $3 % stagger $4" >> $O/knock.txt
  KNOCK_R=1,3 KNOCK_REPS=8 FBPIC_AMD_CYCLE_RANGE=$1 FBPIC_AMD_CYCLE_WAVES_PER_CU=$2 FBPIC_AMD_CYCLE_STATIC=$3 FBPIC_AMD_CYCLE_STAGGER=$4 timeout 300 python tools/cycle_knock.py >> $O/knock.txt 2>&1
done
FBPIC_AMD_CYCLE_STATIC=100 timeout 900 python -m pytest tests/test_gpu_onepass.py -x -q > $O/t_onepass.log 2>&1; echo "onepass rc $?" >> $O/summary.txt
FBPIC_AMD_CYCLE_STATIC=100 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
grep -v amdgpu.ids $O/knock.txt; tail -3 $O/t_onepass.log; cat $O/summary.txt; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['extra'].get('repeat_ms_per_step')); print({k:(v['mean_ms'],v['launches']) for k,v in d['kernels'].items()})"
