#!/bin/bash
# round 6, GPU run 2: persistent waves with a static share (no atomics there) + queue counters on lines
# of their own; A/B against round 5's kernel on a frozen state; parity growth at C2
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_onepass.py -x -q > $O/t_onepass.log 2>&1; echo "onepass rc $?" >> $O/summary.txt
for cfg in "4 12 100" "4 12 75" "4 12 50" "4 12 0" "2 12 75" "8 12 75" "4 12 90" "4 11 75"; do
  set -- $cfg
  echo "== range $1 waves/CU $2 static $3 %" >> $O/knock.txt
  KNOCK_REPS=8 FBPIC_AMD_CYCLE_RANGE=$1 FBPIC_AMD_CYCLE_WAVES_PER_CU=$2 FBPIC_AMD_CYCLE_STATIC=$3 timeout 300 python tools/cycle_knock.py >> $O/knock.txt 2>&1
done
timeout 600 python tools/c2_parity_growth.py 6 > $O/growth.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
grep -v amdgpu.ids $O/knock.txt; cat $O/growth.txt | grep -v amdgpu.ids; tail -3 $O/t_onepass.log; cat $O/summary.txt; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['extra'].get('repeat_ms_per_step'), {k:v for k,v in d['extra'].items() if 'ms' in k}); print(d['roofline']); print(d.get('cpu_baseline'))"
