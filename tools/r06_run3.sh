#!/bin/bash
# round 6, GPU run 3: why are persistent waves slower?  (a) the old cut through the new code path
# (16384 waves, one static range of 4 chunks each), (b) 2x / 1.5x over-subscription, (c) staggered start
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run3; mkdir -p $O
for cfg in "4 64 100 0" "4 32 100 0" "4 24 100 0" "4 16 100 0" "4 12 100 0" "4 12 100 1" "4 12 100 3" "4 12 75 2" "2 24 75 0" "4 64 0 0"; do
  set -- $cfg
  echo "== range $1 waves/CU $2 static $3 % stagger $4" >> $O/knock.txt
  KNOCK_R=1,3 KNOCK_REPS=8 FBPIC_AMD_CYCLE_RANGE=$1 FBPIC_AMD_CYCLE_WAVES_PER_CU=$2 FBPIC_AMD_CYCLE_STATIC=$3 FBPIC_AMD_CYCLE_STAGGER=$4 timeout 300 python tools/cycle_knock.py >> $O/knock.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -s > $O/t_fullsize.log 2>&1; echo "fullsize rc $?" >> $O/summary.txt
grep -v amdgpu.ids $O/knock.txt; tail -15 $O/t_fullsize.log; cat $O/summary.txt
