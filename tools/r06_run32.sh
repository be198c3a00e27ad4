#!/bin/bash
# round 6, GPU run 32: is the slow lattice phase of the C2 workload the flush atomics (waves in lock-step on a regular
# lattice hit the same records at the same moment)?  frozen-state launch time with / without them at three ages
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run32; mkdir -p $O
for age in 4 12 24 48 96; do
  echo "== age $age" >> $O/knock.txt
  KNOCK_AGE=$age KNOCK_REPS=8 timeout 300 python tools/cycle_knock.py 2>&1 | grep -E "default|noflush" >> $O/knock.txt
done
cat $O/knock.txt
