#!/bin/bash
# round 6, GPU run 50: soak of the final code state - the full GPU suite twice more
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run50; mkdir -p $O
for k in 1 2; do
  timeout 1500 python -m pytest tests -q -m gpu > $O/t_all_$k.log 2>&1; echo "suite $k rc $? : $(grep -E 'passed|failed' $O/t_all_$k.log | tail -1)" | tee -a $O/summary.txt
  cp gpurun_out/achieved_errors.json $O/achieved_errors_$k.json 2>/dev/null
done
