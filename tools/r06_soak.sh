#!/bin/bash
# two more executions of the full GPU suite + smoke (flakiness check of the round's new paths)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_soak; mkdir -p $O
for i in 1 2; do
  timeout 1500 python -m pytest tests -q -m gpu > $O/t_all_$i.log 2>&1; echo "run $i rc $?" >> $O/summary.txt
  cp gpurun_out/achieved_errors.json $O/achieved_errors_$i.json 2>/dev/null
  grep -E "passed|failed" $O/t_all_$i.log | tail -1 >> $O/summary.txt
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/summary.txt
cat $O/summary.txt
