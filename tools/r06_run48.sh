#!/bin/bash
# round 6, GPU run 48: the full GPU suite of the r06_v7 state again (its first execution failed the new two-rank test's own
# too strong assertion - the ORDER of handed-over particles is that of the selection's atomics -; nothing else changed)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_v7; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "gpu suite rc $? (second execution, tools/r06_run48.sh)" > $O/summary_suite.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
tail -3 $O/t_all.log; cat $O/summary_suite.txt
