#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py -q -s -k "c3_full_grid" > $O/t_c3.log 2>&1; echo "c3 rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_onepass.py -q --tb=short > $O/t_onepass.log 2>&1; echo "onepass rc $?" >> $O/summary.txt
grep -E "c3_full_grid|passed|failed|Error" $O/t_c3.log | tail -40; tail -5 $O/t_onepass.log; cat $O/summary.txt
