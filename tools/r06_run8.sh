#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_onepass.py -q --tb=line -k "test_one_pass_equals_the_four_entry_points" > $O/t_cubic.log 2>&1; echo "rc $?" >> $O/summary.txt
grep -E "E, B .* vs oracle|^/root|passed|failed" $O/t_cubic.log | head -80
