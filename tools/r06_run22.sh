#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run22; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_c4_golden.py > $O/t_all.log 2>&1; echo "gpu suite rc $?" > $O/summary.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
grep -E "passed|failed|^FAILED" $O/t_all.log | tail -8; cat $O/summary.txt
