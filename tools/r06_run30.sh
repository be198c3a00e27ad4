#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run30; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_c4_golden.py -q -x --durations=3 > $O/t_c4g.log 2>&1; echo "c4 golden rc $?" > $O/summary.txt
tail -60 $O/t_c4g.log; cat $O/summary.txt
