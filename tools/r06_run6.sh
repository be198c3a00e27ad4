#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run6; mkdir -p $O
PYTHONFAULTHANDLER=1 timeout 900 python -m pytest tests/test_gpu_onepass.py -q --tb=line -p no:cacheprovider -k "test_one_pass_equals_the_four_entry_points and (003-3 or 63-3 or 65-3 or 1-3)" > $O/t_cubic.log 2>&1; echo "cubic rc $?" >> $O/summary.txt
PYTHONFAULTHANDLER=1 timeout 900 python -m pytest tests/test_gpu_onepass.py -q --tb=line -p no:cacheprovider -k "test_step_one_pass" > $O/t_step.log 2>&1; echo "step rc $?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q -k "deferral" > $O/t_deferral.log 2>&1; echo "deferral rc $?" >> $O/summary.txt
head -60 $O/t_cubic.log; tail -30 $O/t_step.log; tail -8 $O/t_deferral.log; cat $O/summary.txt
