#!/bin/bash
mkdir -p gpurun_out/r04_8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_8
timeout 600 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_multirank_golden.py -q -m gpu -k "onepass or one_pass or rank_next_home or 8_ranks" 2>&1 | grep -v "^  File\|^Extension" > $O/tests.txt
grep -n "passed\|failed" $O/tests.txt | tail -2; grep -B2 -A40 "^___" $O/tests.txt | head -80
for P in 2 3 4; do timeout 200 python tools/onepass_probe.py --periods $P > $O/probe_$P.txt 2>&1; grep -A6 period $O/probe_$P.txt; done
