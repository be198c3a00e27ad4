#!/bin/bash
mkdir -p gpurun_out/r04_5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_5
timeout 600 python -m pytest tests/test_gpu_onepass.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -8 > $O/onepass_tests.txt
for P in 1 0; do for W in 0 4; do
FBPIC_AMD_CYCLE_PIPE=$P FBPIC_AMD_CYCLE_WPE=$W timeout 300 python tools/onepass_probe.py --periods 2,3,4 > $O/probe_p${P}_w${W}.txt 2>&1
done; done
tail -3 $O/onepass_tests.txt; grep -A1 period $O/probe_*.txt | grep -v "^--"
