#!/bin/bash
mkdir -p gpurun_out/r04_3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_3
timeout 600 python -m pytest tests/test_gpu_onepass.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -40 > $O/onepass_tests.txt
timeout 300 python tools/onepass_probe.py --periods 0,3,5,8 > $O/probe_v1.txt 2>&1
FBPIC_AMD_CYCLE_WPE=4 timeout 300 python tools/onepass_probe.py --periods 3,5,8 > $O/probe_v1_w4.txt 2>&1
FBPIC_AMD_LIB=$PWD/fbpic_amd/csrc/variants/libfbpic_amd_v0.so timeout 300 python tools/onepass_probe.py --periods 3,5 > $O/probe_v0.txt 2>&1
tail -15 $O/onepass_tests.txt; grep period $O/probe_*.txt
