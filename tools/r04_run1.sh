#!/bin/bash
# round 4, first GPU run: new one-pass tests, whole suite, probes of the sort period / occupancy variants
mkdir -p gpurun_out/r04_1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_onepass.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r04_1/onepass_tests.txt
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > gpurun_out/r04_1/suite.txt
timeout 400 python tools/onepass_probe.py --periods 0,1,4,8,16,32 > gpurun_out/r04_1/probe_wpe0.txt 2>&1
FBPIC_AMD_CYCLE_WPE=4 timeout 300 python tools/onepass_probe.py --periods 4,8,16 > gpurun_out/r04_1/probe_wpe4.txt 2>&1
tail -5 gpurun_out/r04_1/onepass_tests.txt; tail -8 gpurun_out/r04_1/suite.txt; cat gpurun_out/r04_1/probe_wpe0.txt | grep period
