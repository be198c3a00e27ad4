#!/bin/bash
mkdir -p gpurun_out/r04_6
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_6
timeout 600 python -m pytest tests/test_gpu_spectral_cycle.py tests/test_gpu_onepass.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" > $O/new_tests.txt
grep -n "passed\|failed" $O/new_tests.txt | tail -2
grep -B2 -A45 "^___" $O/new_tests.txt | head -120
timeout 200 python tools/onepass_probe.py --periods 3 > $O/probe_fused.txt 2>&1
FBPIC_AMD_FUSE_SPECT=0 timeout 200 python tools/onepass_probe.py --periods 3 > $O/probe_unfused.txt 2>&1
grep -A8 period $O/probe_fused.txt $O/probe_unfused.txt
