#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run31; mkdir -p $O
timeout 900 python tools/c4_golden_timing.py 3 2>&1 | grep -v amdgpu.ids | tee $O/timing.txt | head -90
