#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run36; mkdir -p $O
timeout 300 python tools/early_steps.py 2>&1 | grep -v amdgpu.ids | tee $O/early_steps.txt
