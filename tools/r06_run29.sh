#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run29; mkdir -p $O
timeout 300 python tools/early_drift.py 2>&1 | grep -v amdgpu.ids | tee $O/early_drift.txt
