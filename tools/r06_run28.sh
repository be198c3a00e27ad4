#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run28; mkdir -p $O
timeout 300 python tools/hosttime.py 2>&1 | grep -v amdgpu.ids | head -45 | tee $O/hosttime.txt
