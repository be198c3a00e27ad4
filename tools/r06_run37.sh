#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run37; mkdir -p $O
timeout 400 python tools/exchange_cost.py 2>&1 | grep -v amdgpu.ids | head -75 | tee $O/exchange_cost.txt
