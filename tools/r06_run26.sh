#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run26; mkdir -p $O
timeout 600 python tools/zfft_cold.py 2>&1 | grep -v amdgpu.ids | tee $O/zfft_cold.txt
