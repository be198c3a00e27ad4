#!/bin/bash
# round 6, GPU run 19: register stages (chunks of global traffic in flight) of the Hankel GEMM
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run19; mkdir -p $O
timeout 1500 python tools/hankel_tiles.py 2>&1 | grep -v amdgpu.ids | tee $O/tiles.txt
