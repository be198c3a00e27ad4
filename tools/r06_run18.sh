#!/bin/bash
# round 6, GPU run 18: workgroup shapes of the Hankel GEMM at the C3 / C4 / C5 sizes, SQ counters of the default
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run18; mkdir -p $O
timeout 1500 python tools/hankel_tiles.py 2>&1 | grep -v amdgpu.ids | tee $O/tiles.txt
bash tools/sq_probe.sh hk_c3 tools/hankel_probe.py --only 4416,256,8 > $O/sq_hk_c3.txt 2>&1
grep -A40 "k_hankel" $O/sq_hk_c3.txt | head -60
