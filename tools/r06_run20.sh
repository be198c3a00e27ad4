#!/bin/bash
# round 6, GPU run 20: straight-line loads of the Hankel GEMM (FULL) against the branchy ones, tests
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_spectral_cycle.py -q -k "hankel or transformer or rt_pm or spectral" > $O/t_hankel.log 2>&1; echo "hankel tests rc $?" > $O/summary.txt
C="plain:4416,256,8:0,4,1;plain:4416,256,12:0,4;plain:2048,512,16:0,4,1;plain:1152,128,12:0,4,5;plain:1024,128,12:0,4,5;dual:4416,256,4:0,1,2;dual:2048,512,8:0,1,2;dual:1152,128,4:0,1,2;dual:1024,128,4:0,1,2"
echo "== straight-line loads" > $O/tiles.txt
timeout 900 python tools/hankel_tiles.py --cases "$C" 2>&1 | grep -v amdgpu.ids >> $O/tiles.txt
echo "== branchy loads (FBPIC_AMD_HANKEL_NOFULL=1)" >> $O/tiles.txt
FBPIC_AMD_HANKEL_NOFULL=1 timeout 900 python tools/hankel_tiles.py --cases "$C" 2>&1 | grep -v amdgpu.ids >> $O/tiles.txt
cat $O/tiles.txt; tail -3 $O/t_hankel.log; cat $O/summary.txt
