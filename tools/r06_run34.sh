#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run34; mkdir -p $O
timeout 900 python tools/hankel_tiles.py --cases "plain:4416,256,8:0,9,11,12,4;plain:4416,256,12:0,9,11,12,4;plain:2048,512,16:0,9,11,12,4,5;plain:2048,512,24:0,9,4,5;plain:1152,128,12:0,9,11,12,6" 2>&1 | grep -v amdgpu.ids | tee $O/tiles.txt
