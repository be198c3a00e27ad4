#!/bin/bash
# round 6, GPU run 23: what bounds the Hankel GEMM after the straight-line loads - knock-outs (no requests / nor
# LDS writes / nor barriers in the K loop) and the SQ counters of the default
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run23; mkdir -p $O
for v in default hk_knock1 hk_knock2 hk_knock3; do
  for c in "plain 4416,256,8" "plain 2048,512,16" "dual 4416,256,4" "plain 1024,128,12"; do
    FBPIC_AMD_HK_VARIANT=$v timeout 200 python tools/hankel_tiles.py --one $c 2>&1 | grep -E "^(plain|dual)" >> $O/knock.txt
  done
done
cat $O/knock.txt
bash tools/sq_probe.sh hk_c3_full tools/hankel_probe.py --only 4416,256,8 > $O/sq_hk_c3.txt 2>&1
grep -A40 "k_hankel" $O/sq_hk_c3.txt | head -45
