#!/bin/bash
# round 6, GPU run 52: compiler-option variants of cycle.hip (tools/variant.sh f<k> cycle.hip "<option>") on the frozen C2 state
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run52; mkdir -p $O
for k in 1 2; do KNOCK_REPS=24 python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids | tee -a $O/knock.txt; done
