#!/bin/bash
# round 6, GPU run 51: stray scatter of the one-pass kernel with the operands of stray k + 1 read while stray k is written
# (-DFB_SCAT_PIPE, tools/variant.sh scatpipe cycle.hip) against the default, frozen C2 state, r = 1, 2, 3 passes after a sort
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run51; mkdir -p $O
for k in 1 2 3; do KNOCK_REPS=24 python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids | tee -a $O/knock.txt; done
