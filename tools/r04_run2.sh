#!/bin/bash
mkdir -p gpurun_out/r04_2
cd $GRAFT_REPO_ROOT
for args in "100003 0.4 1 1 1 1.03" "99968 0.4 1 1 1 1.03" "100003 0.4 0 1 1 1.03" "100003 0.4 1 0 1 1.03" "100003 0.001 1 1 1 1.03" "100003 0.4 1 1 0 1.03" "100003 0.4 1 1 1 0.9" "4096 0.4 1 1 1 1.03"; do
  echo "=== $args" >> gpurun_out/r04_2/debug.txt
  timeout 120 python tools/onepass_debug.py $args 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | head -12 >> gpurun_out/r04_2/debug.txt
done
cat gpurun_out/r04_2/debug.txt
