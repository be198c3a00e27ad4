#!/bin/bash
mkdir -p gpurun_out/r04_flaky
cd $GRAFT_REPO_ROOT
for k in $(seq 1 12); do
  timeout 300 python -m pytest tests/test_gpu_onepass.py -q -m gpu -x -k "four_entry or without_wrap" > /tmp/run_$k.txt 2>&1
  if ! grep -q " passed" /tmp/run_$k.txt || grep -q "failed" /tmp/run_$k.txt; then
    grep -v "^  File\|^Extension" /tmp/run_$k.txt | head -150 > gpurun_out/r04_flaky/fail_$k.txt
    echo "run $k FAILED"
  else echo "run $k ok"; fi
done
