#!/bin/bash
# the 8-process C4 test over and over: does the loss of rank processes show up, and with what evidence?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4_timing
echo "memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null) cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) pids.max $(cat /sys/fs/cgroup/pids.max 2>/dev/null)" > gpurun_out/c4_timing/box.txt
free -g >> gpurun_out/c4_timing/box.txt
N=${1:-10}
for k in $(seq 1 $N); do
  timeout 400 python -m pytest tests/test_gpu_c4.py -q -m gpu -k "c4_lwfa" -W always 2>&1 | grep -E "passed|failed|disappeared|Warning" | head -5
  echo "run $k: memory.peak $(cat /sys/fs/cgroup/memory.peak 2>/dev/null) events $(cat /sys/fs/cgroup/memory.events 2>/dev/null | tr '\n' ' ')"
done
ls gpurun_out/c4_timing | grep -c loss
