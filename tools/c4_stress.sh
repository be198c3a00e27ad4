#!/bin/bash
# repeated runs of the 8-slab C4 test (both forms) to catch the rank loss with its output
# (gpurun_out/c4_timing/: per-rank stack dumps and stderr, appended across attempts; loss_report_w8.txt)
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/c4_timing/*
N=${1:-5}
for i in $(seq 1 $N); do
  timeout 600 python -m pytest "tests/test_gpu_c4.py::test_c4_lwfa_4096x256_on_8_slabs_reproduces_the_single_domain" -x -q 2>&1 | grep -E "passed|failed|warning" | tail -2
  ls gpurun_out/c4_timing/ | grep -c lost_ | sed "s/^/  lost files so far: /"
done
for f in gpurun_out/c4_timing/w8_r*_stderr.log; do echo "== $f"; grep -v "^----\|amdgpu.ids" $f | tail -8; done | head -80
for f in gpurun_out/c4_timing/w8_r*_stack.log; do n=$(grep -vc "^----" $f); if [ "$n" != "0" ]; then echo "== $f"; tail -40 $f; fi; done | head -150
tail -3 gpurun_out/c4_timing/loss_report_w8.txt 2>/dev/null
