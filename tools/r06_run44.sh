#!/bin/bash
# round 6, GPU run 44: the first half of a particle hand-over posted behind the particle pass of the iteration before
# (Simulation.early_handover): multi-rank parity tests, decomposed loopback step with / without it, trace of the step
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run44; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multirank_golden.py tests/test_gpu_c4.py tests/test_gpu_c4_golden.py tests/test_gpu_first_exchange.py tests/test_gpu_diagnostics.py -q -x > $O/t.log 2>&1; echo "tests rc $?"; tail -2 $O/t.log
for e in 1 0 1 0; do
  FBPIC_AMD_EARLY_HANDOVER=$e python tools/loopback_multirank.py --steps 56 --warmup 16 2>&1 | grep -v amdgpu.ids | sed "s/^/early=$e /" | tee -a $O/loopback_ab.txt
done
python tools/loopback_multirank.py --single --steps 56 --warmup 16 2>&1 | grep -v amdgpu.ids | tee -a $O/loopback_ab.txt
for e in 1 0; do
(cd /tmp && FBPIC_AMD_EARLY_HANDOVER=$e rocprofv3 --kernel-trace -d /root/repo/$O/tr$e -o r -- python /root/repo/tools/loopback_multirank.py --steps 28 --warmup 2 > /root/repo/$O/l$e.log 2>&1)
db=$(find $O/tr$e -name '*.db' | head -1); echo "early=$e" | tee -a $O/loop_gaps.txt; python tools/step_gaps.py $db 14 | head -12 | tee -a $O/loop_gaps.txt; rm -rf $O/tr$e
done
