#!/bin/bash
# round 6, GPU run 45: the counter read-back of a one-pass iteration on a stream of its own (two counter sets), against the
# copy in the compute stream (FBPIC_AMD_STATS_STREAM=0): one-pass / cycle parity tests, C2 bench alternating, gaps of the step
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run45; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_cycle.py tests/test_gpu_lwfa.py -q -x -m gpu > $O/t.log 2>&1; echo "tests rc $?"; tail -2 $O/t.log
for e in 1 0 1 0 1 0; do
  FBPIC_AMD_STATS_STREAM=$e python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('stats_stream=$e', round(d['ms_per_step'],4), [round(x,4) for x in d['extra'].get('repeat_ms_per_step',[])])" | tee -a $O/bench_ab.txt
done
(cd /tmp && rocprofv3 --kernel-trace -d /root/repo/$O/tr -o r -- python /root/repo/bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-side-legs > /root/repo/$O/b.log 2>&1)
db=$(find $O/tr -name '*.db' | head -1); python tools/step_gaps.py $db 24 | head -14 | tee $O/single_gaps.txt; rm -rf $O/tr
FBPIC_AMD_STATS_STREAM=1 python bench.py --config C3 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3', round(d['ms_per_step'],4))" | tee -a $O/bench_ab.txt
