#!/bin/bash
# round 6, GPU run 42: span against busy time of the C2 step (single domain) and of the decomposed loopback step
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run42; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace -d /root/repo/$O/tr -o r -- python /root/repo/bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-side-legs > /root/repo/$O/b.log 2>&1)
db=$(find $O/tr -name '*.db' | head -1); python tools/step_gaps.py $db 24 | tee $O/single_gaps.txt; rm -rf $O/tr
(cd /tmp && rocprofv3 --kernel-trace -d /root/repo/$O/tr2 -o r -- python /root/repo/tools/loopback_multirank.py --steps 28 --warmup 2 > /root/repo/$O/l.log 2>&1)
db=$(find $O/tr2 -name '*.db' | head -1); python tools/step_gaps.py $db 10 | tee $O/loop_gaps.txt; rm -rf $O/tr2
tail -3 $O/b.log | cut -c1-300; tail -4 $O/l.log
