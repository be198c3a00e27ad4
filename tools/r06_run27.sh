#!/bin/bash
# round 6, GPU run 27: why is the first timed call of the C2 line 5 % slower than its repeats? warm-up length scan
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run27; mkdir -p $O
for w in 5 25 60 5 25 60; do
  timeout 300 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-side-legs --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('warmup $w', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], d['extra']['particle_passes'], d['extra'].get('clocks_before'), d['extra'].get('clocks_after_headline_call'))" | tee -a $O/warmup.txt
done
