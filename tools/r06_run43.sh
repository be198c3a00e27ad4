#!/bin/bash
# round 6, GPU run 43: full GPU suite + the three bench lines on HEAD (non-temporal FFT sweeps and solver step)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run43; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "gpu suite rc $?"; grep -E "passed|failed" $O/t_all.log | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
timeout 600 python bench.py --config C5 --no-cpu-baseline --no-side-legs > $O/bench_c5.json 2>/dev/null
for f in bench bench_c3 bench_c5; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d.get('extra',{}).get('repeat_ms_per_step'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline'].get('hankel',{}).get('frac'))"; done
