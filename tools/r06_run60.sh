#!/bin/bash
# round 6, GPU run 60: final state after the range caps - full GPU suite, smoke, C5 bench line, default C2 line
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_v8; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "gpu suite rc $? : $(grep -E 'passed|failed' $O/t_all.log | tail -1)" | tee $O/summary.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/summary.txt
timeout 600 python bench.py --config C5 --no-cpu-baseline --no-side-legs > $O/bench_c5.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
for f in bench bench_c5; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]); r=d['roofline']; print('$f', d['value'], round(d['ms_per_step'],4), d.get('extra',{}).get('repeat_ms_per_step'), r['kernel'], round(r['frac'],3), r.get('frac_of_dp_floor'), (r.get('gather_push') or {}).get('frac_of_dp_floor'))" | tee -a $O/summary.txt; done
