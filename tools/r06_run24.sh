#!/bin/bash
# round 6, GPU run 24: two-engine sorting pass with the depth-2 pipeline (one vector-memory wait per chunk)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run24; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_onepass.py tests/test_gpu_cycle.py -q -x > $O/t_sel.log 2>&1; echo "selected rc $?" > $O/summary.txt
timeout 600 python bench.py --config C5 --no-cpu-baseline --no-side-legs > $O/c5.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c5.json').read().strip().split('\n')[-1]); print('C5', d['ms_per_step'], {k:(round(v['mean_ms'],3),v['launches']) for k,v in d['kernels'].items() if v['mean_ms']*v['launches']>1})"
timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/c3.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c3.json').read().strip().split('\n')[-1]); print('C3', d['ms_per_step'], {k:(round(v['mean_ms'],3),v['launches']) for k,v in d['kernels'].items() if v['mean_ms']*v['launches']>0.3})"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs > $O/c2.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().split('\n')[-1]); print('C2', d['ms_per_step'], d['extra']['repeat_ms_per_step'], {k:(round(v['mean_ms'],4),v['launches']) for k,v in d['kernels'].items() if v['mean_ms']*v['launches']>0.1})"
grep -E "passed|failed|^FAILED" $O/t_sel.log | tail -5; cat $O/summary.txt
