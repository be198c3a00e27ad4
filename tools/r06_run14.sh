#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run14; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_lwfa.py tests/test_gpu_multirank_golden.py tests/test_gpu_carry.py tests/test_gpu_cycle.py -q --tb=short > $O/t_sel.log 2>&1; echo "selected rc $?" >> $O/summary.txt
timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/c3.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/c3.json').read().strip().split('\n')[-1]); print('C3', d['ms_per_step'], {k:(round(v['mean_ms'],3),v['launches']) for k,v in d['kernels'].items() if v['mean_ms']*v['launches']>0.3})"
grep -E "passed|failed|^FAILED|assert" $O/t_sel.log | head -20; cat $O/summary.txt
