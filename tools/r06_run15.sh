#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run15; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "gpu suite rc $?" > $O/summary.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
for i in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/c3_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c3_$i.json').read().strip().split('\n')[-1]); print('C3', d['ms_per_step'])"; done
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c2.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().split('\n')[-1]); print('C2', d['ms_per_step'], d['extra']['repeat_ms_per_step'], d['extra']['particle_passes'])"
grep -E "passed|failed|^FAILED" $O/t_all.log | tail -8; cat $O/summary.txt
