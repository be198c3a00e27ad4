#!/bin/bash
# round 6, GPU run 21: full GPU suite + bench lines after the Hankel change (straight-line loads, 32 x 128 tile)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run21; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/t_all.log 2>&1; echo "gpu suite rc $?" > $O/summary.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
for i in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/c3_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c3_$i.json').read().strip().split('\n')[-1]); print('C3', d['ms_per_step'], d['roofline']['hankel']['frac'], {k:(round(v['mean_ms'],3),v['launches']) for k,v in d['kernels'].items() if 'hankel' in k})"; done
timeout 600 python bench.py --config C5 --no-cpu-baseline --no-side-legs > $O/c5.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c5.json').read().strip().split('\n')[-1]); print('C5', d['ms_per_step'], d['roofline']['hankel']['frac'], {k:(round(v['mean_ms'],3),v['launches']) for k,v in d['kernels'].items() if 'hankel' in k})"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c2.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().split('\n')[-1]); print('C2', d['ms_per_step'], d['extra']['repeat_ms_per_step'], d['extra']['particle_passes'])"
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee $O/loopback_times.txt
python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee -a $O/loopback_times.txt
grep -E "passed|failed|^FAILED" $O/t_all.log | tail -8; cat $O/summary.txt
