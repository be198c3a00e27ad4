#!/bin/bash
# round-5 evidence: bench lines (C2 default with the CPU baseline, C3, C5), rocprofv3 kernel stats + PMC
# passes of the default C2 command, SQ counters of the one-pass sequence, decomposed step on one rank
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_v1}
O=gpurun_out/$TAG
mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
timeout 400 python bench.py --config C5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
for f in bench bench_c3 bench_c5; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d.get('extra',{}).get('repeat_ms_per_step'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline'].get('hankel',{}).get('frac'))"; done
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
head -14 gpurun_out/${TAG}_kernel_stats.csv
bash tools/sq_probe.sh ${TAG}_onepass tools/onepass_only.py 24 > $O/sq.log 2>&1
tail -60 $O/sq.log | head -70
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee $O/loopback_times.txt
python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee -a $O/loopback_times.txt
