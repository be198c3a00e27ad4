#!/bin/bash
# round-4 evidence: bench lines (C2 default, C2 two-pass sequence, C3, C5) + rocprofv3 kernel stats and
# PMC passes of the default C2 command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_v2
mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
FBPIC_AMD_ONE_PASS=0 FBPIC_AMD_FUSE_SPECT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_two_pass.json 2>/dev/null
timeout 300 python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
timeout 400 python bench.py --config C5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
bash tools/profile_round.sh r04_v2 > $O/profile.log 2>&1
for f in bench bench_two_pass bench_c3 bench_c5; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d.get('extra',{}).get('repeat_ms_per_step'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline'].get('hankel',{}).get('frac'))"; done
head -14 gpurun_out/r04_v2_kernel_stats.csv
bash tools/sq_probe.sh r04_onepass_v2 tools/onepass_only.py 24 > $O/sq.log 2>&1
tail -45 $O/sq.log | head -60
