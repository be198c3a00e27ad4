#!/bin/bash
# final-state run of the round: the whole suite, smoke(), the default bench line, C3 / C5 lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest_all.log | tail -8
cp gpurun_out/achieved_errors.json $O/achieved_errors.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
timeout 400 python bench.py --config C5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
for f in bench bench_c3 bench_c5; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d.get('extra',{}).get('repeat_ms_per_step'), {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'J_rho' in k or 'home' in k or 'spect' in k or 'rank_next' in k})"; done
