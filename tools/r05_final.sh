#!/bin/bash
# final-state run of the round: A/B of the ranking kernel (4 against 3 waves per SIMD), the whole suite,
# smoke(), the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final
mkdir -p $O
for lib in "" $PWD/fbpic_amd/csrc/variants/libfbpic_amd_rank3.so; do
  FBPIC_AMD_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'J_rho' in k or 'home' in k})" | tee -a $O/bench_ab.log
done
mkdir -p /tmp/vhold && mv fbpic_amd/csrc/variants/*.so /tmp/vhold/ 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest_all.log | tail -8
cp gpurun_out/achieved_errors.json $O/achieved_errors.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().split(chr(10))[-1]); print('final', d['value'], d['ms_per_step'], d['extra']['repeat_ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_box'))"
