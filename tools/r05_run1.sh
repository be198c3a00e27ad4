#!/bin/bash
# round 5, first GPU run: the merged deposition engine of the one-pass kernel - its tests, frozen-state
# timing against round 4's two-engine form (variant build), bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_onepass.py -x -q > $O/pytest_onepass.log 2>&1
tail -15 $O/pytest_onepass.log
timeout 300 python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids | tee $O/knock.log
for lib in "" $PWD/fbpic_amd/csrc/variants/libfbpic_amd_two_engines.so; do
  FBPIC_AMD_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'deposit' in k or 'gather' in k})" | tee -a $O/bench.log
done
