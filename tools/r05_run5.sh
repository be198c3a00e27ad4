#!/bin/bash
# round 5, fifth GPU run: grouped kernel-argument fetches (A/B, frozen state + bench), then the whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run5
mkdir -p $O
timeout 300 python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids | tee $O/knock.log
for lib in "" $PWD/fbpic_amd/csrc/variants/libfbpic_amd_kp_single.so; do
  FBPIC_AMD_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'spect' in k or 'J_rho' in k or 'home' in k})" | tee -a $O/bench.log
done
mkdir -p /tmp/vhold && mv fbpic_amd/csrc/variants/*.so /tmp/vhold/
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest_all.log | tail -15
cp gpurun_out/achieved_errors.json $O/achieved_errors_full1.json
