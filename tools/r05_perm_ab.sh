#!/bin/bash
# merged engine in the sorting second pass (fb_push_x_sort_deposit_J_rho): tests that cover it, then A/B against
# the two-engine build and two split thresholds at C2 and C3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_perm_ab
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cycle.py tests/test_gpu_onepass.py tests/test_gpu_lwfa.py "tests/test_gpu_configs.py::test_c3_lwfa_full_size" -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
V=$PWD/fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_perm2.so $V/libfbpic_amd_split3.so $V/libfbpic_amd_split12.so; do
  FBPIC_AMD_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('C2 lib=${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'J_rho' in k or 'home' in k})" | tee -a $O/bench.log
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('C3 lib=${lib##*/}', round(d['ms_per_step'],4), {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'J_rho' in k or 'home' in k})" | tee -a $O/bench.log
done
