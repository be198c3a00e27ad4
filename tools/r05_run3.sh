#!/bin/bash
# round 5, third GPU run: merged engine with the 15-row panel (12 waves per CU again), the moving-window
# one-pass path (LWFA tests, C3 at full size, bench C3), C4 with the current correction
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_lwfa.py "tests/test_gpu_configs.py::test_c3_lwfa_full_size" -x -q > $O/pytest_a.log 2>&1
tail -25 $O/pytest_a.log
timeout 300 python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids | tee $O/knock.log
for lib in "" $PWD/fbpic_amd/csrc/variants/libfbpic_amd_two_engines.so; do
  FBPIC_AMD_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'spect' in k or 'J_rho' in k or 'home' in k})" | tee -a $O/bench.log
done
timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>$O/bench_c3.err
python -c "
import json; d=json.loads(open('$O/bench_c3.json').read().strip().split(chr(10))[-1]); print('C3', d['value'], d['ms_per_step'], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items()})"
timeout 900 python -m pytest tests/test_gpu_c4.py -x -q > $O/pytest_c4.log 2>&1
tail -15 $O/pytest_c4.log
