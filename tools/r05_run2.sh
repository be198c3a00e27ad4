#!/bin/bash
# round 5, second GPU run: the paired / re-ordered fused spectral launch (tests, timing alone and inside a
# step against round 4's kernel), SQ counters of the one-pass kernel (merged engine against two engines)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run2
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spectral_cycle.py -x -q > $O/pytest_sc.log 2>&1
tail -12 $O/pytest_sc.log
V=$PWD/fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_sc_sgb.so $V/libfbpic_amd_sc_prev.so; do
  FBPIC_AMD_LIB=$lib timeout 120 python tools/sc_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/sc_time.log
done
for lib in "" $V/libfbpic_amd_sc_sgb.so $V/libfbpic_amd_sc_prev.so; do
  FBPIC_AMD_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'spect' in k or 'zfft' in k or 'J_rho' in k})" | tee -a $O/bench.log
done
mkdir -p /tmp/vhold && mv $V/libfbpic_amd_sc_sgb.so $V/libfbpic_amd_sc_prev.so /tmp/vhold/
timeout 600 bash tools/r05_knock_sq.sh 2>&1 | tee $O/knock_sq.log
