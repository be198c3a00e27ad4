# A/B of the default library against fbpic_amd/csrc/variants/*.so: tests of the touched kernels, the
# frozen-state one-pass timing, the two bench sequences
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_kernels.py -x -q 2>&1 | grep "passed\|failed" | tail -2
python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids
for lib in "" $PWD/fbpic_amd/csrc/variants/libfbpic_amd_prev.so; do
  for seq in "1 1" "0 0"; do set -- $seq
    FBPIC_AMD_LIB=$lib FBPIC_AMD_ONE_PASS=$1 FBPIC_AMD_FUSE_SPECT=$2 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=${lib##*/} one_pass=$1', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'deposit' in k or 'gather' in k})"
  done
done
