# A/B of compiler scheduling strategies (tools/variant.sh builds): the two-pass bench sequence with the
# deposit.hip / particles.hip variants, the cubic C5 line, the tests of the touched kernels
cd $GRAFT_REPO_ROOT
for lib in "" depmc parmc; do
  L=""; [ -n "$lib" ] && L=$PWD/fbpic_amd/csrc/variants/libfbpic_amd_$lib.so
  FBPIC_AMD_LIB=$L FBPIC_AMD_ONE_PASS=0 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('two-pass lib=$lib', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'gather_push' in k or 'sort_deposit' in k})"
  FBPIC_AMD_LIB=$L python bench.py --config C5 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('C5 lib=$lib', round(d['ms_per_step'],4), {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'gather_push' in k or 'sort_deposit' in k})"
done
timeout 600 python -m pytest tests/test_gpu_onepass.py -x -q 2>&1 | grep "passed\|failed" | tail -1
