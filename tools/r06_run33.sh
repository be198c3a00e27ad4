#!/bin/bash
# round 6, GPU run 33: the one-pass kernel's wait for vector memory inside the staging (Ruyten coefficients travel
# during its first half; weights requested a chunk ahead) against the wait in front of it
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run33; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_cycle.py -q -x > $O/t_sel.log 2>&1; echo "onepass+cycle rc $?" > $O/summary.txt
for i in 1 2; do KNOCK_REPS=12 timeout 300 python tools/cycle_knock.py 2>&1 | grep -E "default|waitfirst" >> $O/knock.txt; done
V=fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_waitfirst.so "" $V/libfbpic_amd_waitfirst.so; do
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], 'one-pass', d['kernels']['fb_gather_push_deposit_J_rho']['mean_ms'])" >> $O/c2.txt
done
cat $O/knock.txt $O/c2.txt; grep -E "passed|failed" $O/t_sel.log | tail -3; cat $O/summary.txt
