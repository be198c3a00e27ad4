#!/bin/bash
# round 6, GPU run 25: fused spectral launch with the matrix stream as a rolling prefetch (sched_group_barrier)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run25; mkdir -p $O
V=fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_sc_sgb.so $V/libfbpic_amd_sc_sgb8.so $V/libfbpic_amd_sc_sgb32.so "" $V/libfbpic_amd_sc_sgb.so; do
  FBPIC_AMD_LIB=$lib timeout 100 python tools/sc_time.py 2>&1 | grep "per launch" >> $O/sc_time.txt
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], 'spect', d['kernels']['fb_spect_cycle_standard']['mean_ms'])" >> $O/c2.txt
done
FBPIC_AMD_LIB=$V/libfbpic_amd_sc_sgb.so timeout 600 python -m pytest tests/test_gpu_spectral_cycle.py -q > $O/t_sc.log 2>&1; echo "sc tests (sgb) rc $?" >> $O/summary.txt
cat $O/sc_time.txt $O/c2.txt; tail -2 $O/t_sc.log; cat $O/summary.txt
