#!/bin/bash
# round 6, GPU run 38: particle streams of the one-pass kernel marked non-temporal - do the grids / the spectral slab
# survive in L2 / the Infinity Cache for the field kernels?
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run38; mkdir -p $O
V=fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_nt.so "" $V/libfbpic_amd_nt.so; do
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], 'one-pass', k['fb_gather_push_deposit_J_rho']['mean_ms'], 'spect', k['fb_spect_cycle_standard']['mean_ms'], 'zfft', k['fb_zfft_from_records_consume']['mean_ms'], k['fb_zfft_pm_to_rt']['mean_ms'])" | tee -a $O/c2.txt
done
FBPIC_AMD_LIB=$V/libfbpic_amd_nt.so timeout 600 python -m pytest tests/test_gpu_onepass.py -q -x -k "linear" > $O/t.log 2>&1; echo "onepass (nt) rc $?"; tail -1 $O/t.log
