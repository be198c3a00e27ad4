#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run13; mkdir -p $O
for alt in 1 2; do
FBPIC_AMD_ZFFT_ALT=$alt timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_spectral_cycle.py -q -k "fft or spectral or cycle" > $O/t_fft_$alt.log 2>&1; echo "fft tests alt $alt rc $?" >> $O/summary.txt
done
for alt in 0 1 2 0 1; do
  FBPIC_AMD_ZFFT_ALT=$alt timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs > $O/c2_$alt.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/c2_$alt.json').read().strip().split('\n')[-1]); k=d['kernels']; print('alt $alt', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], 'zfft_rec', k['fb_zfft_from_records_consume']['mean_ms'], 'zfft_pm', k['fb_zfft_pm_to_rt']['mean_ms'])" >> $O/c2.txt 2>&1
done
cat $O/c2.txt; cat $O/summary.txt
