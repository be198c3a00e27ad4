#!/bin/bash
# round 6, GPU run 53: 1024-row z-FFT with more waves per tile (-DFB_ZC1024_ALT=1: 8 points per lane / 512 lanes / radix 8 8 4 4;
# =2: 4 points per lane / 1024 lanes / radix 4^5) against 16 points per lane / 256 lanes / radix 16 8 8: FFT tests, C2 bench
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run53; mkdir -p $O
for a in 1 2; do
  FBPIC_AMD_LIB=/root/repo/fbpic_amd/csrc/variants/libfbpic_amd_zalt$a.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "fft" -m gpu 2>&1 | tail -1 | sed "s/^/zalt$a fft tests: /" | tee -a $O/scan.txt
done
for a in 0 1 2 0 1 2; do
  L=/root/repo/fbpic_amd/csrc/libfbpic_amd.so; [ $a != 0 ] && L=/root/repo/fbpic_amd/csrc/variants/libfbpic_amd_zalt$a.so
  FBPIC_AMD_LIB=$L python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('zalt$a', round(d['ms_per_step'],4), [round(x,4) for x in d['extra']['repeat_ms_per_step']], {n: v['mean_ms'] for n, v in k.items() if 'fft' in n})" | tee -a $O/scan.txt
done
