#!/bin/bash
# round 6, GPU run 40: non-temporal LOADS in the big-grid FFT sweeps (what a sweep writes is what the next launch reads)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run40; mkdir -p $O
V=fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_fftplain.so "" $V/libfbpic_amd_fftplain.so; do
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C3 ${lib##*/}', round(d['ms_per_step'],4), {n[3:]: round(v['mean_ms'],3) for n,v in k.items() if v['mean_ms']*v['launches']>0.3})" | tee -a $O/ab.txt
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "fft" > $O/t.log 2>&1; echo "fft tests rc $?"; tail -1 $O/t.log
