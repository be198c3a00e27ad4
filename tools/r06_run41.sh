#!/bin/bash
# round 6, GPU run 41: non-temporal loads in the solver step of big grids (default) against plain, and the Hankel A operand
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run41; mkdir -p $O
V=fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_psplain.so $V/libfbpic_amd_hknt.so "" $V/libfbpic_amd_psplain.so $V/libfbpic_amd_hknt.so; do
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C3 ${lib##*/}', round(d['ms_per_step'],4), {n[3:]: round(v['mean_ms'],3) for n,v in k.items() if v['mean_ms']*v['launches']>0.3 and ('hankel' in n or 'psatd' in n or 'fft' in n)})" | tee -a $O/ab.txt
done
for lib in "" $V/libfbpic_amd_psplain.so $V/libfbpic_amd_hknt.so; do
  FBPIC_AMD_LIB=$lib timeout 600 python bench.py --config C5 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C5 ${lib##*/}', round(d['ms_per_step'],4), {n[3:]: round(v['mean_ms'],3) for n,v in k.items() if ('hankel' in n or 'psatd' in n or 'fft' in n)})" | tee -a $O/ab.txt
done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lwfa.py -q -x -k "spectral or psatd or lwfa or hankel" > $O/t.log 2>&1; echo "tests rc $?"; tail -1 $O/t.log
