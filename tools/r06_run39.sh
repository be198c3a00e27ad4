#!/bin/bash
# round 6, GPU run 39: non-temporal particle streams in every particle kernel (default) against plain accesses
# (-DFB_NO_NT) and against plain permuted gathers only (-DFB_NO_NT_GATHER): C2, C3, C5; tests
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run39; mkdir -p $O
V=fbpic_amd/csrc/variants
for lib in "" $V/libfbpic_amd_nont.so $V/libfbpic_amd_nogather.so "" $V/libfbpic_amd_nont.so $V/libfbpic_amd_nogather.so; do
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C2 ${lib##*/}', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {n[3:]: round(1e3*v['mean_ms'],1) for n,v in k.items() if v['mean_ms']*v['launches']>0.1})" | tee -a $O/ab.txt
done
for lib in "" $V/libfbpic_amd_nont.so $V/libfbpic_amd_nogather.so "" $V/libfbpic_amd_nont.so; do
  FBPIC_AMD_LIB=$lib timeout 400 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C3 ${lib##*/}', round(d['ms_per_step'],4), {n[3:]: round(v['mean_ms'],3) for n,v in k.items() if v['mean_ms']*v['launches']>0.3})" | tee -a $O/ab.txt
done
for lib in "" $V/libfbpic_amd_nont.so $V/libfbpic_amd_nogather.so; do
  FBPIC_AMD_LIB=$lib timeout 600 python bench.py --config C5 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']; print('C5 ${lib##*/}', round(d['ms_per_step'],4), {n[3:]: round(v['mean_ms'],3) for n,v in k.items() if v['mean_ms']*v['launches']>1})" | tee -a $O/ab.txt
done
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_onepass.py tests/test_gpu_cycle.py -q -x > $O/t_sel.log 2>&1; echo "selected rc $?"; tail -1 $O/t_sel.log
