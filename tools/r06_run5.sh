#!/bin/bash
# round 6, GPU run 5: cubic one-pass kernel (k_cycle_cubic): parity, then C5 with and without it
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_onepass.py -x -q > $O/t_onepass.log 2>&1; echo "onepass rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "array_wrapper or push_sort_deposit_J_rho_fused" > $O/t_kernels.log 2>&1; echo "kernels rc $?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q -k "deferral" > $O/t_deferral.log 2>&1; echo "deferral rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "cubic_nm4 or c5_size" > $O/t_c5.log 2>&1; echo "c5 tests rc $?" >> $O/summary.txt
FBPIC_AMD_ONE_PASS_CUBIC=1 timeout 900 python bench.py --config C5 --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs > $O/bench_c5_onepass.json 2> $O/bench_c5_onepass.err; echo "c5 onepass rc $?" >> $O/summary.txt
FBPIC_AMD_ONE_PASS_CUBIC=0 timeout 900 python bench.py --config C5 --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs > $O/bench_c5_twopass.json 2> $O/bench_c5_twopass.err; echo "c5 twopass rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -s > $O/t_fullsize.log 2>&1; echo "fullsize rc $?" >> $O/summary.txt
tail -25 $O/t_onepass.log; tail -5 $O/t_kernels.log $O/t_deferral.log $O/t_c5.log; grep -E "C2 full|C5 full|Error" $O/t_fullsize.log; cat $O/summary.txt
for f in onepass twopass; do python -c "
import json; d=json.load(open('$O/bench_c5_$f.json')); print('$f', d['ms_per_step'], d['extra'].get('repeat_ms_per_step'), d['extra']['particle_passes']); print({k:(round(v['mean_ms'],3),v['launches']) for k,v in d['kernels'].items() if v['mean_ms']>0.05})"; done
