#!/bin/bash
# round 6, GPU run 68: the cubic sorting deposition pass back on the plain cut - C5 line, cubic kernel / config tests
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_v10; mkdir -p $O
python bench.py --config C5 --no-cpu-baseline --no-side-legs 2>/dev/null > $O/bench_c5.json; python -c "
import json; d=json.loads(open('$O/bench_c5.json').read().strip().split('\n')[-1]); k=d['kernels']; print('C5 final', round(d['ms_per_step'],4), [round(x,4) for x in d['extra']['repeat_ms_per_step']], round(k['fb_push_x_sort_deposit_J_rho']['mean_ms'],4), round(k['fb_gather_push_rank_next']['mean_ms'],4))" | tee -a $O/scan_ab.txt
timeout 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_kernels.py -q -m gpu -k "c5 or C5 or cubic or fused" > $O/t_cubic.log 2>&1; echo "cubic tests rc $? : $(grep -E 'passed|failed' $O/t_cubic.log | tail -1)" | tee -a $O/scan_ab.txt
