#!/bin/bash
# round 6, GPU run 1: persistent-wave one-pass kernel - parity first, then A/B against round 5's kernel
# (fbpic_amd/csrc/variants/libfbpic_amd_r05cycle.so) on a frozen state, scan of range cap / waves per CU
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_first_exchange.py -x -q > $O/t_onepass.log 2>&1; echo "onepass rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -s > $O/t_fullsize.log 2>&1; echo "fullsize rc $?" >> $O/summary.txt
for cfg in "4 12" "2 12" "1 12" "8 12" "4 16" "2 16" "3 12" "4 11"; do
  set -- $cfg
  echo "== range $1 waves/CU $2" >> $O/knock.txt
  FBPIC_AMD_CYCLE_RANGE=$1 FBPIC_AMD_CYCLE_WAVES_PER_CU=$2 timeout 300 python tools/cycle_knock.py >> $O/knock.txt 2>&1
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc $?" >> $O/summary.txt
tail -5 $O/t_onepass.log $O/t_fullsize.log $O/t_all.log; cat $O/knock.txt | grep -v Warn | tail -60; cat $O/summary.txt
