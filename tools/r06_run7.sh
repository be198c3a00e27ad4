#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run7; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -k "not c3_full_grid" > $O/t_all.log 2>&1; echo "gpu suite rc $?" > $O/summary.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee $O/loopback_times.txt
python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee -a $O/loopback_times.txt
grep -E "passed|failed|FAILED|Error" $O/t_all.log | tail -15; cat $O/summary.txt
