#!/bin/bash
# round 5, sixth GPU run: SQ counters (merged engine, 15-row panel, grouped kernarg fetches) against two
# engines; spectral forward + correction launch (tests, decomposed golden runs); decomposed loopback step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run6
mkdir -p $O
timeout 600 bash tools/r05_knock_sq.sh 2>&1 | grep -v "^W2026\|^E2026" | tee $O/knock_sq.log
mkdir -p /tmp/vhold && mv fbpic_amd/csrc/variants/*.so /tmp/vhold/
timeout 900 python -m pytest tests/test_gpu_spectral_cycle.py tests/test_gpu_multirank_golden.py tests/test_gpu_multirank.py "tests/test_gpu_configs.py::test_c3_lwfa_full_size" -x -q > $O/pytest_a.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest_a.log | tail -8
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee $O/loopback_times.txt
python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee -a $O/loopback_times.txt
bash tools/loopback_profile.sh > $O/loopback_trace.txt 2>&1
tail -45 $O/loopback_trace.txt
