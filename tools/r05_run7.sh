#!/bin/bash
# round 5, seventh GPU run: LDS co-residency of one-wave workgroups around 12-14 KB; the merged engine with an
# odd panel stride (bank conflicts) against 66 / 67 and two engines: frozen timing + SQ counters; loopback trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run7
mkdir -p $O
./tools/lds_probe 8192 10240 11264 12032 12288 12544 12800 12896 13056 13312 13568 13824 14208 14336 15360 2>&1 | tee $O/lds_probe.txt
timeout 300 python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids | tee $O/knock.log
timeout 600 bash tools/r05_knock_sq.sh 2>&1 | grep -v "^W2026\|^E2026" | tee $O/knock_sq.log
mkdir -p /tmp/vhold && mv fbpic_amd/csrc/variants/*.so /tmp/vhold/
timeout 300 python -m pytest tests/test_gpu_onepass.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('C2', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'spect' in k or 'J_rho' in k or 'home' in k})" | tee -a $O/bench.log
bash tools/loopback_profile.sh > $O/loopback_trace.txt 2>&1
tail -45 $O/loopback_trace.txt
