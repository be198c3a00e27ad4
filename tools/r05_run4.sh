#!/bin/bash
# round 5, fourth GPU run: where the one-pass kernel's time goes (atomics / stores knocked out, frozen
# state), the bad-chunk sort policy (one-pass tests, LWFA, C3 test + bench), C4 with the correction
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run4
mkdir -p $O
timeout 300 python tools/cycle_knock.py 2>&1 | grep -v amdgpu.ids | tee $O/knock.log
mkdir -p /tmp/vhold && mv fbpic_amd/csrc/variants/*.so /tmp/vhold/
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_lwfa.py "tests/test_gpu_configs.py::test_c3_lwfa_full_size" -x -q > $O/pytest_a.log 2>&1
tail -8 $O/pytest_a.log
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('C2', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], d['extra']['particle_passes'], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'spect' in k or 'J_rho' in k or 'home' in k})" | tee -a $O/bench.log
timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>$O/bench_c3.err
python -c "
import json; d=json.loads(open('$O/bench_c3.json').read().strip().split(chr(10))[-1]); print('C3', d['value'], d['ms_per_step'], {k: (v['launches'], round(v['mean_ms'],4)) for k,v in d['kernels'].items()})"
timeout 900 python -m pytest "tests/test_gpu_c4.py::test_c4_lwfa_4096x256_on_8_slabs_reproduces_the_single_domain" -x -q > $O/pytest_c4.log 2>&1
tail -14 $O/pytest_c4.log
