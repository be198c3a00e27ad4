cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_run11
timeout 300 python -m pytest tests/test_gpu_onepass.py -x -q 2>&1 | tail -2
python tools/hosttime.py 2>&1 | grep -v amdgpu.ids | head -30 | tee gpurun_out/r05_run11/hosttime.txt
bash tools/stats_only.sh 2>&1 | grep -v "^W2026\|^E2026" | head -40 | tee gpurun_out/r05_run11/gaps.txt
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('C2', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], d['extra']['particle_passes'], {k: round(v['mean_ms'],4) for k,v in d['kernels'].items() if 'spect' in k or 'J_rho' in k or 'home' in k})"
