cd $GRAFT_REPO_ROOT
for p in 2 3 4 5 6; do
FBPIC_AMD_SORT_PERIOD=$p python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('period $p', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], d['extra']['particle_passes'])"
done
