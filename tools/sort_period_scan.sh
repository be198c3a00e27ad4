# sort period of the one-pass cycle at C2: the driver's command (headline = first value) and a longer run
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for P in 2 3; do
  FBPIC_AMD_SORT_PERIOD=$P timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('driver cmd, period $P', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], d['extra']['particle_passes'])"
done; done
for P in 2 3 4; do
  FBPIC_AMD_SORT_PERIOD=$P timeout 300 python bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('60 steps, period $P', round(d['ms_per_step'],4), [round(v,4) for v in d['extra']['repeat_ms_per_step']], d['extra']['particle_passes'])"
done
