mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_configs.py -q -k "cubic_nm4 or c5_size" 2>&1 | grep -v amdgpu.ids | tail -12
for t in 0 auto; do
  if [ $t = auto ]; then unset FBPIC_AMD_GATHER_TILES; else export FBPIC_AMD_GATHER_TILES=$t; fi
  python bench.py --config C5 --no-cpu-baseline --steps 12 2>/dev/null > gpurun_out/r03/c5_tiles_$t.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r03/c5_tiles_$t.json'))
print('tiles=$t', d['ms_per_step'], {k:v['mean_ms'] for k,v in d['kernels'].items() if v['mean_ms']>0.3})
PY
done
