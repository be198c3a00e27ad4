mkdir -p gpurun_out/r03
bash tools/profile_round.sh r03_v1_c5 --traffic-only --config C5
python bench.py --config C5 --no-cpu-baseline > gpurun_out/r03_v1_bench_c5.json 2>/dev/null; tail -c 400 gpurun_out/r03_v1_bench_c5.json
