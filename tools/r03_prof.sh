mkdir -p gpurun_out/r03
python -m pytest "tests/test_gpu_multirank_golden.py::test_decomposed_with_second_stream_vs_reference_ranks" -q -x 2>&1 | grep -v amdgpu.ids | grep -B5 -A40 "Error\|error" | head -80
bash tools/profile_round.sh r03_v1
bash tools/profile_round.sh r03_v1_c3 --traffic-only --config C3
python bench.py --config C3 --no-cpu-baseline > gpurun_out/r03_v1_bench_c3.json 2>/dev/null; tail -c 600 gpurun_out/r03_v1_bench_c3.json
