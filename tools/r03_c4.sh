mkdir -p gpurun_out/r03
timeout 420 python -m pytest tests/test_gpu_c4.py -q --durations=3 2>&1 | grep -v amdgpu.ids | tail -15
cat gpurun_out/c4_timing/*.log
python -m pytest tests/test_gpu_carry.py "tests/test_gpu_multirank_golden.py::test_decomposed_with_second_stream_vs_reference_ranks" -q 2>&1 | grep -v amdgpu.ids | tail -12
