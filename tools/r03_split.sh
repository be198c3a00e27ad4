python -m pytest tests/test_gpu_kernels.py -q -k "range or rank_next" 2>&1 | grep -v amdgpu.ids | tail -5
python -m pytest "tests/test_gpu_multirank_golden.py::test_decomposed_with_second_stream_vs_reference_ranks" -q 2>&1 | grep -v amdgpu.ids | tail -14
