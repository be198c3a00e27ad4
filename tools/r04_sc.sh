cd $GRAFT_REPO_ROOT
python tools/sc_time.py 2>&1 | grep -v amdgpu.ids
FBPIC_AMD_SC_WAVES=4 python tools/sc_time.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_spectral_cycle.py -x -q 2>&1 | tail -3
