# decomposed step on one rank (loopback): overlap on / off, kernel trace + host time
mkdir -p gpurun_out/r03
for ov in 1 0; do
  export FBPIC_AMD_OVERLAP=$ov
  python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loop_ov$ov.log
  python tools/hosttime.py --decomposed 2>&1 | grep -v amdgpu.ids | head -30 | tee gpurun_out/r03/host_ov$ov.log
  bash tools/loopback_profile.sh > gpurun_out/r03/looptrace_ov$ov.txt 2>&1
  tail -45 gpurun_out/r03/looptrace_ov$ov.txt
done
python -m pytest tests/test_gpu_c4.py tests/test_gpu_multirank.py tests/test_gpu_multirank_golden.py tests/test_gpu_configs.py -x -q 2>&1 | tail -40
