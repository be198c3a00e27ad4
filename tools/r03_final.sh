mkdir -p gpurun_out/r03
timeout 560 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r03/pytest_final.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest_final.log
grep -v "amdgpu.ids" gpurun_out/r03/pytest_final.log | grep -v "(< " | tail -22
cp gpurun_out/achieved_errors.json gpurun_out/r03_achieved_errors.json
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py > gpurun_out/r03_v2_bench.json 2> gpurun_out/r03/bench.err; tail -c 600 gpurun_out/r03_v2_bench.json; tail -3 gpurun_out/r03/bench.err
bash tools/profile_round.sh r03_v2 --traffic-only > gpurun_out/r03/prof_final.log 2>&1; tail -6 gpurun_out/r03/prof_final.log
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loop_single_final.log
for ov in off fft; do FBPIC_AMD_OVERLAP=$ov python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loop_${ov}_final.log; done
