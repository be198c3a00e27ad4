# one GPU call: parity suite, bench line, step(1) loop, decomposed loopback
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r03/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest.log
grep -v "amdgpu.ids" gpurun_out/r03/pytest.log | tail -150
python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err; tail -c 1500 gpurun_out/r03/bench.json
python tools/step1_loop.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/step1.log
python tools/first_exchange.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r03/first_exchange.log
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loop_single.log
for ov in fft off; do FBPIC_AMD_OVERLAP=$ov python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loop_$ov.log; done
