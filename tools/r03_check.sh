# one GPU call: parity suite, bench line, step(1) loop, decomposed loopback
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest.log
tail -60 gpurun_out/r03/pytest.log
python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err; tail -c 3000 gpurun_out/r03/bench.json
python tools/step1_loop.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/step1.log
python tools/first_exchange.py 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r03/first_exchange.log
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loop_single.log
python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loop.log
