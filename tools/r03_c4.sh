mkdir -p gpurun_out/r03
for i in 1 2; do
  timeout 400 python -m pytest tests/test_gpu_c4.py -q -k c4_lwfa --durations=2 2>&1 | grep -v amdgpu.ids | tail -6
  tail -n 2 gpurun_out/c4_timing/w8_r0.log
done
ls gpurun_out/c4_timing/; for f in gpurun_out/c4_timing/*stack.log; do if [ -s $f ]; then echo "== $f"; head -40 $f; fi; done
