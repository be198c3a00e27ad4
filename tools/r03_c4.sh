mkdir -p gpurun_out/r03
for i in 1 2 3 4; do
  timeout 200 python -m pytest tests/test_gpu_c4.py -q -s -k c4_lwfa > gpurun_out/r03/c4_try$i.log 2>&1
  if grep -q "1 passed" gpurun_out/r03/c4_try$i.log; then echo "try $i ok"; else echo "try $i FAILED"; mkdir -p gpurun_out/r03/c4_fail; cp gpurun_out/c4_timing/* gpurun_out/r03/c4_fail/; break; fi
done
