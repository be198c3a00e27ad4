mkdir -p gpurun_out/r03
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_c4.py -q -s -k c4_lwfa > gpurun_out/r03/c4_try$i.log 2>&1
  if grep -q "1 failed" gpurun_out/r03/c4_try$i.log; then echo "try $i FAILED"; break; else echo "try $i ok"; fi
done
dmesg 2>/dev/null | tail -20 > gpurun_out/r03/dmesg.log
