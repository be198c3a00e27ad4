mkdir -p gpurun_out/r03
timeout 420 python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_final2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest_final2.log
grep -v "amdgpu.ids" gpurun_out/r03/pytest_final2.log | grep -v "(< " | tail -14
cp gpurun_out/achieved_errors.json gpurun_out/r03_achieved_errors_final.json
