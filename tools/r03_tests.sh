mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest.log
tail -80 gpurun_out/r03/pytest.log
