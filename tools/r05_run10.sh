cd $GRAFT_REPO_ROOT
bash tools/r05_evidence.sh r05_v1 2>&1 | grep -v "^W2026\|^E2026"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_v1/pytest_all.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r05_v1/pytest_all.log | tail -8
cp gpurun_out/achieved_errors.json gpurun_out/r05_v1/achieved_errors_full2.json
