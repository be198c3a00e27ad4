set -x
mkdir -p gpurun_out/v3
python -m pytest tests -m gpu -x -q > gpurun_out/v3/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/v3/pytest.log
tail -3 gpurun_out/v3/pytest.log
python bench.py > gpurun_out/v3/bench.json 2> gpurun_out/v3/bench.err; cat gpurun_out/v3/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/v3/stats -o r01 -- $CMD > /root/repo/gpurun_out/v3/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/v3/fetch -o r01 -- $CMD > /root/repo/gpurun_out/v3/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/v3/write -o r01 -- $CMD > /root/repo/gpurun_out/v3/write.log 2>&1
cd /root/repo
for d in stats fetch write; do
  db=$(find gpurun_out/v3/$d -name '*.db' | head -1)
  mode=pmc; [ $d = stats ] && mode=stats
  python tools/rocpd_summary.py $mode $db gpurun_out/v3/${d}_summary.csv > gpurun_out/v3/${d}_summary.log 2>&1
  rm -f $db
done
ls -la gpurun_out/v3
