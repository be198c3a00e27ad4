set -x
V=${1:-v4}
mkdir -p gpurun_out/$V
python -m pytest tests -m gpu -x -q > gpurun_out/$V/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/$V/pytest.log
tail -3 gpurun_out/$V/pytest.log
python bench.py > gpurun_out/$V/bench.json 2> gpurun_out/$V/bench.err; cat gpurun_out/$V/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/$V/stats -o r01 -- $CMD > /root/repo/gpurun_out/$V/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/$V/fetch -o r01 -- $CMD > /root/repo/gpurun_out/$V/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/$V/write -o r01 -- $CMD > /root/repo/gpurun_out/$V/write.log 2>&1
cd /root/repo
for d in stats fetch write; do
  db=$(find gpurun_out/$V/$d -name '*.db' | head -1)
  mode=pmc; [ $d = stats ] && mode=stats
  python tools/rocpd_summary.py $mode $db gpurun_out/$V/${d}_summary.csv > gpurun_out/$V/${d}_summary.log 2>&1
  rm -f $db
done
ls -la gpurun_out/$V
