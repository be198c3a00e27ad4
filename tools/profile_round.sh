# usage: bash tools/profile_round.sh <tag> [bench args]
# One --kernel-trace --stats run and separate --pmc runs (FETCH_SIZE, WRITE_SIZE, MFMA counters) of
# the same bench.py command; summaries under gpurun_out/<tag>_*.csv (copy to profiles/).
TAG=${1:-r02}; shift
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $@"
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/$TAG/stats -o r -- $CMD > /root/repo/gpurun_out/$TAG/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/$TAG/fetch -o r -- $CMD > /root/repo/gpurun_out/$TAG/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/$TAG/write -o r -- $CMD > /root/repo/gpurun_out/$TAG/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES -d /root/repo/gpurun_out/$TAG/mfma -o r -- $CMD > /root/repo/gpurun_out/$TAG/mfma.log 2>&1
rocprofv3 --pmc MfmaUtil -d /root/repo/gpurun_out/$TAG/mfmautil -o r -- $CMD > /root/repo/gpurun_out/$TAG/mfmautil.log 2>&1
cd /root/repo
for d in stats fetch write mfma mfmautil; do
  db=$(find gpurun_out/$TAG/$d -name '*.db' | head -1)
  mode=pmc; [ $d = stats ] && mode=stats
  name=$d; [ $d = stats ] && name=kernel_stats; [ $d = fetch ] && name=pmc_fetch_size; [ $d = write ] && name=pmc_write_size
  [ $d = mfma ] && name=pmc_mfma; [ $d = mfmautil ] && name=pmc_mfmautil
  if [ -n "$db" ]; then python tools/rocpd_summary.py $mode $db gpurun_out/${TAG}_${name}.csv; fi
  rm -rf gpurun_out/$TAG/$d
done
head -16 gpurun_out/${TAG}_kernel_stats.csv
grep -i hankel gpurun_out/${TAG}_pmc_mfma.csv gpurun_out/${TAG}_pmc_mfmautil.csv
