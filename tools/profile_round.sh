# usage: bash tools/profile_round.sh <tag> [--traffic-only] [bench args]
# One --kernel-trace --stats run and separate --pmc runs (FETCH_SIZE, WRITE_SIZE, MFMA counters) of
# the same bench.py command; summaries under gpurun_out/<tag>_*.csv (copy to profiles/).
# --traffic-only: kernel stats + FETCH_SIZE + WRITE_SIZE (the passes bench.py's roofline.traffic reads).
TAG=${1:-r03}; shift
PASSES="stats fetch write mfma mfmautil"
if [ "$1" = "--traffic-only" ]; then PASSES="stats fetch write"; shift; fi
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-legs $@"
for d in $PASSES; do
  case $d in
    stats) OPT="--kernel-trace --stats";;
    fetch) OPT="--pmc FETCH_SIZE";;
    write) OPT="--pmc WRITE_SIZE";;
    mfma) OPT="--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES";;
    mfmautil) OPT="--pmc MfmaUtil";;
  esac
  rocprofv3 $OPT -d /root/repo/gpurun_out/$TAG/$d -o r -- $CMD > /root/repo/gpurun_out/$TAG/$d.log 2>&1
done
cd /root/repo
# which box the counters come from (bench.py prints it next to roofline.traffic: the passes are runs of
# their own, usually on another box of the pool than the bench line that quotes them)
python - > gpurun_out/${TAG}_pmc_box.json <<'PY'
import json, sys
sys.path.insert(0, '/root/repo')
import torch, bench
print(json.dumps(bench.device_identity(torch)))
PY
for d in $PASSES; do
  db=$(find gpurun_out/$TAG/$d -name '*.db' | head -1)
  mode=pmc; [ $d = stats ] && mode=stats
  name=$d; [ $d = stats ] && name=kernel_stats; [ $d = fetch ] && name=pmc_fetch_size; [ $d = write ] && name=pmc_write_size
  [ $d = mfma ] && name=pmc_mfma; [ $d = mfmautil ] && name=pmc_mfmautil
  if [ -n "$db" ]; then python tools/rocpd_summary.py $mode $db gpurun_out/${TAG}_${name}.csv; fi
  rm -rf gpurun_out/$TAG/$d
done
head -12 gpurun_out/${TAG}_kernel_stats.csv
