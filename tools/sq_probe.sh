# usage: bash tools/sq_probe.sh <tag> <python script + args>
# Collects SQ issue / wait / instruction-mix counters of every fb:: kernel the command
# launches, one rocprofv3 --pmc pass per counter group (counters need runs of their own),
# into gpurun_out/sq_<tag>.csv (kernel, counter, dispatches, avg_value, avg_duration_us).
TAG=$1; shift
mkdir -p gpurun_out/probe
OUT=/root/repo/gpurun_out/sq_$TAG.csv
: > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in \
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
 "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH" ; do
  i=$((i+1))
  rocprofv3 --pmc $G -d /root/repo/gpurun_out/probe/sq$i -o p -- python /root/repo/$@ > /root/repo/gpurun_out/probe/sq$i.log 2>&1
  db=$(find /root/repo/gpurun_out/probe/sq$i -name '*.db' | head -1)
  if [ -n "$db" ]; then
    python /root/repo/tools/rocpd_summary.py pmc $db /root/repo/gpurun_out/probe/sq$i.csv && grep "fb::" /root/repo/gpurun_out/probe/sq$i.csv >> $OUT
  else
    echo "pass $i failed" >> $OUT; tail -3 /root/repo/gpurun_out/probe/sq$i.log >> $OUT
  fi
  rm -rf /root/repo/gpurun_out/probe/sq$i
done
cd /root/repo
python - <<PY
import csv, collections
rows = list(csv.reader(open('$OUT')))
t = collections.defaultdict(dict)
for r in rows:
    if len(r) >= 5: t[r[0]][r[1]] = (float(r[3]), int(r[2]), float(r[4]))
for k, c in t.items():
    print(k)
    for n in sorted(c): print('   %-32s %14.0f  (n=%d, %.1f us)' % (n, c[n][0], c[n][1], c[n][2]))
PY
