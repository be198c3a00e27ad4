#!/bin/bash
# which waits: instruction fetch / scalar cache / vector memory counters of the one-pass kernel (frozen state)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run8
mkdir -p $O gpurun_out/probe
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|SQC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TA_[A-Z0-9_]+|TCC_[A-Z_0-9]+)\b" | sort -u > /root/repo/$O/counters.txt
wc -l /root/repo/$O/counters.txt
grep -E "WAIT|IFETCH|ICACHE|DCACHE|SMEM|LEVEL|STALL|BUSY" /root/repo/$O/counters.txt | tr '\n' ' '
export KNOCK_R=1 KNOCK_REPS=6
i=0
for G in \
 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM" \
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVE_CYCLES" \
 "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --pmc $G -d /root/repo/gpurun_out/probe/kw$i -o p -- python /root/repo/tools/cycle_knock.py > /root/repo/gpurun_out/probe/kw$i.log 2>&1
  db=$(find /root/repo/gpurun_out/probe/kw$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python /root/repo/tools/rocpd_summary.py pmcseq $db /root/repo/gpurun_out/knockw_seq$i.csv; else tail -3 /root/repo/gpurun_out/probe/kw$i.log; fi
  rm -rf /root/repo/gpurun_out/probe/kw$i
done
cd /root/repo
python - <<'PY' | tee $O/waits.txt
import csv, collections, glob, os
libs = ['default'] + sorted(os.path.basename(p).replace('libfbpic_amd_', '').replace('.so', '') for p in glob.glob('fbpic_amd/csrc/variants/*.so'))
for f in sorted(glob.glob('gpurun_out/knockw_seq*.csv')):
    rows = [r for r in csv.DictReader(open(f)) if 'false, false' in r['kernel'] or 'ELb0ELb0' in r['kernel']]
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(r['dispatch'], {})[r['counter']] = float(r['value'])
    d = list(disp.values())
    n = 6 * len(libs)
    tail = d[-n:]
    for i, name in enumerate(libs):
        g = tail[6 * i: 6 * i + 6][2:]
        if not g: continue
        avg = {k: sum(x[k] for x in g) / len(g) for k in g[0]}
        print('%-12s' % name + '  '.join('%s %.1f' % (k.replace('SQ_', '').replace('INSTS_', ''), v / 65536.) for k, v in sorted(avg.items())))
PY
