#!/bin/bash
# SQ counters per 64 particles of the one-pass kernel on the frozen state (tools/cycle_knock.py), default
# library and every variant build; one rocprofv3 --pmc pass per counter group
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
export KNOCK_R=${KNOCK_R:-1} KNOCK_REPS=6
cd /tmp && export TMPDIR=/tmp
i=0
for G in \
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  rocprofv3 --pmc $G -d /root/repo/gpurun_out/probe/kn$i -o p -- python /root/repo/tools/cycle_knock.py > /root/repo/gpurun_out/probe/kn$i.log 2>&1
  db=$(find /root/repo/gpurun_out/probe/kn$i -name '*.db' | head -1)
  python /root/repo/tools/rocpd_summary.py pmcseq $db /root/repo/gpurun_out/knock_seq$i.csv
  rm -rf /root/repo/gpurun_out/probe/kn$i
done
cd /root/repo
tail -4 gpurun_out/probe/kn1.log
python - <<'PY'
import csv, collections, glob, os
libs = ['default'] + sorted(os.path.basename(p).replace('libfbpic_amd_', '').replace('.so', '') for p in glob.glob('fbpic_amd/csrc/variants/*.so'))
for f in sorted(glob.glob('gpurun_out/knock_seq*.csv')):
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
        print('%-12s' % name + '  '.join('%s %.0f' % (k.replace('SQ_', '').replace('INSTS_', ''), v / 65536.) for k, v in sorted(avg.items())))
PY
