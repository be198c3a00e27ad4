cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
export KNOCK_R=1 KNOCK_REPS=6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR -d /root/repo/gpurun_out/probe/kn -o p -- python /root/repo/tools/cycle_knock.py > /root/repo/gpurun_out/probe/kn.log 2>&1
cd /root/repo
tail -12 gpurun_out/probe/kn.log
db=$(find gpurun_out/probe/kn -name '*.db' | head -1)
python tools/rocpd_summary.py pmcseq $db gpurun_out/knock_seq.csv
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/knock_seq.csv')))
rows = [r for r in rows if 'false, false' in r['kernel'] or 'ELb0ELb0' in r['kernel']]
disp = collections.OrderedDict()
for r in rows:
    disp.setdefault(r['dispatch'], {})[r['counter']] = float(r['value'])
d = list(disp.values())
print(len(d), 'dispatches')
import glob, os
libs = ['default'] + sorted(os.path.basename(p).replace('libfbpic_amd_', '').replace('.so', '') for p in glob.glob('fbpic_amd/csrc/variants/*.so'))
n = 6 * len(libs)
tail = d[-n:]
for i, name in enumerate(libs):
    g = tail[6 * i: 6 * i + 6][2:]
    avg = {k: sum(x[k] for x in g) / len(g) for k in g[0]}
    ch = 65536.
    print('%-10s' % name + '  '.join('%s %.0f' % (k.replace('SQ_INSTS_', ''), v / ch) for k, v in sorted(avg.items())))
PY
rm -rf gpurun_out/probe/kn
