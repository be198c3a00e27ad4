# kernel trace of the decomposed-domain step (loopback transport), with per-step summary
mkdir -p gpurun_out/loop
python tools/loopback_multirank.py --single > gpurun_out/loop/single.log 2>&1
python tools/loopback_multirank.py > gpurun_out/loop/loop.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/loop/stats -o r01 -- python /root/repo/tools/loopback_multirank.py --steps 28 --warmup 2 > /root/repo/gpurun_out/loop/stats.log 2>&1
cd /root/repo
db=$(find gpurun_out/loop/stats -name '*.db' | head -1)
python tools/rocpd_summary.py stats $db gpurun_out/loop/stats_summary.csv > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/loop/stats/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute('select name, start, end from kernels order by start').fetchall()
g = [i for i, r in enumerate(rows) if 'k_gather' in r[0] or 'k_cycle_linear' in r[0]]   # one per step
# steps 16..27 of the timed call: between two particle exchanges
i0, i1 = g[-11], g[-1]
busy = sum(r[2] - r[1] for r in rows[i0:i1])
span = rows[i1][1] - rows[i0][1]
print('steps', 10, 'span/step us', span / 10e3, 'busy/step us', busy / 10e3, 'kernels/step', (i1 - i0) / 10)
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.])
for r in rows[i0:i1]:
    acc[r[0][:70]][0] += 1
    acc[r[0][:70]][1] += (r[2] - r[1]) / 1e3
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-72s %5.1f/step %8.1f us/step' % (k, v[0] / 10, v[1] / 10))
gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, rows[i][0][:40], rows[i + 1][0][:40]) for i in range(i0, i1))
print('largest gaps (us):')
for gp in gaps[-10:]:
    print(gp)
PY
rm -f $db
cat gpurun_out/loop/single.log gpurun_out/loop/loop.log | grep -v amdgpu.ids
