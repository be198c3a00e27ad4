mkdir -p gpurun_out/v5
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/v5/stats -o r01 -- $CMD > /root/repo/gpurun_out/v5/stats.log 2>&1
cd /root/repo
db=$(find gpurun_out/v5/stats -name '*.db' | head -1)
python tools/rocpd_summary.py stats $db gpurun_out/v5/stats_summary.csv
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/v5/stats/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute('select name, start, end from kernels order by start').fetchall()
# steady-state window: last 6 steps = find k_gather launches
g = [i for i, r in enumerate(rows) if 'k_gather' in r[0] or 'k_cycle_linear' in r[0]]   # one per step
i0, i1 = g[-7], g[-1]
busy = sum(r[2] - r[1] for r in rows[i0:i1])
span = rows[i1][1] - rows[i0][1]
print('steps', 6, 'span/step us', span / 6e3, 'busy/step us', busy / 6e3, 'kernels/step', (i1 - i0) / 6)
gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, rows[i][0][:40], rows[i + 1][0][:40]) for i in range(i0, i1))
print('largest gaps (us):')
for gp in gaps[-12:]:
    print(gp)
PY
rm -f $db
cat gpurun_out/v5/stats_summary.csv
