# kernel trace around the first real particle hand-over of the loopback run
mkdir -p gpurun_out/fx
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/fx/t -o r -- python /root/repo/tools/first_exchange.py > /root/repo/gpurun_out/fx/log.txt 2>&1
cd /root/repo
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/fx/t/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith('kernels')][0] if any(t.startswith('kernels') for t in tabs) else None
rows = cur.execute('select name, start, end from kernels order by start').fetchall()
sc = [i for i, r in enumerate(rows) if 'k_scatter<false>' in r[0]]
print('k_scatter<false> launches at', sc[:6])
for which in (1, 2):
    if len(sc) <= which:
        continue
    i1 = sc[which]
    # back to the previous k_perm_deposit (end of the step before the hand-over)
    i0 = max(j for j in range(i1) if 'k_perm_deposit' in rows[j][0])
    t0 = rows[i0][2]
    print('--- hand-over %d: %d kernels, %.1f us from the end of the previous step to the re-sort' % (which, i1 - i0, (rows[i1][1] - t0) / 1e3))
    for j in range(i0 + 1, i1 + 1):
        n, s, e = rows[j]
        print('  +%8.1f us  %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, n[:90]))
PY
rm -rf gpurun_out/fx/t
