# usage: bash tools/prof_probe.sh <python script + args>   -> per (kernel, grid) device durations
mkdir -p gpurun_out/probe
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/probe/trace -o p -- python /root/repo/$@ > /root/repo/gpurun_out/probe/run.log 2>&1
cd /root/repo
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/probe/trace/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
gx = [c for c in cols if 'grid' in c.lower()]
sel = ', '.join(gx) if gx else "''"
rows = cur.execute('select name, %s, count(*), avg(duration), min(duration) from kernels group by name, %s order by name' % (sel, sel)).fetchall()
for r in rows:
    if 'fb::' in r[0]:
        print(r[0][:60], r[1:-3], 'n=%d avg=%.1f us min=%.1f us' % (r[-3], r[-2] / 1e3, r[-1] / 1e3))
PY
rm -rf gpurun_out/probe/trace
