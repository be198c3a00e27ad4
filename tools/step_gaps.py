#!/usr/bin/env python3
"""Span against busy time per step from a rocprofv3 kernel trace (rocpd .db): the last `n` steps of the run, a step =
from one particle launch (k_cycle_linear / k_gather*) to the next.  usage: step_gaps.py <results.db> [n]"""
import sqlite3, sys
from collections import defaultdict
db = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cur = sqlite3.connect(db).cursor()
rows = cur.execute('select name, start, end from kernels order by start').fetchall()
g = [i for i, r in enumerate(rows) if 'k_gather' in r[0] or 'k_cycle_linear<' in r[0] and 'true>' not in r[0]]
i0, i1 = g[-n - 1], g[-1]
busy = sum(r[2] - r[1] for r in rows[i0:i1])
span = rows[i1][1] - rows[i0][1]
print('steps', n, 'span/step us %.1f busy/step us %.1f kernels/step %.1f' % (span / n / 1e3, busy / n / 1e3, (i1 - i0) / n))
acc = defaultdict(lambda: [0, 0., 0.])
for i in range(i0, i1):
    r = rows[i]
    a = acc[r[0][:70]]
    a[0] += 1; a[1] += (r[2] - r[1]) / 1e3; a[2] += (rows[i + 1][1] - r[2]) / 1e3
print('%-72s %9s %12s %14s' % ('kernel', 'per step', 'us per step', 'gap after, us'))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-72s %9.1f %12.1f %14.1f' % (k, v[0] / n, v[1] / n, v[2] / n))
