#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd SQLite output) runs into small CSV files that can
be committed under profiles/:
  kernel stats  : rocprofv3 --kernel-trace --stats -d DIR -o NAME -- <cmd>
  PMC counters  : rocprofv3 --pmc FETCH_SIZE -d DIR ... (one counter set per run)
usage: rocpd_summary.py stats  <results.db> <out.csv>
       rocpd_summary.py pmc    <results.db> <out.csv>
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)            # drop the argument list
    name = re.sub(r'^void ', '', name)
    if 'rocprim' in name:
        m = re.search(r'(radix_sort_onesweep_\w+|radix_sort\w*)', name)
        return 'rocprim::' + (m.group(1) if m else 'kernel')
    return name[:110]


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name, count(*), sum(duration), avg(duration), min(duration), '
                       'max(duration) from kernels group by name order by sum(duration) desc').fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'percent'])
        for n, c, s, a, mi, ma in rows:
            w.writerow([short(n), c, '%.1f' % (s / 1e3), '%.2f' % (a / 1e3), '%.2f' % (mi / 1e3),
                        '%.2f' % (ma / 1e3), '%.2f' % (100. * s / tot)])


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select kernel_name, counter_name, count(*), avg(value), avg(duration) '
                       'from counters_collection group by kernel_name, counter_name '
                       'order by sum(value) desc').fetchall()
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'counter', 'dispatches', 'avg_value', 'avg_duration_us'])
        for n, cn, c, v, d in rows:
            w.writerow([short(n), cn, c, '%.3f' % v, '%.2f' % (d / 1e3)])


def pmcseq(db, out, pattern='k_cycle_linear'):
    """Counter values per dispatch, in dispatch order, of the kernels whose name holds `pattern`."""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')").fetchall()]
    key = next((c for c in ('dispatch_id', 'start', 'id') if c in cols), 'rowid')
    rows = cur.execute('select %s, kernel_name, counter_name, value from counters_collection '
                       'where kernel_name like ? order by %s' % (key, key), ('%' + pattern + '%',)).fetchall()
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['dispatch', 'kernel', 'counter', 'value'])
        for k, n, cn, v in rows:
            w.writerow([k, short(n)[:60], cn, '%.0f' % v])


if __name__ == '__main__':
    if sys.argv[1] == 'pmcseq':
        pmcseq(*sys.argv[2:])
        sys.exit(0)
    {'stats': stats, 'pmc': pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
