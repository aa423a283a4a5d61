"""Summarises rocprofv3 rocpd databases (kernel trace + PMC passes) as text.
usage: python tools/prof_summary.py gpurun_out/prof > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
for db in sorted(glob.glob(os.path.join(root, '*', '*_results.db'))):
    c = sqlite3.connect(db)
    tag = os.path.basename(os.path.dirname(db))
    print('== %s (%s)' % (tag, os.path.basename(db)))
    if tag.startswith('trace'):
        rows = c.execute(
            "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
            "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print('%-60s %8s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct'))
        for r in rows[:12]:
            print('%-60s %8d %14d %12.0f %12d %12d %6.2f%%' % (r[0][:60], r[1], r[2], r[3], r[4], r[5], 100. * r[2] / tot))
    else:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
        rows = c.execute(
            "select %s, counter_name, count(*), avg(value), sum(value) from counters_collection "
            "group by 1, 2 order by 1, 2" % name_col).fetchall()
        print('%-50s %-22s %8s %18s' % ('kernel', 'counter', 'launches', 'avg_per_launch'))
        for r in rows:
            if 'moog' in r[0]:
                print('%-50s %-22s %8d %18.1f' % (r[0][:50], r[1], r[2], r[3]))
