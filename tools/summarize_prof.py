"""Summarise rocprofv3 (rocpd sqlite) output directories -- kernel stats and PMC counters -- into a text report.
usage: python tools/summarize_prof.py <dir with */*_results.db>"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
dbs = sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True))
for db in dbs:
    con = sqlite3.connect(db)
    rel = os.path.relpath(db, root)
    try:
        rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    except sqlite3.Error:
        rows = []
    pmc = []
    try:
        pmc = con.execute("select kernel_name, counter_name, avg(value), min(value), max(value), count(*), avg(duration) "
                          "from counters_collection group by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        pass
    if pmc:
        print("== %s : PMC counters, per-dispatch ==" % rel)
        print("%-58s %-16s %16s %16s %16s %6s %10s" % ("kernel", "counter", "avg", "min", "max", "n", "avg_us"))
        for k, c, a, lo, hi, n, d in pmc:
            print("%-58s %-16s %16.1f %16.1f %16.1f %6d %10.1f" % (k[:58], c, a, lo, hi, n, (d or 0) / 1e3))
    elif rows:
        print("== %s : kernel stats (rocprofv3 --kernel-trace --stats) ==" % rel)
        print("%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            print("%-70s %8d %14.1f %12.2f %7.2f" % (name[:70], calls, tot, avg, pct))
    print()
