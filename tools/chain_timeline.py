"""Timeline of the LAST front-end solve in a rocprofv3 --kernel-trace --output-format csv run of tools/chain_ab.py: every kernel's
duration and the idle gap in front of it, split by whether the gather ran (> 4 us) or returned at its gate.
usage: python tools/chain_timeline.py <dir with *kernel_trace.csv> [n_last_kernels]"""
import csv
import glob
import os
import sys
from collections import defaultdict

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
short = lambda n: n.split("(")[0].replace("void cmx::", "")[:40]
cls = defaultdict(lambda: [0, 0.0, 0.0])
prev = None
for s, e, n in rows[len(rows) // 2:]:
    k = short(n)
    d = (e - s) / 1e3
    if "fe_gather" in k:
        k += " [ran]" if d > 4.0 else " [gated off]"
    c = cls[k]
    c[0] += 1
    c[1] += d
    if prev is not None:
        c[2] += (s - prev) / 1e3
    prev = e
print("%-52s %7s %9s %10s" % ("kernel (second half of the run)", "calls", "avg_us", "gap_before"))
for k in sorted(cls, key=lambda k: -cls[k][0]):
    c = cls[k]
    print("%-52s %7d %9.2f %10.2f" % (k, c[0], c[1] / c[0], c[2] / c[0]))
print("\nlast %d kernels:" % n_last)
prev = None
for s, e, n in rows[-n_last:]:
    print("  %-44s %7.2f us   gap %6.2f" % (short(n), (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
