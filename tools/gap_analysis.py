"""Inter-kernel gaps of the evaluation loop from a rocprofv3 --kernel-trace CSV.
usage: python tools/gap_analysis.py <dir containing *kernel_trace.csv>
For every kernel name: average duration and the average idle time between the previous kernel's end and its start
(same queue, steady state = the last 60 % of the dispatches)."""
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
rows = rows[int(len(rows) * 0.4):]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for s, e, n in rows:
    n = n.split("(")[0][-48:]
    dur[n] += e - s
    if prev_end is not None:
        gap[n] += s - prev_end
    cnt[n] += 1
    prev_end = e
print("%-50s %8s %10s %12s" % ("kernel", "calls", "avg_us", "gap_before_us"))
for n in sorted(cnt, key=lambda k: -cnt[k]):
    print("%-50s %8d %10.2f %12.2f" % (n, cnt[n], dur[n] / cnt[n] / 1e3, gap[n] / cnt[n] / 1e3))
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print("span %.1f us, kernels busy %.1f us (%.1f %%)" % (span / 1e3, busy / 1e3, 100.0 * busy / span))
