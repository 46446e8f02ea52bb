#!/usr/bin/env python3
"""Timeline of one iteration from a rocprofv3 kernel trace: python tools/kt_timeline.py <dir> <kernel-substring> [which]
prints the kernels between the `which`-th last (default 3rd last) and the following launch of the named kernel."""
import csv, glob, sys
d, key = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
import os
f = max(glob.glob(d + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
a, b = idx[-which], idx[-which + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:64]))
    prev = e
print("total %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
