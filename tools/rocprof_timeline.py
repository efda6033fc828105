#!/usr/bin/env python3
"""Time line of the kernels of ONE step from a rocprofv3 kernel-trace database: for every run of consecutive launches between two launches of
the step's first kernel, prints each kernel's start offset, duration and the idle gap in front of it (us), then busy / idle totals.
    rocprof_timeline.py <db> [which step, default the last complete one] [substring filter for the printed lines]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
c = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in c else "kernel_name"
rows = db.execute(f"select {name}, start, end from kernels order by start").fetchall()
def short(n):
    m = re.search(r"(k_\w+(?:<\d+>)?)", n)
    return m.group(1) if m else n[:40]
starts = [i for i, r in enumerate(rows) if "k_encode_etc1s_blocks" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
flt = sys.argv[3] if len(sys.argv) > 3 else ""
a, b = starts[which], starts[which + 1]
t0 = rows[a][1]; prev_end = t0; busy = 0; idle = 0
for n, s, e in rows[a:b]:
    gap = max(0, s - prev_end); idle += gap; busy += e - s
    if flt in n:
        print(f"{(s - t0) / 1e3:9.1f} +{gap / 1e3:6.1f} {(e - s) / 1e3:8.1f}  {short(n)}")
    prev_end = max(prev_end, e)
print(f"step: {(prev_end - t0) / 1e3:.0f} us, kernels busy {busy / 1e3:.0f} us, idle between kernels {idle / 1e3:.0f} us, {b - a} launches")
