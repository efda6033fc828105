#!/usr/bin/env python3
"""How well do several images in flight overlap on the GPU? From a rocprofv3 --kernel-trace database of `bench.py --streams N`:
  * per hardware queue / stream: launches, busy time, share of the traced window
  * concurrency histogram: for how long were exactly k kernels executing (k = 0 is an idle GPU)
  * the longest idle windows, with the kernels either side
  * (optional) the HIP API side, when the run also had --hip-trace: per API name calls / total / mean, so that blocking calls show up
    rocprof_concurrency.py <db> [t_from_ms t_to_ms]     (window relative to the first kernel, or to the END of the trace when negative; default = the whole trace)"""
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", n)
    return m.group(1)[:44] if m else n[:44]


db = sqlite3.connect(sys.argv[1])
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("kernels columns:", cols)
name = "name" if "name" in cols else "kernel_name"
qcol = [c for c in ("queue_id", "stream_id", "queue", "stream", "tid") if c in cols]
sel = ", ".join([name, "start", "end"] + qcol)
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
if not rows:
    sys.exit("no kernels")
t0 = rows[0][1]
t_end = max(r[2] for r in rows) - t0
lo = float(sys.argv[2]) * 1e6 if len(sys.argv) > 3 else 0
hi = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else t_end + 1
if lo < 0 or hi <= 0:   # negative = relative to the end of the trace
    lo, hi = t_end + lo, t_end + hi
rows = [r for r in rows if lo <= r[1] - t0 < hi]
w0, w1 = rows[0][1], max(r[2] for r in rows)
span = w1 - w0
print(f"window {span / 1e6:.2f} ms, {len(rows)} kernel launches")
for qi, qn in enumerate(qcol):
    acc = {}
    for r in rows:
        a = acc.setdefault(r[3 + qi], [0, 0])
        a[0] += 1
        a[1] += r[2] - r[1]
    print(f"-- by {qn}")
    for k, (n, busy) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"   {qn} {k}: {n:6d} launches, busy {busy / 1e6:8.2f} ms = {100.0 * busy / span:5.1f} % of the window")
# concurrency histogram by sweep
ev = []
for r in rows:
    ev.append((r[1], 1))
    ev.append((r[2], -1))
ev.sort()
hist = {}
k = 0
prev = ev[0][0]
idle = []
for t, d in ev:
    if t > prev:
        hist[k] = hist.get(k, 0) + (t - prev)
        if k == 0:
            idle.append((t - prev, prev))
    k += d
    prev = t
print("-- kernels executing at once: time (ms), share")
for kk in sorted(hist):
    print(f"   {kk:3d}: {hist[kk] / 1e6:9.2f} ms {100.0 * hist[kk] / span:5.1f} %")
# work-weighted: kernel-seconds / window = average number of kernels in flight
print(f"   average kernels in flight: {sum(r[2] - r[1] for r in rows) / span:.2f}")
idle.sort(reverse=True)
print(f"-- idle GPU: {sum(i[0] for i in idle) / 1e6:.2f} ms in {len(idle)} windows; windows > 20 us: {sum(1 for i in idle if i[0] > 20000)} "
      f"({sum(i[0] for i in idle if i[0] > 20000) / 1e6:.2f} ms); the 12 longest:")
ends = sorted(rows, key=lambda r: r[2])
for d, at in idle[:12]:
    before = [r for r in ends if r[2] == at]
    after = [r for r in rows if r[1] == at + d]
    print(f"   {d / 1e3:8.1f} us at {(at - w0) / 1e6:8.2f} ms  after {short(before[0][0]) if before else '?'} -> before {short(after[0][0]) if after else '?'}")
# per kernel: how much slower than alone? (mean duration; compare with the single-image stats file)
acc = {}
for r in rows:
    a = acc.setdefault(short(r[0]), [0, 0])
    a[0] += 1
    a[1] += r[2] - r[1]
print("-- kernels by total time (top 25): calls, total ms, mean us")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"   {n:44s} {c:6d} {t / 1e6:9.2f} {t / c / 1e3:9.1f}")
for v in ("regions", "hip_api", "regions_and_samples"):
    if v in views:
        c = [r[1] for r in db.execute(f"pragma table_info({v})")]
        if "name" in c and "start" in c and "end" in c:
            print(f"-- {v}: host API calls by total time (top 25): calls, total ms, mean us, max us")
            q = f"select name, count(*), sum(end-start), avg(end-start), max(end-start) from {v} group by name order by 3 desc limit 25"
            for n, k2, t, a, m in db.execute(q):
                print(f"   {n:44s} {k2:7d} {t / 1e6:9.2f} {a / 1e3:9.1f} {m / 1e3:9.1f}")
            break
