#!/usr/bin/env python
"""Where a step's time goes between batches, from a rocprofv3 --kernel-trace database of a bench.py run: for every pair of
consecutive count-kernel launches the period (start to start), the count kernel's duration, the span of the batch's other
kernels, and the idle time on the device between the batch's last kernel (copies and fills included) and the next count
kernel.  Prints the median / max over the last `n` steps and the first few steps' numbers.
    python profiles/step_gaps.py <dir with the .db> [n = 20]"""
import glob
import sqlite3
import statistics
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 20
db = sorted(glob.glob(d + "/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "lfq_count_" in r[0]]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    t0, t1 = rows[a][1], rows[b][1]
    inner = rows[a + 1:b]
    last_end = max([rows[a][2]] + [r[2] for r in inner])
    # what sits between the previous batch's kernels and this count kernel on the device: the fills (memsets) right in
    # front of the next count kernel belong to the next batch
    fills_front = 0
    k = len(inner) - 1
    while k >= 0 and "fillBuffer" in inner[k][0]:
        fills_front += 1
        k -= 1
    last_own = max([rows[a][2]] + [r[2] for r in inner[:k + 1]])
    steps.append(dict(period=(t1 - t0) / 1e6, count=(rows[a][2] - rows[a][1]) / 1e6, tail=(last_own - rows[a][2]) / 1e6,
                      idle=(t1 - last_end) / 1e6, gap_to_count=(t1 - last_own) / 1e6, fills=fills_front))
last = steps[-n_last:]
print("steps in trace: %d; over the last %d:" % (len(steps), len(last)))
for key in ("period", "count", "tail", "gap_to_count", "idle"):
    v = [s[key] for s in last]
    print("  %-13s median %.3f  min %.3f  max %.3f ms" % (key, statistics.median(v), min(v), max(v)))
print("  (period = count start to next count start; tail = count end -> the batch's last kernel; gap_to_count = that kernel's"
      " end -> next count start, memsets in between included; idle = nothing at all running)")
for s in last[:6]:
    print("   period %.3f  count %.3f  tail %.3f  gap_to_count %.3f  idle %.3f  fills %d" % (
        s["period"], s["count"], s["tail"], s["gap_to_count"], s["idle"], s["fills"]))
