#!/usr/bin/env python
"""How busy was the GPU?  From a rocprofv3 --kernel-trace database: the union of all kernel intervals over the last
`frac` of the trace (the warm-up is at the front), the idle gaps between them, and the kernels that fill the window.
    python profiles/gpu_busy.py <dir with the .db> [frac = 0.5]
(a value above 1 = the last that many MILLISECONDS of the trace: the timed steps of a run whose set-up is longer than they are)"""
import glob
import sqlite3
import sys
from collections import defaultdict

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
db = sorted(glob.glob(d + "/*.db") + glob.glob(d + "/*/*.db"))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(con.execute("select s.display_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
t_end = max(r[2] for r in rows)
t_beg = min(r[1] for r in rows)
w0 = t_end - (frac * (t_end - t_beg) if frac <= 1.0 else frac * 1e6)
rows = [r for r in rows if r[1] >= w0]
w0 = rows[0][1]
busy, cur_s, cur_e, gaps = 0, None, None, []
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
win = t_end - w0
per = defaultdict(lambda: [0, 0])
for n, s, e in rows:
    k = n.split("(")[0].replace("void ", "")
    per[k][0] += e - s
    per[k][1] += 1
print("window %.1f ms, GPU busy (union of kernels) %.1f ms = %.1f %%, %d kernels" % (win / 1e6, busy / 1e6, 100.0 * busy / win, len(rows)))
big = sorted(gaps, reverse=True)
print("idle: %.1f ms in %d gaps; > 1 ms: %d (%.1f ms), 0.1-1 ms: %d (%.1f ms), < 0.1 ms: %d (%.1f ms); largest %s ms" % (
    sum(gaps) / 1e6, len(gaps), sum(g > 1e6 for g in gaps), sum(g for g in gaps if g > 1e6) / 1e6,
    sum(1e5 < g <= 1e6 for g in gaps), sum(g for g in gaps if 1e5 < g <= 1e6) / 1e6,
    sum(g <= 1e5 for g in gaps), sum(g for g in gaps if g <= 1e5) / 1e6, [round(g / 1e6, 2) for g in big[:6]]))
print("| kernel | calls | total ms | % of window |")
print("|---|---|---|---|")
for k, (t, n) in sorted(per.items(), key=lambda x: -x[1][0])[:14]:
    print("| %s | %d | %.2f | %.1f |" % (k[:60], n, t / 1e6, 100.0 * t / win))
