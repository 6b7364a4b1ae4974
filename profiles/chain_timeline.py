import sqlite3, sys, glob
db = sorted(glob.glob(sys.argv[1] + "/*.db") + glob.glob(sys.argv[1] + "/*/*.db"))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(con.execute("select s.display_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
# last iteration: from the last lfq_baq_reg_kernel group start
idx = [i for i, r in enumerate(rows) if "baq_reg" in r[0] or "lfq_baq_kernel" in r[0]]
# find start of last group: go back while gaps < 20 ms
i = idx[-1]
while i - 1 in idx or (i > 0 and rows[i][1] - rows[i - 1][2] < 2e6 and i - 1 >= idx[0] and any(j == i - 1 for j in idx)):
    i -= 1
start = idx[-1]
for j in reversed(idx):
    if rows[start][1] - rows[j][1] < 32e6:
        start = j
t0 = rows[start][1]
last = None
for name, a, b in rows[start:]:
    nm = name.split("(")[0].replace("void ", "")[:40]
    if "copyBuffer" in nm or "fillBuffer" in nm:
        if last and last[0] == nm:
            last[2] = b; last[3] += 1
            continue
        if last: print("%-42s %9.3f %9.3f x%d" % (last[0], (last[1]-t0)/1e6, (last[2]-t0)/1e6, last[3]))
        last = [nm, a, b, 1]
        continue
    if last: print("%-42s %9.3f %9.3f x%d" % (last[0], (last[1]-t0)/1e6, (last[2]-t0)/1e6, last[3])); last = None
    print("%-42s %9.3f %9.3f" % (nm, (a - t0) / 1e6, (b - t0) / 1e6))
if last: print("%-42s %9.3f %9.3f x%d" % (last[0], (last[1]-t0)/1e6, (last[2]-t0)/1e6, last[3]))
