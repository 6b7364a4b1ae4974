"""per-launch counter values of the BAQ kernels from the passes of profiles/baq_pmc.sh"""
import glob
import os
import sqlite3
import sys

tag = sys.argv[1]
rows = {}
for d in sorted(glob.glob("gpurun_out/pmc_%s/*/" % tag)):
    c = os.path.basename(d.rstrip("/"))
    dbs = glob.glob(d + "*.db") + glob.glob(d + "*/*.db")
    if not dbs:
        continue
    con = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = ("select s.display_name, count(distinct d.id), sum(p.value) from %s p join %s d on p.event_id=d.event_id "
         "join %s s on d.kernel_id=s.id group by 1" % (pmc, kd, ks))
    for name, n, v in con.execute(q):
        if "baq" in name:
            rows.setdefault(name.split("(")[0], {})[c] = v / max(n, 1)
for k, cs in rows.items():
    print("## %s (per launch)" % k)
    print()
    print("| counter | value |")
    print("|---|---|")
    for c, v in sorted(cs.items()):
        print("| %s | %.4g |" % (c, v))
