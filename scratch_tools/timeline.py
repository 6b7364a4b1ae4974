import sqlite3,glob,sys
db=sqlite3.connect(sys.argv[1] if sys.argv[1].endswith('.db') else sorted(glob.glob(sys.argv[1]+'/**/*.db', recursive=True))[-1])
rows=db.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'lfq_count_kernel' in r[0]]
i0=idx[-2]; t0=rows[i0][1]
for r in rows[i0:idx[-1]]:
    if 'rocclr' in r[0]: continue
    print("%-45s start %8.3f end %8.3f dur %7.3f"%(r[0][:45],(r[1]-t0)/1e6,(r[2]-t0)/1e6,(r[2]-r[1])/1e6))
