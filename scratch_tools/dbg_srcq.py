import sys, json
sys.path.insert(0, "tests")
import numpy as np
import golden_util as gu
import lofreq_amd as la
c = la.SnvCaller(0)
fx, reads, nmq, ign = gu.load_srcq(gu.srcq_fixtures()[0])
ref = fx["genome"].encode()
for n in (1, 2, 8, 32, 220):
    for i0 in range(0, 220, n):
        sub = reads[i0:i0 + n]
        print("reads", i0, len(sub), [r["cigar"] for r in sub][:2], flush=True)
        sq, sqb = la.source_qual_batch(c, sub, ref, def_nm_q=nmq, min_bq=6, ign=ign)
        print(sq.tolist(), flush=True)
