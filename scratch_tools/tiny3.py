import sys, os
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"tests"))
import numpy as np, torch
from lofreq_amd import _lib
if os.environ.get("LFQ_TRACE_LIB"): _lib.LIB_PATH=os.path.join(root,"scratch_tools","liblofreq_amd_%s.so" % os.environ["LFQ_TRACE_LIB"])
import lofreq_amd as la, util
n=int(sys.argv[1]); c=tuple(int(x) for x in sys.argv[2].split(","))
host = util.uniform_p_column(n, c)
kw = dict(bonf_dynamic=0, bonf_subst=3000000, min_bq=0, min_alt_bq=0)
cl=la.SnvCaller(0); conf=la.VarcallConf(**kw)
recs,counts,st=cl.call_snvs(util.to_pileup_batch(la,host), conf, want_counts=True)
print("ok", len(recs), st.n_tested, st.n_pvals, counts["kmax"])
