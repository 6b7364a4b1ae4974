import sys, os
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"tests"))
import numpy as np, torch
import lofreq_amd as la, util
rng=np.random.default_rng(1)
host=util.random_batch(rng, 40, 50, 300, planted={5:0.3, 9:0.05})
c=la.SnvCaller(0); conf=la.VarcallConf()
recs,counts,st=c.call_snvs(util.to_pileup_batch(la,host), conf, want_counts=True)
print("ok", len(recs), st.n_tested, st.n_pvals)
