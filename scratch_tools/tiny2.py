import sys, os
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"tests"))
import numpy as np, torch
import lofreq_amd as la, util
which=sys.argv[1]
if which=="fe":
    cases = [(10000, (60, 40, 0)), (10000, (1000, 40, 0)), (1000, (100, 12, 0)), (1000, (300, 12, 0))]
    host = util.concat_batches([util.uniform_p_column(n, c) for n, c in cases])
    kw = dict(bonf_dynamic=0, bonf_subst=3000000, min_bq=0, min_alt_bq=0)
else:
    rng=np.random.default_rng(1)
    host=util.random_batch(rng, 400, 0, 300, planted={5:0.3, 9:0.05}, ref_n_frac=0.02)
    kw={}
c=la.SnvCaller(0); conf=la.VarcallConf(**kw)
recs,counts,st=c.call_snvs(util.to_pileup_batch(la,host), conf, want_counts=True)
print("ok", len(recs), st.n_tested, st.n_pvals)
