import sys, os
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"tests"))
import numpy as np, torch
import lofreq_amd as la
import util
which = sys.argv[1]
caller = la.SnvCaller(0)
rng = np.random.default_rng(2)
if which == "t2":
    ncols=200
    planted = {c: af for c, af in zip(range(5, ncols, 37), [0.01, 0.03, 0.1, 0.3, 0.6, 0.02, 0.05, 0.9])}
    host = util.random_batch(rng, ncols, 900, 1100, planted=planted, ref_n_frac=0.02)
else:
    afs = [0.004, 0.008, 0.012, 0.02, 0.03, 0.05, 0.08, 0.12, 0.2, 0.24, 0.0, 0.0]
    planted = {c: af for c, af in enumerate(afs) if af > 0}
    host = util.random_batch(rng, len(afs), 6000, 10000, planted=planted)
conf = la.VarcallConf()
recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
print(which, os.environ.get("LFQ_DEBUG_SKIP"), "OK records", len(recs), "tested", st.n_tested, flush=True)
