import sys, os
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"tests")); sys.path.insert(0, os.path.join(root,"oracle"))
import numpy as np
from lofreq_amd import _lib
_lib.LIB_PATH=os.path.join(root,"scratch_tools","liblofreq_amd_trace.so")
import lofreq_amd as la, util, pyoracle as orc
caller = la.SnvCaller(0)
rng = np.random.default_rng(7)
dicts = util.random_indel_columns(rng, 80, 30, 900)
cols = la.IndelColumns.from_columns(dicts)
kw = dict(bonf_dynamic=0, bonf_indel=1, sig=1.0)
oconf = orc.default_conf(); conf = la.VarcallConf(**kw)
for k, v in kw.items(): setattr(oconf, k, v)
tests = orc.call_indels_batch(cols.flat(), oconf)
recs, ntests = la.call_indels(caller, cols, conf)
exp = tests[tests["emitted"] == 1]
got = set((int(r["col"]), int(r["side"]), int(r["event"])) for r in recs)
miss = [t for t in exp if (int(t["col"]), int(t["side"]), int(t["event"])) not in got]
print("missing", len(miss))
for t in miss[:40]:
    print("col %d side %d ev %d n=%d count=%d pv=%s" % (t["col"], t["side"], t["event"], t["n_err_probs"], t["count"], t["pvalue"]))
