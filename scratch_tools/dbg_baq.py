import sys, os
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"tests")); sys.path.insert(0, os.path.join(root,"oracle"))
import numpy as np
from lofreq_amd import _lib
_lib.LIB_PATH=os.path.join(root,"scratch_tools","liblofreq_amd_trace.so")
import lofreq_amd as la, pyoracle as orc
import test_gpu_baq as T
caller = la.SnvCaller(0)
rng = np.random.default_rng(5)
genome = "".join(rng.choice(list("ACGT"), 3000))
reads = T._random_reads(rng, genome, 400, 20, 160)
bad=[r for r in reads if r["pos0"]==1]
print(bad[0]["cigar"], len(bad[0]["seq"]))
out = la.baq_batch(caller, bad, genome.encode(), extended=True)
exp = orc.baq_read(bad[0]["pos0"], bad[0]["cigar"], bad[0]["seq"], bad[0]["qual"], genome.encode(), True)
print(out[0].tobytes()[:40]); print(exp.tobytes()[:40])
# oracle state
r=bad[0]
