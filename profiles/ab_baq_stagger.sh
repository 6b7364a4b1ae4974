# BAQ register kernel: start phases of a launch's first round (LFQ_BAQ_STAGGER_US x LFQ_BAQ_STAGGER_PHASES), 400 K x 150 bp reads
cd $GRAFT_REPO_ROOT
LFQ_BAQ_STAGGER_US=300 python -m pytest tests/test_gpu_baq.py -x -q -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for v in "0 2" "100 2" "200 2" "300 2" "400 2" "500 2" "100 4" "150 4" "200 4" "80 8"; do
  set -- $v
  LFQ_BAQ_STAGGER_US=$1 LFQ_BAQ_STAGGER_PHASES=$2 python - <<PY
import sys, time
sys.path.insert(0, ".")
import bench, lofreq_amd as la, numpy as np, torch
c = la.SnvCaller(0)
R = bench.make_reads(400000, 2000000, indel_frac=0.0)
rs = la.ReadSet.from_arrays(c, R)
rs.baq(extended=True, idaq=False); torch.cuda.synchronize()
ms = []
for _ in range(6):
    rs.baq(extended=True, idaq=False); ms.append(c.baq_times()["ms_kernels"])
print("stagger %4s us x %s phases: kernels %.3f ms (min %.3f) per 400 K reads" % ("$1", "$2", float(np.median(ms)), min(ms)))
rs.close(); c.close()
PY
done
done
