# BAQ register kernel as two launches (forward pass / sweep): parity, then kernel time per 400 K x 150 bp reads
# usage: bash profiles/ab_baq_split.sh "ENV1=.. ENV2=.." "ENV.." ...   (each argument = one variant's environment; "X=0" = default)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  env $v python -m pytest tests/test_gpu_baq.py -x -q -p no:cacheprovider 2>&1 | tail -1
  for rep in 1 2; do
  env $v python - <<PY
import sys
sys.path.insert(0, ".")
import bench, lofreq_amd as la, numpy as np, torch
c = la.SnvCaller(0)
R = bench.make_reads(400000, 2000000, indel_frac=0.0)
rs = la.ReadSet.from_arrays(c, R)
rs.baq(extended=True, idaq=False); torch.cuda.synchronize()
ms = []
for _ in range(6):
    rs.baq(extended=True, idaq=False); ms.append(c.baq_times()["ms_kernels"])
print("   kernels %.3f ms (min %.3f) per 400 K reads, %d launches" % (float(np.median(ms)), min(ms), c.baq_times()["n_launches"]))
rs.close(); c.close()
PY
  done
done 2>&1 | grep -v amdgpu.ids
