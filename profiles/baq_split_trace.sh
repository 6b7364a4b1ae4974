# kernel durations of the split BAQ launches (rocprofv3 --kernel-trace --stats), 400 K x 150 bp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cat > /tmp/baq_snip.py <<PY
import sys
sys.path.insert(0, "$R")
import bench, lofreq_amd as la, numpy as np, torch
c = la.SnvCaller(0)
Rr = bench.make_reads(400000, 2000000, indel_frac=0.0)
rs = la.ReadSet.from_arrays(c, Rr)
for _ in range(4):
    rs.baq(extended=True, idaq=False)
torch.cuda.synchronize()
rs.close(); c.close()
PY
for v in "$@"; do
  out=$R/gpurun_out/prof_baqsplit; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && env $v timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t -- python /tmp/baq_snip.py > $out/log 2>&1)
  echo "== $v"
  python - <<PY
import glob, sqlite3
db = sorted(glob.glob("$out/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
baq = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in rows if "lfq_baq" in n]
import collections
agg = collections.defaultdict(list)
for n, s, e in baq: agg[n].append((e - s) / 1e6)
for n, v in agg.items(): print("   %-48s calls %3d  avg %.3f ms  total %.2f" % (n[:48], len(v), sum(v) / len(v), sum(v)))
# the last call: first to last BAQ kernel
k = len(baq) // 4
last = baq[-k:]
print("   last call: %d kernels, span %.3f ms" % (len(last), (max(e for _, _, e in last) - min(s for _, s, _ in last)) / 1e6))
for n, s, e in last[:16]: print("      %-44s start %.3f dur %.3f" % (n[:44], (s - last[0][1]) / 1e6, (e - s) / 1e6))
PY
done 2>&1 | grep -v amdgpu.ids
