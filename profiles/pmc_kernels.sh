#!/bin/bash
# Counter passes (one counter per pass) over `bench.py --pmc-child` with any environment / arguments, summarised per kernel.
#     CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY ..." bash profiles/pmc_kernels.sh <tag> <kernel name substring> [bench.py arguments]
set -u
tag=${1:-pmc}; pat=${2:-lfq}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in ${CTRS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU}; do
    out=$R/gpurun_out/pmck_${tag}/$ctr
    mkdir -p "$out"
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d "$out" -o pmc -- python "$R/bench.py" --pmc-child "$@" > "$out.log" 2>&1 || echo "pass $ctr failed ($?)"
done
cd "$R" && python - "$tag" "$pat" <<'PY'
import glob, os, sqlite3, sys
tag, pat = sys.argv[1], sys.argv[2]
rows = {}
for d in sorted(glob.glob("gpurun_out/pmck_%s/*/" % tag)):
    c = os.path.basename(d.rstrip("/"))
    dbs = glob.glob(d + "*.db") + glob.glob(d + "*/*.db")
    if not dbs:
        continue
    con = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = ("select s.display_name, count(distinct d.id), sum(p.value) from %s p join %s d on p.event_id=d.event_id "
         "join %s s on d.kernel_id=s.id group by 1" % (pmc, kd, ks))
    for name, n, v in con.execute(q):
        if pat in name:
            rows.setdefault(name.split("(")[0].replace("void ", ""), {})[c] = v / max(n, 1)
for k, cs in rows.items():
    print("## %s (per launch)" % k)
    for c, v in sorted(cs.items()):
        print("| %s | %.4g |" % (c, v))
PY
