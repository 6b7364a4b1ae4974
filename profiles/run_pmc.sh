#!/bin/bash
# Counter passes of the bench step (what bench.py spawns by itself for roofline.traffic / dp.valu_busy, kept here so
# the numbers can be reproduced by hand).  From the repo root on the GPU box:
#     bash profiles/run_pmc.sh <tag> [bench.py arguments]
# One counter per pass (FETCH_SIZE + WRITE_SIZE exceed the TCC slots of gfx950); never combined with --sys-trace etc.
set -u
tag=${1:-pmc}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export LFQ_SINGLE_STREAM=1
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    out=$R/gpurun_out/pmc_${tag}/pmc_$ctr
    mkdir -p "$out"
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d "$out" -o pmc -- \
        python "$R/bench.py" --pmc-child "$@" > "$out.log" 2>&1 || echo "pass $ctr failed ($?)"
done
cd "$R" && python profiles/summarize_pmc.py gpurun_out/pmc_${tag} $tag
