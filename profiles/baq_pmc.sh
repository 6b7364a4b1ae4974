#!/bin/bash
# Counter passes over `bench.py --mode baq` (one counter per pass).  From the repo root on the GPU box:
#     bash profiles/baq_pmc.sh <tag> [library file under lofreq_amd/]
set -u
tag=${1:-baq}; lib=${2:-liblofreq_amd.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export LFQ_AMD_LIB=$R/lofreq_amd/$lib
for ctr in ${CTRS:-FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE}; do
    out=$R/gpurun_out/pmc_${tag}/$ctr
    mkdir -p "$out"
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d "$out" -o pmc -- \
        python "$R/bench.py" --mode baq --steps 20 > "$out.log" 2>&1 || echo "pass $ctr failed ($?)"
done
cd "$R" && python profiles/baq_pmc_summary.py "$tag"
