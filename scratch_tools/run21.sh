set -u
cd $GRAFT_REPO_ROOT
LFQ_COUNT_LPG4_BELOW=320 LFQ_TIMING=1 python bench.py --cols 3750000 --depth 200 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>&1 | grep "lfq timing" | tail -8
