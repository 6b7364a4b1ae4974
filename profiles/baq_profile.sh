#!/bin/bash
# Kernel-trace profile of `bench.py --mode baq` (400 K reads x 150 bp).  From the repo root on the GPU box:
#     bash profiles/baq_profile.sh <tag> [extra bench.py arguments, e.g. --idaq]
set -u
tag=${1:-baq}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag; mkdir -p "$out"
timeout 300 rocprofv3 --kernel-trace --stats -d "$out" -o trace -- python "$R/bench.py" --mode baq --steps 100 "$@" > "$out/bench.log" 2>&1
db=$(ls "$out"/*.db "$out"/*/*.db 2>/dev/null | tail -1)
{
  echo "# rocprofv3 --kernel-trace --stats: bench.py --mode baq --steps 100 $*"
  echo
  python "$R/profiles/summarize_rocprof.py" "$db" | head -12
  echo
  echo "bench line of the profiled run:"
  echo
  grep '^{' "$out/bench.log" | tail -1
} > "$R/gpurun_out/${tag}_rocprof_stats.md"
cat "$R/gpurun_out/${tag}_rocprof_stats.md"
