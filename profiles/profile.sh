#!/bin/bash
# Kernel-trace profile of the bench step on the GPU box.  From the repo root:
#     bash profiles/profile.sh <tag> [bench.py arguments]
# writes gpurun_out/<tag>_rocprof_stats.md (per-kernel stats + the timeline of one step); copy what is to be kept
# into profiles/.  Counter passes: profiles/run_pmc.sh.
set -u
tag=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out" -o trace -- \
    python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary "$@" > "$out/bench.log" 2>&1
db=$(ls "$out"/*.db "$out"/*/*.db 2>/dev/null | tail -1)
md=$R/gpurun_out/${tag}_rocprof_stats.md
{
  echo "# rocprofv3 --kernel-trace --stats: bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary $*"
  echo
  python "$R/profiles/summarize_rocprof.py" "$db"
  echo
  echo "## Timeline of the last step (ms from the start of its count kernel)"
  echo
  echo '```'
  python "$R/profiles/timeline.py" "$db"
  echo '```'
  echo
  echo "bench line of the profiled run:"
  echo
  grep '^{' "$out/bench.log" | tail -1
} > "$md"
cat "$md"
