# Round 5, after the combine / prep footprint fix (DP chain of a batch ends 1.4 ms into the next count kernel): the lean count
# kernel's knobs again -- chunks in flight per lane x columns per workgroup
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_lean_knobs2.txt; : > $out
for rep in 1 2; do for u in 2 3 4; do for w in 16 8; do
  echo "== LFQ_COUNT_AHEAD_DEEP=$u LFQ_COUNT_WAVES_PER_WG=$w (round $rep)" >> $out
  LFQ_COUNT_AHEAD_DEEP=$u LFQ_COUNT_WAVES_PER_WG=$w python bench.py --in-flight 4 --gate none --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], 'count', k['ms_count'], 'dp', k['ms_dp'])" >> $out
done; done; done
cat $out
