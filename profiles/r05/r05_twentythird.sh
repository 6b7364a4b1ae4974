# Round 5: count workgroups of 12 wavefronts (three per SIMD: two of them leave two slots and 320 registers of every SIMD to
# the DP kernels at all times) against 16 and 8; the 512-thread DP kernels at <= 168 registers
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_wg12.txt; : > $out
for rep in 1 2; do for w in 16 12 8; do for mode in "--in-flight 4 --gate none" "--in-flight 4 --gate end"; do
  echo "== LFQ_COUNT_WAVES_PER_WG=$w $mode (round $rep)" >> $out
  LFQ_COUNT_WAVES_PER_WG=$w python bench.py $mode --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], 'count', k['ms_count'], 'dp', k['ms_dp'])" >> $out
done; done; done
cat $out
