# Round 5: what the lean count kernel needs of a CU.  Its workgroups per CU capped by an unused LDS request (one 1024-thread
# workgroup per CU = 4 wavefronts per SIMD instead of 8), 2 / 4 chunks in flight per lane; batch after batch (gate end: the
# count kernel alone on the machine) and four queued without a gate (beside the DP kernels of the batch before).
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_count_occupancy.txt; : > $out
for pad in 0 65536 40000; do for u in 2 4; do for mode in "--in-flight 4 --gate end" "--in-flight 4 --gate none"; do
  echo "== LFQ_COUNT_LDS_PAD=$pad LFQ_COUNT_AHEAD_DEEP=$u $mode" >> $out
  LFQ_COUNT_LDS_PAD=$pad LFQ_COUNT_AHEAD_DEEP=$u python bench.py $mode --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], c['kernel_ms']['ms_count'], c['kernel_ms']['ms_dp'])" >> $out
done; done; done
cat $out
