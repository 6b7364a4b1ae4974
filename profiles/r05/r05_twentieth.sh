# Round 5: lfq_dp_combine_kernel / lfq_dp_big_prep_kernel at <= 168 registers (they fit into what one retiring count workgroup frees)
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
out=gpurun_out/r05_dp_footprint.txt; : > $out
for cfg in "" "--config C2" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['workload'][:50], d['ms_per_step'], d['repeats']['ms_per_step_median'], c['pipeline'][c['pipeline'].find('chosen'):]); print(c['kernel_ms'])" >> $out
done
cat $out
for cfg in C3; do for gate in none; do
  o=$R/gpurun_out/prof_ov2_${cfg}_$gate; rm -rf $o; mkdir -p $o
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $o -o t -- python $R/bench.py --config $cfg --in-flight 4 --gate $gate --steps 16 --warmup 4 --repeats 1 --no-cpu-baseline --no-pmc --no-secondary --no-full-check > $o/bench.log 2>&1)
  { echo "# $cfg, four batches queued, gate $gate"; python profiles/overlap_timeline.py $o 8; } > gpurun_out/r05_overlap2_${cfg}_$gate.txt 2>&1
done; done
cat gpurun_out/r05_overlap2_C3_none.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
