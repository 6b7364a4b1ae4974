# Round 5, lean count kernel: chunks in flight per lane x columns per workgroup, four batches queued (what the warm-up picks)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_lean_knobs.txt; : > $out
for u in 2 3 4; do for w in 8 16; do
  echo "== LFQ_COUNT_AHEAD_DEEP=$u LFQ_COUNT_WAVES_PER_WG=$w" >> $out
  LFQ_COUNT_AHEAD_DEEP=$u LFQ_COUNT_WAVES_PER_WG=$w python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], c['pipeline'][c['pipeline'].find('chosen'):]); print(c['kernel_ms']); print(d['roofline'].get('kernel_alone'))" >> $out
done; done
cat $out
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 > gpurun_out/r05_lean_gpu_tests.txt
cat gpurun_out/r05_lean_gpu_tests.txt
