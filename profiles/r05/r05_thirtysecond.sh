# Round 5: lean count kernel, 1 / 2 / 4 columns per wavefront with all their headers requested at the start
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_lean_cpw.txt; : > $out
for rep in 1 2; do for cp in 1 2 4; do
  echo "== LFQ_COUNT_COLS_PER_WAVE=$cp (round $rep)" >> $out
  LFQ_COUNT_COLS_PER_WAVE=$cp python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], 'count', k['ms_count'], 'dp', k['ms_dp'], (d['roofline'].get('kernel_alone') or {}).get('avg_launch_ms'), c['pipeline'][c['pipeline'].find('chosen'):])" >> $out
done; done
LFQ_COUNT_COLS_PER_WAVE=4 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -3 >> $out
cat $out
