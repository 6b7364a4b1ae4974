# Round 5, shared-wavefront count kernel with the next step's loads requested before the current one is counted
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_shallow_prefetch.txt; : > $out
for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['workload'][:50], d['ms_per_step'], c['kernel_ms']['ms_count'], d['roofline']['kernel'], d['roofline']['frac'], (d['roofline'].get('kernel_alone') or {}).get('avg_launch_ms'), c['pipeline'][c['pipeline'].find('chosen'):])" >> $out
done
cat $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py tests/test_gpu_knobs.py tests/test_gpu_configs.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
