# Round 5, shared-wavefront count kernel at eight wavefronts per SIMD (records over sums + headers in LDS: 16 KB per workgroup)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_shallow_occ.txt; : > $out
for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['workload'][:50], d['ms_per_step'], d['repeats']['ms_per_step_median'], c['pipeline'][c['pipeline'].find('chosen'):]); print(c['kernel_ms']); print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('kernel_alone'))" >> $out
done
cat $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py tests/test_gpu_knobs.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
