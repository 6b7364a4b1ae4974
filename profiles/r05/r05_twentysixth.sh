# Round 5: lean count kernel with the end chunks' and the remainder's loads ahead of the loop (two round trips fewer per wavefront)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_lean_hoist.txt; : > $out
for rep in 1 2; do for u in 2 3; do
  echo "== LFQ_COUNT_AHEAD_DEEP=$u (round $rep)" >> $out
  LFQ_COUNT_AHEAD_DEEP=$u python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], 'count', k['ms_count'], 'dp', k['ms_dp'], d['roofline'].get('kernel_alone'), c['pipeline'][c['pipeline'].find('chosen'):])" >> $out
done; done
python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
rm -f lofreq_amd/csrc/build/lfq_kernels.o
make -C lofreq_amd/csrc EXTRA=-DLFQ_COUNT_STAMP 2>&1 | grep -i "error" >> $out
python profiles/wave_stamps.py 2>&1 | grep -v amdgpu.ids >> $out
cat $out
