# Round 5, lean count kernel: non-temporal track loads (the guide's weight stream gains 2-6 % from them)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_lean_nt.txt; : > $out
for rep in 1 2; do for nt in 0 1; do
  echo "== LFQ_COUNT_NT_LOADS=$nt (round $rep)" >> $out
  LFQ_COUNT_NT_LOADS=$nt python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], c['pipeline'][c['pipeline'].find('chosen'):]); print(c['kernel_ms']); print(d['roofline'].get('kernel_alone'))" >> $out
done; done
cat $out
