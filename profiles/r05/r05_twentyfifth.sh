# Round 5: C2 -- where the host's side of a step goes (LFQ_BENCH_TRACE_STEPS), and the 512-thread DP kernels at 168 registers
# (default) against 256 (LFQ_DP512_WAVES=2: as before the bound), same box
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_c2_host.txt; : > $out
LFQ_BENCH_TRACE_STEPS=1 python bench.py --config C2 --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2> gpurun_out/r05_c2_trace.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['ms_per_step'], c['pipeline'][c['pipeline'].find('chosen'):], c.get('host_ms_per_step_not_hidden'))" >> $out
grep "^\[step" gpurun_out/r05_c2_trace.err | tail -12 >> $out
for w in 3 2 3 2; do
  rm -f lofreq_amd/csrc/build/lfq_dp.o
  make -C lofreq_amd/csrc EXTRA=-DLFQ_DP512_WAVES=$w 2>&1 | grep -i "error" >> $out
  echo "== LFQ_DP512_WAVES=$w" >> $out
  for cfg in "--config C2" ""; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['workload'][:3], d['ms_per_step'], d['repeats']['ms_per_step_median'], c['pipeline'][c['pipeline'].find('chosen'):])" >> $out
  done
done
cat $out
