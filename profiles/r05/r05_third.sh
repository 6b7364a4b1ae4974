set -u
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for mode in "2 end" "1 tail"; do
    set -- $mode
    echo "== run $i in-flight $1 gate $2"
    LFQ_BENCH_TRACE_STEPS=1 python bench.py --steps 20 --warmup 5 --in-flight $1 --gate $2 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2> gpurun_out/trace.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('first %.3f min %.3f med %.3f max %.3f  kernels %.3f  host_not_hidden %s' % (r['ms_per_step_first'], r['ms_per_step_min'], r['ms_per_step_median'], r['ms_per_step_max'], c['ms_kernels'], c['host_ms_per_step_not_hidden']))"
    grep "^\[step" gpurun_out/trace.err | awk '{ if ($9 > 3.6 || $5 > 1.0 || $7 > 1.0) print }' | head -40
    grep "^\[step" gpurun_out/trace.err | head -5
  done
done > gpurun_out/r05_step_trace.txt 2>&1
cat gpurun_out/r05_step_trace.txt
