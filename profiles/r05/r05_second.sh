set -u
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for mode in "1 tail" "2 end"; do
  set -- $mode
  out=$R/gpurun_out/prof_r05_gap_$1$2; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $out -o t -- python $R/bench.py --steps 40 --warmup 5 --repeats 1 --in-flight $1 --gate $2 --no-cpu-baseline --no-pmc --no-secondary --no-full-check > $out/bench.log 2>&1)
  echo "== in-flight $1 gate $2"; tail -1 $out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['kernel_ms'])"
  python profiles/step_gaps.py $out 30
done > gpurun_out/r05_gaps.txt 2>&1
cat gpurun_out/r05_gaps.txt
# noise: the same two modes alternating, five times each, driver-like flags
for i in 1 2 3 4 5; do
  for mode in "1 tail" "2 end"; do
    set -- $mode
    python bench.py --steps 20 --warmup 5 --in-flight $1 --gate $2 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('in-flight $1 gate $2: first %.3f min %.3f med %.3f max %.3f  kernels %.3f  host_not_hidden %s' % (r['ms_per_step_first'], r['ms_per_step_min'], r['ms_per_step_median'], r['ms_per_step_max'], c['ms_kernels'], c['host_ms_per_step_not_hidden']))"
  done
done > gpurun_out/r05_noise.txt 2>&1
cat gpurun_out/r05_noise.txt
