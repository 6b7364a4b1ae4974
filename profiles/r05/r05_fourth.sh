set -u
cd $GRAFT_REPO_ROOT
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cpuset $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
summ() {
python - "$1" <<'PY'
import re, sys
seen = set(); worst = []
for ln in open(sys.argv[1]):
    m = re.match(r'\[step\s+(\d+)\] wait ([\d.]+)\s+finish ([\d.]+)\s+submit ([\d.]+)\s+sum ([\d.]+) ms\s+kernels ([\d.]+)', ln)
    if m:
        k = int(m.group(1)); w, f, s, t, kk = map(float, m.groups()[1:])
        if (k, t) in seen: continue
        seen.add((k, t)); worst.append((f, k))
worst.sort(reverse=True)
print('     finish() worst:', ' '.join('%.2f@%d' % x for x in worst[:6]), ' n>1ms:', sum(1 for x in worst if x[0] > 1.0), 'of', len(worst))
PY
}
for v in "X=0" "LFQ_BENCH_NO_GC=1" "LFQ_HOST_SPIN_US=0" "LFQ_HOST_LOOP_THREADS=1" "LFQ_NO_SB_PRECOMPUTE=1" "LFQ_HOST_LOOP_THREADS=1 LFQ_NO_SB_PRECOMPUTE=1 LFQ_BENCH_NO_GC=1"; do
  for i in 1 2 3; do
    env $v LFQ_BENCH_TRACE_STEPS=1 LFQ_TIMING=1 python bench.py --steps 20 --warmup 5 --repeats 5 --in-flight 2 --gate end --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2> gpurun_out/trace.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('$v: first %.3f min %.3f med %.3f max %.3f  kernels %.3f' % (r['ms_per_step_first'], r['ms_per_step_min'], r['ms_per_step_median'], r['ms_per_step_max'], c['ms_kernels']))"
    summ gpurun_out/trace.err
    grep "lfq timing" gpurun_out/trace.err | awk '{ if ($5 > 1.0 || $7 > 1.0 || $9 > 1.0 || $11 > 1.5) print "     ", $0 }' | tail -4
  done
done
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
