# Round 5: row segments per split column, mid class (first stretch included: 2 = the rest in one piece) x big class, four
# batches queued without a gate
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = shape args, $2 = mode args; ENVV = env
  env $ENVV python bench.py $1 $2 --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-12s %-40s step %.3f (min %.3f max %.3f)  count %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$1', '$ENVV', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))" || tail -3 gpurun_out/r05_x.err
}
for sh in "--config C3" "--config C2"; do
for mid in 8 2 3 4; do
for big in 8 2 3 4; do
ENVV="LFQ_SEG_MAX_MID=$mid LFQ_SEG_MAX_BIG=$big" one "$sh" "--in-flight 4 --gate none"
done
done
ENVV="LFQ_SEG_MAX_MID=8 LFQ_SEG_MAX_BIG=8" one "$sh" "--in-flight 4 --gate none"
done
