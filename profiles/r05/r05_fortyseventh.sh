# Round 5: row segments per split column (LFQ_SEG_MAX, 8 so far) under the queued form -- 2 and 3 against 8, alternating, every
# shape, with the gate the warm-up picks (auto) and with four queued batches without a gate
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = shape args, $2 = mode args; ENVV = env
  env $ENVV python bench.py $1 $2 --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-28s %-26s %-16s step %.3f (min %.3f max %.3f)  count %.3f  dp %.3f (l %.3f m %.3f b %.3f)  in flight %s %s  records %d' % (
    '$1', '$2', '$ENVV', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config'].get('batches_in_flight'), d['config'].get('batch_gate'), d['config']['records_per_step']))" || tail -3 gpurun_out/r05_x.err
}
for i in 1 2; do
for sh in "--config C3" "--config C2" "--depth 500 --cols 4600000" "--depth 200 --cols 3750000"; do
for kv in "LFQ_SEG_MAX=8" "LFQ_SEG_MAX=2" "LFQ_SEG_MAX=3"; do
ENVV="$kv" one "$sh" "--in-flight 4 --gate none"
done
done
done
for sh in "--config C3" "--config C2" "--depth 500 --cols 4600000" "--depth 200 --cols 3750000"; do
for kv in "LFQ_SEG_MAX=8" "LFQ_SEG_MAX=2"; do
ENVV="$kv" one "$sh" ""
ENVV="$kv" one "$sh" "--in-flight 4 --gate end"
done
done
