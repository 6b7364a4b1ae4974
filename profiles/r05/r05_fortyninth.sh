# Round 5: the count kernel's shape knobs again now that a gate-none context's DP chains are lighter (two segments / none)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = shape args, $2 = mode args; ENVV = env
  env $ENVV python bench.py $1 $2 --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-12s %-60s step %.3f (min %.3f max %.3f)  count %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$1', '$ENVV', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))" || tail -3 gpurun_out/r05_x.err
}
for i in 1 2; do
for kv in "X=0" "LFQ_COUNT_WAVES_PER_WG=8" "LFQ_COUNT_WAVES_PER_WG=12" "LFQ_COUNT_AHEAD_DEEP=3" "LFQ_COUNT_AHEAD_DEEP=4" \
          "LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_AHEAD_DEEP=4" "LFQ_SCREEN_WAVES_PER_CU=6" "LFQ_SCREEN_WAVES_PER_CU=8" "LFQ_PHASE1_CHUNKS=16" "LFQ_PHASE1_CHUNKS=4"; do
ENVV="$kv" one "--config C3" "--in-flight 4 --gate none"
done
done
for kv in "X=0" "LFQ_COUNT_SHALLOW_WGS_NONE=3" "LFQ_COUNT_SHALLOW_WGS_NONE=4" "LFQ_COUNT_SHALLOW_WGS_NONE=1" "LFQ_PHASE1_CHUNKS=16" "LFQ_PHASE1_CHUNKS=4" "X=1"; do
ENVV="$kv" one "--config C2" "--in-flight 4 --gate none"
ENVV="$kv" one "--config C2" "--in-flight 3 --gate none"
done
