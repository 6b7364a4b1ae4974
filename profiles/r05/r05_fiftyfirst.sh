# Round 5: the screen kernel's wavefronts as 1024-thread workgroups on a quarter of the CUs (LFQ_SCREEN_WG=1024) instead of one
# 256-thread workgroup on every CU: beside the next batch's count kernel a 256-thread DP workgroup leaves its CU room for ONE
# 1024-thread count workgroup instead of two (5 of 8 wave slots per SIMD in use); a 1024-thread one replaces a count workgroup
# (LFQ_SCREEN_WG existed for this measurement only, removed again -- profiles/NOTES.md)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = shape args, $2 = mode args; ENVV = env
  env $ENVV python bench.py $1 $2 --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-12s %-50s step %.3f (min %.3f max %.3f)  count %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$1', '$ENVV', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))" || tail -3 gpurun_out/r05_x.err
}
for i in 1 2 3; do
for kv in "X=0" "LFQ_SCREEN_WG=1024" "LFQ_SCREEN_WG=1024 LFQ_SCREEN_WAVES_PER_CU=8" "LFQ_SCREEN_WG=1024 LFQ_SCREEN_WAVES_PER_CU=2"; do
ENVV="$kv" one "--config C3" "--in-flight 4 --gate none"
done
done
for kv in "X=0" "LFQ_SCREEN_WG=1024"; do
ENVV="$kv" one "--config C2" "--in-flight 4 --gate none"
ENVV="$kv" one "--config C3" "--in-flight 4 --gate end"
done
