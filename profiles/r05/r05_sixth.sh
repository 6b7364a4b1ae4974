set -u
cd $GRAFT_REPO_ROOT
one() {   # label, args..., env via ENVV
  lab=$1; shift
  env $ENVV python bench.py --steps 40 --warmup 5 --repeats 3 "$@" --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('%-52s first %.3f min %.3f med %.3f max %.3f  count %.3f dp %.3f  [%s %s] records %d' % ('$lab', r['ms_per_step_first'], r['ms_per_step_min'], r['ms_per_step_median'], r['ms_per_step_max'], c['kernel_ms']['ms_count'], c['kernel_ms']['ms_dp'], c['batches_in_flight'], c['batch_gate'], c['records_per_step']))"
}
for i in 1 2; do
  ENVV="LFQ_COUNT_WAVES_PER_WG=16" one "4 end, 16/wg" --in-flight 4 --gate end
  ENVV="X=0" one "4 none" --in-flight 4 --gate none
  ENVV="LFQ_BIG_ON_SIDE=1" one "4 none, big on side" --in-flight 4 --gate none
  ENVV="LFQ_BIG_ON_SIDE=1 LFQ_COUNT_WAVES_PER_WG=16" one "4 none, big on side, 16/wg" --in-flight 4 --gate none
  ENVV="LFQ_BIG_ON_SIDE=1 LFQ_COUNT_WAVES_PER_WG=16" one "3 none, big on side, 16/wg" --in-flight 3 --gate none
  ENVV="LFQ_BIG_ON_SIDE=1 LFQ_COUNT_WAVES_PER_WG=16" one "4 tail, big on side, 16/wg" --in-flight 4 --gate tail
  ENVV="LFQ_BIG_ON_SIDE=1 LFQ_COUNT_PERSIST=6" one "4 none, big on side, resident 6" --in-flight 4 --gate none
  ENVV="LFQ_BIG_ON_SIDE=1 LFQ_COUNT_PERSIST=5" one "4 none, big on side, resident 5" --in-flight 4 --gate none
  ENVV="LFQ_BIG_ON_SIDE=1 LFQ_COUNT_PERSIST=4" one "4 none, big on side, resident 4" --in-flight 4 --gate none
done
