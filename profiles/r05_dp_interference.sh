# which DP class slows the count kernel it runs beside (four queued batches, no gate)?  LFQ_DEBUG_SKIP leaves classes out (wrong
# results, timing only)
cd $GRAFT_REPO_ROOT
one() {
  lab=$1; shift
  env $ENVV python bench.py --steps 40 --warmup 5 --repeats 3 --in-flight 4 --gate none --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('%-40s step %.3f (min %.3f)  count %.3f  dp span %.3f  records %d' % ('$lab', r['ms_per_step_median'], r['ms_per_step_min'], c['kernel_ms']['ms_count'], c['kernel_ms']['ms_dp'], c['records_per_step']))"
}
for i in 1 2; do
ENVV="X=0" one "all DP classes"
ENVV="LFQ_DEBUG_SKIP=light" one "without the light class (screen + retry)"
ENVV="LFQ_DEBUG_SKIP=mid,big" one "without mid and big"
ENVV="LFQ_DEBUG_SKIP=mid" one "without mid"
ENVV="LFQ_DEBUG_SKIP=big" one "without big"
ENVV="LFQ_DEBUG_SKIP=light,mid,big" one "no DP class at all (scan + strand kernels only)"
ENVV="LFQ_SCREEN_WAVES_PER_CU=2" one "screen: 2 wavefronts per CU"
ENVV="LFQ_SCREEN_WAVES_PER_CU=1" one "screen: 1 wavefront per CU"
done
