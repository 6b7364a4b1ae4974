cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check --repeats 3 $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['kernel_ms']; print(round(d['value']/1e6,1),'M', round(d['ms_per_step'],3), 'median', round(d['repeats']['ms_per_step_median'],3), 'count',round(k['ms_count'],3),'dp',round(k['ms_dp'],3), 'recs', d['config']['records_per_step'])"; }
ARGS="" run LFQ_X=0
ARGS="--in-flight 2" run LFQ_X=0
for s in 32 48 64 96; do ARGS="--in-flight 2" run LFQ_CU_SPLIT=$s; done
ARGS="" run LFQ_CU_SPLIT=64
