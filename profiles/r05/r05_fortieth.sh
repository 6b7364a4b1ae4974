# Round 5: the residency of an ungated context's shallow count kernel by its GRID (n_cu x N workgroups, each walking its share of the
# passes) instead of by unused LDS, which leaves no LDS for the DP kernels' workgroups
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = label, $2 = in-flight, $3 = gate, $4.. = shape args; ENVV = env
  lab=$1; nf=$2; gate=$3; shift 3
  env $ENVV python bench.py "$@" --in-flight $nf --gate $gate --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-52s [%s %-4s] step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$lab', '$nf', '$gate', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))"
}
for i in 1 2; do
for sh in "--config C2" "--depth 500 --cols 4600000"; do
ENVV="X=0" one "$sh by LDS, 2 per CU" 4 none $sh
ENVV="LFQ_COUNT_SHALLOW_BY_GRID=1" one "$sh by grid, 2 per CU" 4 none $sh
ENVV="LFQ_COUNT_SHALLOW_BY_GRID=1 LFQ_COUNT_SHALLOW_WGS_NONE=3" one "$sh by grid, 3 per CU" 4 none $sh
ENVV="LFQ_COUNT_SHALLOW_BY_GRID=1 LFQ_COUNT_SHALLOW_WGS_NONE=1" one "$sh by grid, 1 per CU" 4 none $sh
ENVV="LFQ_COUNT_SHALLOW_BY_GRID=1" one "$sh by grid, 2 per CU, three queued" 3 none $sh
done
done
