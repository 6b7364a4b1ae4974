# Round 5: the shared-wavefront count kernel with two workgroups per CU ALONE (gate end / tail): the collection's trace showed 0.28 ms
# at C2 where four per CU take 0.30
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
for sh in "--config C2" "--depth 500 --cols 4600000" "--depth 200 --cols 3750000"; do
for pad in 0 17000 44000; do
ENVV="LFQ_COUNT_SHALLOW_LDS_PAD=$pad" one "$sh pad $pad" 4 end $sh
done
ENVV="LFQ_COUNT_SHALLOW_LDS_PAD=44000" one "$sh pad 44000" 3 tail $sh
ENVV="LFQ_COUNT_SHALLOW_LDS_PAD=0" one "$sh pad 0" 3 tail $sh
done
done
