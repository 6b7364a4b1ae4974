# Round 5: a batch's join on the big chain's stream instead of the light chain's (LFQ_JOIN_ON_SIDE): the next batch's scan no longer waits
# for this batch's last fold; crossed with the light chain's tail event, the scan as one launch, count workgroups per CU
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py tests/test_gpu_knobs.py tests/test_gpu_configs.py tests/test_gpu_shard.py -x -q -p no:cacheprovider -k "not full_batch" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
one() {     # $1 = label, $2 = in-flight, $3 = gate, $4.. = shape args; ENVV = env
  lab=$1; nf=$2; gate=$3; shift 3
  env $ENVV python bench.py "$@" --in-flight $nf --gate $gate --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-52s [%s %-4s] step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$lab', '$nf', '$gate', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))"
}
B="LFQ_SCAN_FUSED=0"
for i in 1 2; do
ENVV="$B LFQ_JOIN_ON_SIDE=0 LFQ_TAIL_LIGHT=0" one "C2 join on dps, tail behind retry (r04 form)" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=0" one "C2 join on dps" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_TAIL_LIGHT=0" one "C2 join on side, tail behind retry" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "C2 join on side" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_HEAVY_AFTER_SCREEN=0" one "C2 join on side, heavy first" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_TAIL_LIGHT=2" one "C2 join on side, tail behind scan" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_JOIN_ON_SIDE=1" one "C2 join on side, scan fused 1 per CU" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=2 LFQ_JOIN_ON_SIDE=1" one "C2 join on side, scan fused 2 per CU" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_COUNT_SHALLOW_LDS_PAD=16000" one "C2 join on side, 3 count workgroups per CU" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_PHASE1_CHUNKS=8" one "C2 join on side, first stretch 8 chunks" 3 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "C2 join on side" 4 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "C2 join on side" 2 tail --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "C2 join on side" 4 none --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_COUNT_SHALLOW_LDS_PAD=16000" one "C2 join on side, 3 count workgroups per CU" 4 none --config C2
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_COUNT_SHALLOW_LDS_PAD=40000" one "C2 join on side, 2 count workgroups per CU" 4 none --config C2
done
for sh in "--depth 200 --cols 3750000" "--depth 500 --cols 4600000"; do
ENVV="$B LFQ_JOIN_ON_SIDE=0 LFQ_TAIL_LIGHT=0" one "$sh r04 form" 3 tail $sh
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "$sh join on side" 3 tail $sh
ENVV="LFQ_SCAN_FUSED=2 LFQ_JOIN_ON_SIDE=1" one "$sh join on side, scan fused 2 per CU" 3 tail $sh
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "$sh join on side" 4 end $sh
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "$sh join on side" 4 none $sh
ENVV="$B LFQ_JOIN_ON_SIDE=1 LFQ_COUNT_SHALLOW_LDS_PAD=16000" one "$sh join on side, 3 wg per CU" 4 none $sh
done
for i in 1 2; do
ENVV="$B LFQ_JOIN_ON_SIDE=0" one "C3 join on dps" 4 none --config C3
ENVV="$B LFQ_JOIN_ON_SIDE=1" one "C3 join on side" 4 none --config C3
done
