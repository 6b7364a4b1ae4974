# Round 5: what a shallow batch's period is made of under gate "tail" -- the light chain's tail event behind the screen kernel instead of
# behind the retry kernel (LFQ_TAIL_LIGHT), the heavy columns' strand counts behind the screen kernel (LFQ_HEAVY_AFTER_SCREEN), scan tiles
# of 1024 instead of 4096 columns (a build: liblofreq_amd_scan256.so); C2, 200x, 500x, C3
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py tests/test_gpu_knobs.py -x -q -p no:cacheprovider -k "not full_batch" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
one() {     # $1 = label, $2 = in-flight, $3 = gate, $4.. = shape args; ENVV = env
  lab=$1; nf=$2; gate=$3; shift 3
  env $ENVV python bench.py "$@" --in-flight $nf --gate $gate --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-52s [%s %-4s] step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$lab', '$nf', '$gate', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))"
}
SCAN256=$GRAFT_REPO_ROOT/lofreq_amd/liblofreq_amd_scan256.so
for i in 1 2; do
ENVV="LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0" one "C2 tail behind retry, heavy first (before)" 3 tail --config C2
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=0" one "C2 tail behind screen, heavy first" 3 tail --config C2
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1" one "C2 tail behind screen, heavy late" 3 tail --config C2
ENVV="LFQ_TAIL_LIGHT=2 LFQ_HEAVY_AFTER_SCREEN=1" one "C2 tail behind scan, heavy late" 3 tail --config C2
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1 LFQ_AMD_LIB=$SCAN256" one "C2 tail behind screen, heavy late, scan tiles 1024" 3 tail --config C2
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1" one "C2 ... two in flight" 2 tail --config C2
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1" one "C2 ... four in flight" 4 tail --config C2
done
ENVV="LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0" one "200x before" 3 tail --depth 200 --cols 3750000
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1" one "200x tail behind screen, heavy late" 3 tail --depth 200 --cols 3750000
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1 LFQ_AMD_LIB=$SCAN256" one "200x ... scan tiles 1024" 3 tail --depth 200 --cols 3750000
ENVV="LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0" one "200x before, gate end" 4 end --depth 200 --cols 3750000
ENVV="LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0" one "500x before" 3 tail --depth 500 --cols 4600000
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1" one "500x tail behind screen, heavy late" 3 tail --depth 500 --cols 4600000
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1 LFQ_AMD_LIB=$SCAN256" one "500x ... scan tiles 1024" 3 tail --depth 500 --cols 4600000
ENVV="LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0" one "500x before, gate end" 4 end --depth 500 --cols 4600000
ENVV="LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0" one "C3 before" 4 none --config C3
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1" one "C3 heavy late" 4 none --config C3
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1 LFQ_AMD_LIB=$SCAN256" one "C3 heavy late, scan tiles 1024" 4 none --config C3
ENVV="LFQ_TAIL_LIGHT=1 LFQ_HEAVY_AFTER_SCREEN=1" one "C3 three in flight, tail behind screen" 3 tail --config C3
