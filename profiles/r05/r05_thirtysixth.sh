# Round 5: (A) shallow batches queued without a gate, the count kernel with 2 / 1 / 3 workgroups per CU (LFQ_COUNT_SHALLOW_WGS_NONE);
# (B) C4 / C5 with a launch stream per context (LFQ_PRIVATE_STREAM) instead of the device's shared one, 4 / 8 hardware queues
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
genome() {  # $1 = label, $2 = config; ENVV = env
  env $ENVV python bench.py --config $2 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('repeats',{}); print('%-44s %s  ms_per_step %.1f (min %.1f max %.1f)  %s' % ('$1', '$2', d['ms_per_step'], r.get('ms_per_step_min',0), r.get('ms_per_step_max',0), d['config']['vcf_sha256'][:12]))"
}
for i in 1 2; do
ENVV="X=0" genome "shared launch stream" C4
ENVV="LFQ_PRIVATE_STREAM=1" genome "a launch stream per context" C4
ENVV="LFQ_PRIVATE_STREAM=1 GPU_MAX_HW_QUEUES=8" genome "a launch stream per context, 8 hardware queues" C4
ENVV="GPU_MAX_HW_QUEUES=8" genome "shared launch stream, 8 hardware queues" C4
done
ENVV="X=0" genome "shared launch stream" C5
ENVV="LFQ_PRIVATE_STREAM=1" genome "a launch stream per context" C5
ENVV="LFQ_PRIVATE_STREAM=1 GPU_MAX_HW_QUEUES=8" genome "a launch stream per context, 8 hardware queues" C5
LFQ_PRIVATE_STREAM=1 python -m pytest tests/test_gpu_chain.py tests/test_gpu_bench_configs.py tests/test_gpu_plpindel.py tests/test_gpu_pileup.py tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "not full_batch" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for i in 1 2; do
ENVV="X=0" one "C2 (tail)" 3 tail --config C2
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=0" one "C2 none, 4 count workgroups per CU" 4 none --config C2
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=3" one "C2 none, 3" 4 none --config C2
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=2" one "C2 none, 2" 4 none --config C2
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=1" one "C2 none, 1" 4 none --config C2
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=2" one "C2 none, 2, three in flight" 3 none --config C2
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=2 LFQ_SCREEN_WAVES_PER_CU=8" one "C2 none, 2, 8 screen waves per CU" 4 none --config C2
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=2 LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0" one "C2 none, 2, heavy first" 4 none --config C2
done
for sh in "--depth 200 --cols 3750000" "--depth 500 --cols 4600000"; do
ENVV="X=0" one "$sh (tail)" 3 tail $sh
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=0" one "$sh none, 4" 4 none $sh
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=2" one "$sh none, 2" 4 none $sh
ENVV="LFQ_COUNT_SHALLOW_WGS_NONE=1" one "$sh none, 1" 4 none $sh
done
python bench.py --config C2 --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('C2 auto:', d['ms_per_step'], d['config']['pipeline'][-220:])"
