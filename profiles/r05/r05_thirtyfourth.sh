# Round 5: the scan as one launch (LFQ_SCAN_FUSED), then what else is on a shallow batch's chain under gate "tail": screen wavefronts
# per CU, first stretch of the mid class (LFQ_PHASE1_CHUNKS), count workgroups per CU when batches are queued without a gate
# (LFQ_COUNT_SHALLOW_LDS_PAD)
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
for i in 1 2; do
ENVV="LFQ_SCAN_FUSED=0" one "C2 three scan kernels" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=1" one "C2 scan fused" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_SCREEN_WAVES_PER_CU=8" one "C2 scan fused, 8 screen waves per CU" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_SCREEN_WAVES_PER_CU=16" one "C2 scan fused, 16 screen waves per CU" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_PHASE1_CHUNKS=4" one "C2 scan fused, first stretch 4 chunks" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_PHASE1_CHUNKS=8" one "C2 scan fused, first stretch 8 chunks" 3 tail --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_COUNT_SHALLOW_LDS_PAD=16000" one "C2 scan fused, 3 count workgroups per CU, no gate" 4 none --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_COUNT_SHALLOW_LDS_PAD=40000" one "C2 scan fused, 2 count workgroups per CU, no gate" 4 none --config C2
ENVV="LFQ_SCAN_FUSED=1 LFQ_COUNT_SHALLOW_LDS_PAD=16000" one "C2 scan fused, 3 count workgroups per CU, tail" 3 tail --config C2
done
ENVV="LFQ_SCAN_FUSED=0" one "200x three scan kernels" 3 tail --depth 200 --cols 3750000
ENVV="LFQ_SCAN_FUSED=1" one "200x scan fused" 3 tail --depth 200 --cols 3750000
ENVV="LFQ_SCAN_FUSED=1" one "200x scan fused, gate end" 4 end --depth 200 --cols 3750000
ENVV="LFQ_SCAN_FUSED=0" one "500x three scan kernels" 3 tail --depth 500 --cols 4600000
ENVV="LFQ_SCAN_FUSED=1" one "500x scan fused" 3 tail --depth 500 --cols 4600000
ENVV="LFQ_SCAN_FUSED=0" one "C3 three scan kernels" 4 none --config C3
ENVV="LFQ_SCAN_FUSED=1" one "C3 scan fused" 4 none --config C3
ENVV="LFQ_SCAN_FUSED=0" one "C3 three scan kernels" 4 none --config C3
ENVV="LFQ_SCAN_FUSED=1" one "C3 scan fused" 4 none --config C3
# the host's side of a C2 step (wait / finish / submit of bench.py's loop; launch / wait / d2h / finalize inside the library calls)
LFQ_TIMING=1 LFQ_BENCH_TRACE_STEPS=1 python bench.py --config C2 --in-flight 3 --gate tail --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_c2_host_trace.err >/dev/null
python - <<'P'
import re, statistics as S
a = {'wait': [], 'finish': [], 'submit': [], 'sum': [], 'kernels': []}
b = {'launch': [], 'wait': [], 'd2h': [], 'finalize': []}
for ln in open('gpurun_out/r05_c2_host_trace.err'):
    m = re.match(r'\[step\s+\d+\] wait ([\d.]+)\s+finish ([\d.]+)\s+submit ([\d.]+)\s+sum ([\d.]+) ms\s+kernels ([\d.]+)', ln)
    if m:
        for k, v in zip(a, m.groups()): a[k].append(float(v))
    m = re.match(r'\[lfq timing\] launch ([\d.]+)\s+wait ([\d.]+)\s+d2h ([\d.]+)\s+finalize ([\d.]+)', ln)
    if m:
        for k, v in zip(b, m.groups()): b[k].append(float(v))
print('C2 host trace (bench loop, ms): ' + '  '.join('%s median %.3f mean %.3f' % (k, S.median(v), S.mean(v)) for k, v in a.items() if v))
print('C2 library calls (ms):          ' + '  '.join('%s median %.3f mean %.3f' % (k, S.median(v[-180:]), S.mean(v[-180:])) for k, v in b.items() if v))
P
