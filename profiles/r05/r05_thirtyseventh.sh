# Round 5: the lean count kernel (10 000x) capped to N workgroups per CU by unused LDS (LFQ_COUNT_LEAN_LDS_PAD), with smaller workgroups and
# more chunks in flight per lane: constant residency for the count kernel AND free wave slots for the DP kernels of the batch before
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = label, $2 = in-flight, $3 = gate, $4.. = shape args; ENVV = env
  lab=$1; nf=$2; gate=$3; shift 3
  env $ENVV python bench.py "$@" --in-flight $nf --gate $gate --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-64s [%s %-4s] step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$lab', '$nf', '$gate', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))"
}
for i in 1 2; do
ENVV="X=0" one "C3 16 waves per wg, 2 chunks in flight (default)" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_LEAN_LDS_PAD=80000" one "8 waves, 4 chunks, 2 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_LEAN_LDS_PAD=54000" one "8 waves, 4 chunks, 3 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_AHEAD_DEEP=3 LFQ_COUNT_LEAN_LDS_PAD=54000" one "8 waves, 3 chunks, 3 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_AHEAD_DEEP=2 LFQ_COUNT_LEAN_LDS_PAD=54000" one "8 waves, 2 chunks, 3 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=4 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_LEAN_LDS_PAD=32000" one "4 waves, 4 chunks, 5 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=4 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_LEAN_LDS_PAD=40000" one "4 waves, 4 chunks, 4 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=4 LFQ_COUNT_AHEAD_DEEP=3 LFQ_COUNT_LEAN_LDS_PAD=27000" one "4 waves, 3 chunks, 6 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=4 LFQ_COUNT_AHEAD_DEEP=2 LFQ_COUNT_LEAN_LDS_PAD=27000" one "4 waves, 2 chunks, 6 wg per CU" 4 none --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=16 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_LEAN_LDS_PAD=100000" one "16 waves, 4 chunks, 1 wg per CU" 4 none --config C3
done
# the count kernel alone in the same forms (gate end: one batch's kernels at a time)
ENVV="X=0" one "C3 default, gate end" 4 end --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_LEAN_LDS_PAD=54000" one "8 waves, 4 chunks, 3 wg per CU, gate end" 4 end --config C3
ENVV="LFQ_COUNT_WAVES_PER_WG=4 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_LEAN_LDS_PAD=32000" one "4 waves, 4 chunks, 5 wg per CU, gate end" 4 end --config C3
# C4 / C5 as bench.py runs them now (a launch stream per context, 8 hardware queues), and the new test
python -m pytest tests/test_gpu_stability.py tests/test_gpu_bench_configs.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for c in C4 C5 C4; do python bench.py --config $c --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], d['ms_per_step'], d['config']['vcf_sha256'][:12])"; done
LFQ_BENCH_SHARED_STREAM=1 python bench.py --config C4 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shared stream, 8 queues:', d['config']['workload'][:3], d['ms_per_step'], d['config']['vcf_sha256'][:12])"
for t in 6 8; do python bench.py --config C4 --host-threads $t --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t host threads:', d['config']['workload'][:3], d['ms_per_step'], d['config']['vcf_sha256'][:12])"; done
ENVV="X=0" one "200x none (4 wg per CU by the lanes-per-column rule)" 4 none --depth 200 --cols 3750000
