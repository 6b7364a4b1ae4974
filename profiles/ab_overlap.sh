# Round 5: how a caller keeps the device fed at 10 000x (and at 1000x), A/B in one session on one box.
#   in-flight 1                      wait(k); submit(k + 1); finish(k)          -- a host wake-up + launch between batches
#   in-flight 2, gate end            submit(k + 1) before wait(k), its count kernel behind batch k's last kernel (device event)
#   in-flight 2, gate tail           ... behind batch k's row-bound DP kernels
#   in-flight 2, gate none           ... behind batch k's count kernel only (all of DP(k) beside count(k + 1))
# crossed with the count kernel's form: 4 / 8 / 16 columns per workgroup (LFQ_COUNT_WAVES_PER_WG), or resident with W
# workgroups per CU (LFQ_COUNT_PERSIST=W), which leaves 8 - W wave slots per SIMD to the other batch's DP kernels.
# usage: bash profiles/ab_overlap.sh [C3|C2] > gpurun_out/r05_ab_overlap_C3.txt
CFG=${1:-C3}
STEPS=${2:-60}
run() {     # $1 = label, $2 = in-flight, $3 = gate, rest = env
  lab=$1; nf=$2; gate=$3; shift 3
  env "$@" python bench.py --config $CFG --in-flight $nf --gate $gate --steps $STEPS --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-44s in-flight $nf gate %-4s  step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$lab', '$gate', d['ms_per_step'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))"
}
run "default (4 columns per workgroup)" 1 tail X=0
run "default" 2 end X=0
run "default" 2 tail X=0
run "default" 2 none X=0
run "default, big kernel on the side stream" 2 none LFQ_BIG_ON_SIDE=1
[ "$CFG" = C3 ] || exit 0       # (the forms below are those of the one-column-per-wavefront kernel: depth >= 4096)
for w in 8 16; do
  run "$w columns per workgroup" 1 tail LFQ_COUNT_WAVES_PER_WG=$w
  run "$w columns per workgroup" 2 tail LFQ_COUNT_WAVES_PER_WG=$w
  run "$w columns per workgroup, big on side" 2 none LFQ_COUNT_WAVES_PER_WG=$w LFQ_BIG_ON_SIDE=1
done
for W in 7 6 5 4 3; do
  run "resident, $W workgroups per CU" 1 tail LFQ_COUNT_PERSIST=$W
  run "resident, $W per CU" 2 end LFQ_COUNT_PERSIST=$W
  run "resident, $W per CU" 2 tail LFQ_COUNT_PERSIST=$W
  run "resident, $W per CU, big on side" 2 none LFQ_COUNT_PERSIST=$W LFQ_BIG_ON_SIDE=1
done
run "resident, 5 per CU, slices of 1" 1 tail LFQ_COUNT_PERSIST=5 LFQ_COUNT_SLICE=1
run "resident, 5 per CU, slices of 4" 1 tail LFQ_COUNT_PERSIST=5 LFQ_COUNT_SLICE=4
