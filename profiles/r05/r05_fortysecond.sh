# Round 5: the DP streams confined to N CUs (LFQ_DP_CUS; hipExtStreamCreateWithCUMask) under the queue form of the second half
# of the round (four batches queued, no gate, lean count kernel with 1024-thread workgroups): does the count kernel keep its
# residency on the other CUs?  (Round 3 measured masks with two batches in flight and the host in the loop: NOTES.)
# (LFQ_DP_CUS existed for this measurement only: a dozen lines in acquire_streams, removed again -- profiles/NOTES.md)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = label, $2 = in-flight, $3 = gate, $4.. = shape args; ENVV = env
  lab=$1; nf=$2; gate=$3; shift 3
  env $ENVV python bench.py "$@" --in-flight $nf --gate $gate --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_dp_cus.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-40s [%s %-4s] step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$lab', '$nf', '$gate', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))" || tail -3 gpurun_out/r05_dp_cus.err
}
for i in 1 2; do
ENVV="X=0" one "C3 no mask" 4 none --config C3
for n in 32 64 128; do
ENVV="LFQ_DP_CUS=$n" one "C3 DP streams on $n CUs" 4 none --config C3
done
done
ENVV="X=0" one "C2 no mask" 4 none --config C2
for n in 64 128; do
ENVV="LFQ_DP_CUS=$n" one "C2 DP streams on $n CUs" 4 none --config C2
done
