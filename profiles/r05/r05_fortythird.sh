# Round 5: the LAUNCH stream (count kernels) confined to N of the 256 CUs (LFQ_COUNT_CUS), DP streams unmasked at high priority:
# the DP kernels of the batch before then find CUs no count workgroup ever takes.  (The other way round -- DP streams masked,
# r05_fortysecond.sh -- loses the priority with the mask and the DP kernels starve beside the count kernel.)
# (LFQ_COUNT_CUS existed for this measurement only, removed again -- profiles/NOTES.md)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = label, $2 = in-flight, $3 = gate, $4.. = shape args; ENVV = env
  lab=$1; nf=$2; gate=$3; shift 3
  env $ENVV python bench.py "$@" --in-flight $nf --gate $gate --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_count_cus.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-40s [%s %-4s] step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$lab', '$nf', '$gate', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))" || tail -3 gpurun_out/r05_count_cus.err
}
for i in 1 2; do
ENVV="X=0" one "C3 no mask" 4 none --config C3
for n in 248 240 224 208 192; do
ENVV="LFQ_COUNT_CUS=$n" one "C3 count kernel on $n CUs" 4 none --config C3
done
done
ENVV="X=0" one "C3 no mask" 4 end --config C3
for n in 240 224; do
ENVV="LFQ_COUNT_CUS=$n" one "C3 count kernel on $n CUs" 4 end --config C3
done
ENVV="X=0" one "C2 no mask" 4 none --config C2
for n in 240 224 192; do
ENVV="LFQ_COUNT_CUS=$n" one "C2 count kernel on $n CUs" 4 none --config C2
done
