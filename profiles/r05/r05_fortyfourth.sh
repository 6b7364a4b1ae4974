# Round 5: (a) what the N > 1 step's exchange costs a rank -- --shard-path with a one-rank RCCL communicator
# (LFQ_BENCH_FORCE_DIST=1), the communicator's stream at normal / high priority, host side of the steps traced;
# (b) two sets of DP streams taken in turn by the contexts (LFQ_DP_STREAM_SETS=2) at the shallow shapes, where a batch's
# period is its mid chain behind the previous batch's on the same stream
# (LFQ_DP_STREAM_SETS existed for this measurement only, removed again -- profiles/NOTES.md)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = label, $2.. = bench args; ENVV = env
  lab=$1; shift 1
  env $ENVV LFQ_BENCH_TRACE_STEPS=1 python bench.py "$@" --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-50s step %.3f (min %.3f max %.3f)  count %.3f  scan %.3f  dp %.3f (l %.3f m %.3f b %.3f)  in flight %s %s' % (
    '$lab', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_scan'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config'].get('batches_in_flight'), d['config'].get('batch_gate')))" || tail -3 gpurun_out/r05_x.err
  grep '^\[step' gpurun_out/r05_x.err | awk '{w+=$4; f+=$6; s+=$8; n++} END {if (n) printf("    host per step: wait %.3f  finish %.3f  submit %.3f ms (%d steps)\n", w/n, f/n, s/n, n)}'
}
for sh in "--config C3" "--config C2"; do
ENVV="X=0" one "$sh --shard-path" $sh --shard-path
ENVV="LFQ_BENCH_FORCE_DIST=1 LFQ_BENCH_RCCL_PRIO=0" one "$sh --shard-path, one-rank RCCL" $sh --shard-path
ENVV="LFQ_BENCH_FORCE_DIST=1" one "$sh --shard-path, one-rank RCCL, high priority" $sh --shard-path
done
for i in 1 2; do
for sh in "--config C2" "--depth 500 --cols 4600000" "--depth 200 --cols 3750000" "--config C3"; do
ENVV="X=0" one "$sh" $sh --in-flight 4 --gate none
ENVV="LFQ_DP_STREAM_SETS=2" one "$sh two DP stream sets" $sh --in-flight 4 --gate none
ENVV="LFQ_DP_STREAM_SETS=2 GPU_MAX_HW_QUEUES=8" one "$sh two DP stream sets, 8 queues" $sh --in-flight 4 --gate none
done
done
