# Round 5: the sharded step's exchange after the change -- test counts over a host group (gloo), ONE asynchronous RCCL gather
# per step collected a step later -- against the blocking forms, with a one-rank RCCL communicator (LFQ_BENCH_FORCE_DIST=1)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = label, $2.. = bench args; ENVV = env
  lab=$1; shift 1
  env $ENVV LFQ_BENCH_TRACE_STEPS=1 python bench.py "$@" --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-66s step %.3f (min %.3f max %.3f)  count %.3f  dp %.3f  in flight %s %s  records %d' % (
    '$lab', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    d['config'].get('batches_in_flight'), d['config'].get('batch_gate'), d['config']['records_per_step']))" || tail -5 gpurun_out/r05_x.err
  grep '^\[step' gpurun_out/r05_x.err | awk '{w+=$4; f+=$6; s+=$8; n++} END {if (n) printf("    host per step: wait %.3f  finish %.3f  submit %.3f ms (%d steps)\n", w/n, f/n, s/n, n)}'
}
for i in 1; do
for sh in "--config C3" "--config C2"; do
ENVV="X=0" one "$sh layer 2" $sh
ENVV="X=0" one "$sh --shard-path (no communicator)" $sh --shard-path
ENVV="LFQ_BENCH_FORCE_DIST=1" one "$sh one-rank RCCL: counts gloo, gather async" $sh --shard-path
ENVV="LFQ_BENCH_FORCE_DIST=1 LFQ_BENCH_EXCHANGE=rccl" one "$sh one-rank RCCL: counts RCCL, gather async" $sh --shard-path
ENVV="LFQ_BENCH_FORCE_DIST=1 LFQ_BENCH_EXCHANGE_LAG=0" one "$sh one-rank RCCL: counts gloo, gather collected at once" $sh --shard-path
ENVV="LFQ_BENCH_FORCE_DIST=1 LFQ_BENCH_EXCHANGE=rccl LFQ_BENCH_EXCHANGE_LAG=0" one "$sh one-rank RCCL: counts RCCL, gather collected at once" $sh --shard-path
done
done
