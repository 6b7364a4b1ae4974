# Round 5: what the N > 1 form of the step costs a rank (layer 1 + shard exchange instead of layer 2), measured on one GPU:
# --shard-path alone (no collectives: world of one), and with LFQ_BENCH_FORCE_DIST=1 (a one-rank RCCL communicator, every
# collective of the step -- two all-gathers and the record gather -- inside the timed region)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = label, $2.. = args; ENVV = env
  lab=$1; shift 1
  env $ENVV python bench.py "$@" --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_shard_form.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-44s step first %.3f median %.3f (min %.3f max %.3f)  count %.3f  dp %.3f  in flight %s gate %s  records %d' % (
    '$lab', r['ms_per_step_first'], r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    d['config'].get('batches_in_flight'), d['config'].get('batch_gate'), d['config']['records_per_step']))" || tail -5 gpurun_out/r05_shard_form.err
}
for sh in "--config C3" "--config C3" "--config C2"; do
ENVV="X=0" one "$sh layer 2 (N = 1 form)" $sh
ENVV="X=0" one "$sh --shard-path" $sh --shard-path
ENVV="LFQ_BENCH_FORCE_DIST=1" one "$sh --shard-path, one-rank RCCL" $sh --shard-path
done
