# Round 5: the DP knobs again under the queue form of the second half of the round (four batches queued, no gate): with the
# chains beside the next count kernel their LENGTH matters less and their total work more than when they were swept alone
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = shape args; ENVV = env
  env $ENVV python bench.py $1 --in-flight 4 --gate none --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-12s %-44s step %.3f (min %.3f max %.3f)  count %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$1', '$ENVV', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))" || tail -3 gpurun_out/r05_x.err
}
for sh in "--config C3" "--config C2"; do
for kv in "X=0" "LFQ_SEG_MAX=2" "LFQ_SEG_MAX=4" "LFQ_SPLIT_POOL_CELLS=0" "LFQ_SEG_BUDGET_MID=1024 LFQ_SEG_BUDGET_BIG=1024" \
          "LFQ_PHASE1_CHUNKS=8" "LFQ_PHASE1_CHUNKS=64" "LFQ_LIGHT_KERNEL=wave" "LFQ_SCREEN_WAVES_PER_CU=2" "LFQ_SCREEN_WAVES_PER_CU=3" \
          "LFQ_SCREEN_WAVES_PER_CU=6" "LFQ_SCREEN_ROUNDS=8" "LFQ_SCREEN_ROUNDS=12" "LFQ_SCREEN_ROUNDS=48" "X=1"; do
ENVV="$kv" one "$sh"
done
done
