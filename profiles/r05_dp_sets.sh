# DP chains of consecutive queued batches beside each other: LFQ_DP_STREAM_SETS sets of DP streams x GPU_MAX_HW_QUEUES
cd $GRAFT_REPO_ROOT
one() {
  lab=$1; shift
  env $ENVV python bench.py --steps 40 --warmup 5 --repeats 3 "$@" --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('%-52s step %.3f (min %.3f max %.3f)  count %.3f  dp span %.3f  [%s %s] records %d' % ('$lab', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], c['kernel_ms']['ms_count'], c['kernel_ms']['ms_dp'], c['batches_in_flight'], c['batch_gate'], c['records_per_step']))"
}
LFQ_DP_STREAM_SETS=2 GPU_MAX_HW_QUEUES=8 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py -x -q -p no:cacheprovider -k "not full_batch" 2>&1 | tail -2
for i in 1 2; do
ENVV="X=0" one "1 set, default queues" --in-flight 4 --gate none
ENVV="GPU_MAX_HW_QUEUES=8" one "1 set, 8 queues" --in-flight 4 --gate none
ENVV="LFQ_DP_STREAM_SETS=2" one "2 sets, default queues" --in-flight 4 --gate none
ENVV="LFQ_DP_STREAM_SETS=2 GPU_MAX_HW_QUEUES=8" one "2 sets, 8 queues" --in-flight 4 --gate none
ENVV="LFQ_DP_STREAM_SETS=4 GPU_MAX_HW_QUEUES=16" one "4 sets, 16 queues" --in-flight 4 --gate none
ENVV="LFQ_DP_STREAM_SETS=2 GPU_MAX_HW_QUEUES=8" one "2 sets, 8 queues, gate end" --in-flight 4 --gate end
ENVV="LFQ_DP_STREAM_SETS=2 GPU_MAX_HW_QUEUES=8" one "C2: 2 sets, 8 queues, none" --config C2 --in-flight 4 --gate none
ENVV="LFQ_DP_STREAM_SETS=2 GPU_MAX_HW_QUEUES=8" one "C2: 2 sets, 8 queues, tail" --config C2 --in-flight 3 --gate tail
ENVV="X=0" one "C2: 1 set, tail" --config C2 --in-flight 3 --gate tail
done
