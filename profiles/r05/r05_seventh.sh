set -u
cd $GRAFT_REPO_ROOT
one() {   # label, args..., env via ENVV
  lab=$1; shift
  env $ENVV python bench.py --steps 40 --warmup 5 --repeats 3 "$@" --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('%-44s first %.3f min %.3f med %.3f max %.3f  count %.3f dp %.3f  [%s %s] records %d' % ('$lab', r['ms_per_step_first'], r['ms_per_step_min'], r['ms_per_step_median'], r['ms_per_step_max'], c['kernel_ms']['ms_count'], c['kernel_ms']['ms_dp'], c['batches_in_flight'], c['batch_gate'], c['records_per_step']))"
}
python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "not full_batch" 2>&1 | tail -2
LFQ_COUNT_UNROLL=4 LFQ_COUNT_PRIO=1 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "not full_batch" 2>&1 | tail -2
for i in 1 2; do
  ENVV="X=0" one "4 end" --in-flight 4 --gate end
  ENVV="LFQ_COUNT_UNROLL=4" one "4 end, unroll 4" --in-flight 4 --gate end
  ENVV="LFQ_COUNT_PRIO=1" one "4 end, prio" --in-flight 4 --gate end
  ENVV="X=0" one "4 none" --in-flight 4 --gate none
  ENVV="LFQ_COUNT_PRIO=1" one "4 none, prio" --in-flight 4 --gate none
  ENVV="LFQ_COUNT_UNROLL=4" one "4 none, unroll 4" --in-flight 4 --gate none
  ENVV="LFQ_COUNT_UNROLL=4 LFQ_COUNT_PRIO=1" one "4 none, unroll 4, prio" --in-flight 4 --gate none
  ENVV="LFQ_COUNT_PRIO=1 LFQ_SCREEN_WAVES_PER_CU=8" one "4 none, prio, 8 screen waves" --in-flight 4 --gate none
  ENVV="X=0" one "C2 4 none" --config C2 --in-flight 4 --gate none
  ENVV="X=0" one "C2 3 tail" --config C2 --in-flight 3 --gate tail
  ENVV="X=0" one "C2 2 tail" --config C2 --in-flight 2 --gate tail
done
