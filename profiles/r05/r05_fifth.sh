set -u
cd $GRAFT_REPO_ROOT
one() {   # label, args..., env via ENVV
  lab=$1; shift
  env $ENVV python bench.py --steps 20 --warmup 5 "$@" --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('%-34s first %.3f min %.3f med %.3f max %.3f  kernels %.3f  count %.3f dp %.3f  [%s %s]' % ('$lab', r['ms_per_step_first'], r['ms_per_step_min'], r['ms_per_step_median'], r['ms_per_step_max'], c['ms_kernels'], c['kernel_ms']['ms_count'], c['kernel_ms']['ms_dp'], c['batches_in_flight'], c['batch_gate']))"
}
for i in 1 2 3 4; do
  ENVV="X=0" one "1 tail" --in-flight 1 --gate tail
  ENVV="X=0" one "2 end" --in-flight 2 --gate end
  ENVV="X=0" one "3 end" --in-flight 3 --gate end
  ENVV="X=0" one "4 end" --in-flight 4 --gate end
  ENVV="X=0" one "auto" 
  ENVV="LFQ_COUNT_WAVES_PER_WG=8" one "3 end, 8 columns per wg" --in-flight 3 --gate end
  ENVV="LFQ_COUNT_WAVES_PER_WG=16" one "3 end, 16 columns per wg" --in-flight 3 --gate end
done
echo "--- C2"
for i in 1 2; do
  ENVV="X=0" one "C2 auto" --config C2
  ENVV="X=0" one "C2 2 tail" --config C2 --in-flight 2 --gate tail
  ENVV="X=0" one "C2 3 tail" --config C2 --in-flight 3 --gate tail
  ENVV="X=0" one "C2 3 end" --config C2 --in-flight 3 --gate end
  ENVV="X=0" one "200x auto" --cols 3750000 --depth 200
  ENVV="X=0" one "500x auto" --cols 4600000 --depth 500
done
