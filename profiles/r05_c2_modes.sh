cd $GRAFT_REPO_ROOT
one() {
  lab=$1; shift
  python bench.py --config C2 --steps 40 --warmup 5 --repeats 3 "$@" --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['repeats']; c = d['config']
print('%-28s step %.3f (min %.3f max %.3f) count %.3f dp %.3f | %s' % ('$lab', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], c['kernel_ms']['ms_count'], c['kernel_ms']['ms_dp'], c['pipeline'][70:]))"
}
for i in 1 2; do
one "3 tail" --in-flight 3 --gate tail
one "2 tail" --in-flight 2 --gate tail
one "4 tail" --in-flight 4 --gate tail
one "4 none" --in-flight 4 --gate none
one "auto"
done
