set -u
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print('%-50s ms/step %.3f count %.3f scan %.3f dp %.3f  recs %s host_not_hidden %.3f' % (sys.argv[1], d['ms_per_step'], k['ms_count'], k['ms_scan'], k['ms_dp'], c.get('records_per_step'), c.get('host_ms_per_step_not_hidden', -1)))" "$*"; }
run --cols 3750000 --depth 200 --in-flight 2
run --cols 4600000 --depth 500 --in-flight 2
run --config C2 --in-flight 2
run --in-flight 2
run
