set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_stability.py 2>&1 | tail -2
run() { env $1 python bench.py $2 --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print('%-28s ms/step %.3f count %.3f scan %.3f dp %.3f roof %.2f recs %s' % (sys.argv[2], d['ms_per_step'], k['ms_count'], k['ms_scan'], k['ms_dp'], d['roofline']['frac'], c.get('records_per_step')))" "$1" "$2"; }
run X=1 "--cols 3750000 --depth 200"
run X=1 "--cols 4600000 --depth 500"
run X=1 "--config C2"
