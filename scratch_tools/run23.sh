set -u
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print('%-40s ms/step %.3f count %.3f scan %.3f dp %.3f  recs %s roof %.2f host_not_hidden %.3f' % (sys.argv[1], d['ms_per_step'], k['ms_count'], k['ms_scan'], k['ms_dp'], c.get('records_per_step'), d['roofline']['frac'], c.get('host_ms_per_step_not_hidden', -1)))" "$*"; }
for env in "LFQ_SB_PAR_MIN_COST=20000" "LFQ_SB_PAR_MIN_COST=3000"; do
  echo "== $env"
  env $env bash -c "$(declare -f run); run --cols 3750000 --depth 200; run --cols 4600000 --depth 500; run --config C2"
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('host_ms_per_step_not_hidden'), d['repeats'])"
python bench.py --mode chain --steps 400 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chain', d['ms_per_step'])"
python bench.py --mode host-abi --steps 60 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('host-abi', d['ms_per_step'], d['config']['effective_GBps'])"
