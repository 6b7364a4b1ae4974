set -u
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print('%-40s ms/step %.3f count %.3f scan %.3f dp %.3f  recs %s roof %.2f' % (sys.argv[1], d['ms_per_step'], k['ms_count'], k['ms_scan'], k['ms_dp'], c.get('records_per_step'), d['roofline']['frac']))" "$*"; }
for env in "LFQ_COUNT_LPG4_BELOW=0 LFQ_COUNT_LPG8_BELOW=0" "LFQ_COUNT_LPG4_BELOW=0 LFQ_COUNT_LPG8_BELOW=100000" "LFQ_COUNT_LPG4_BELOW=100000"; do
  echo "== $env"
  env $env bash -c "$(declare -f run); run --cols 3750000 --depth 200; run --cols 4600000 --depth 500; run --config C2"
done
