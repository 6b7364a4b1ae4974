set -u
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py $2 --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print('%-34s %-28s ms/step %.3f count %.3f scan %.3f dp %.3f (light %.2f mid %.2f big %.2f) recs %s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], k['ms_count'], k['ms_scan'], k['ms_dp'], k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], c.get('records_per_step')))" "$1" "$2"; }
for w in 4 8 16; do
run LFQ_SCREEN_WAVES_PER_CU=$w "--cols 3750000 --depth 200"
run LFQ_SCREEN_WAVES_PER_CU=$w "--cols 4600000 --depth 500"
run LFQ_SCREEN_WAVES_PER_CU=$w "--config C2"
done
