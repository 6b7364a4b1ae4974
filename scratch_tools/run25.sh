set -u
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py $2 --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print('%-28s %-14s ms/step %.3f  %.1f M col/s  count %.3f dp %.3f (light %.2f mid %.2f big %.2f)' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['value']/1e6, k['ms_count'], k['ms_dp'], k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big']))" "$1" "$2"; }
run LFQ_CU_SPLIT=0 ""
run LFQ_CU_SPLIT=0 "--in-flight 2"
run LFQ_CU_SPLIT=32 "--in-flight 2"
run LFQ_CU_SPLIT=64 "--in-flight 2"
run LFQ_CU_SPLIT=128 "--in-flight 2"
run LFQ_CU_SPLIT=64 ""
