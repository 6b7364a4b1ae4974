set -u
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print('%-110s ms/step %.3f  %.1f M  dp %.3f (light %.2f mid %.2f big %.2f) recs %s' % (sys.argv[1][-110:], d['ms_per_step'], d['value']/1e6, k['ms_dp'], k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], c.get('records_per_step')))" "$1"; }
L=LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/liblofreq_amd_seg16.so
run "X=1"
run "$L LFQ_SEG_MAX=8"
run "$L LFQ_SEG_MAX=16 LFQ_SEG_BUDGET_MID=8192 LFQ_SEG_BUDGET_BIG=8192 LFQ_SPLIT_POOL_CELLS=33554432"
run "$L LFQ_SEG_MAX=12 LFQ_SEG_BUDGET_MID=8192 LFQ_SEG_BUDGET_BIG=8192 LFQ_SPLIT_POOL_CELLS=33554432"
run "$L LFQ_SEG_MAX=16 LFQ_SEG_BUDGET_MID=8192 LFQ_SEG_BUDGET_BIG=4096 LFQ_SPLIT_POOL_CELLS=33554432"
run "$L LFQ_SEG_MAX=16 LFQ_SEG_BUDGET_MID=4096 LFQ_SEG_BUDGET_BIG=8192 LFQ_SPLIT_POOL_CELLS=33554432"
