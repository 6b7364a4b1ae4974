set -u
cd $GRAFT_REPO_ROOT
LFQ_TIMING=1 python bench.py --mode chain --steps 6 --warmup 2 2>&1 | grep -v "^{" | tail -40
python bench.py --mode chain --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['ms_per_step'], {k:round(v*1e3,2) for k,v in c.items() if k.startswith('s_') and isinstance(v,float)})"
