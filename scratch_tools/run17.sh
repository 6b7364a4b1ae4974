set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_chain.py tests/test_gpu_plpindel.py tests/test_gpu_configs.py tests/test_gpu_pileup.py 2>&1 | tail -5
LFQ_TIMING=1 python bench.py --mode chain --steps 4 --warmup 2 2>&1 | grep "indel pileup\|baq:" | tail -4
for i in 1 2; do python bench.py --mode chain --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['ms_per_step'], {k:round(v*1e3,2) for k,v in c.items() if k.startswith('s_') and isinstance(v,float)})"; done
