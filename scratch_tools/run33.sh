set -u
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_plpindel.py tests/test_gpu_chain.py 2>&1 | tail -2
LFQ_TIMING=1 python bench.py --mode chain --steps 300 2>&1 | grep "indel pileup" | tail -2
LFQ_PILEUP_TILES=0 LFQ_TIMING=1 python bench.py --mode chain --steps 300 2>&1 | grep "indel pileup" | tail -2
one() { python bench.py --mode chain "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-40s ms/region %.2f  %s' % (sys.argv[1], d['ms_per_step'], {k:round(v*1e3,1) for k,v in c.items() if k.startswith('s_') and isinstance(v,float)}))" "$*"; }
one --steps 600
one --steps 800 --overlap-regions
one --steps 800 --overlap-regions
one --steps 600 --workers 2
one --steps 600 --workers 3
