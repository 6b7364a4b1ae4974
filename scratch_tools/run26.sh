set -u
cd $GRAFT_REPO_ROOT
one() { python bench.py --mode chain "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-34s ms/region %.2f  %s' % (sys.argv[1], d['ms_per_step'], {k:round(v*1e3,1) for k,v in c.items() if k.startswith('s_') and isinstance(v,float)}))" "$*"; }
for t in 8 12 16; do
  export LFQ_HOST_LOOP_THREADS=$t
  echo "== LFQ_HOST_LOOP_THREADS=$t"
  one --steps 600
  one --steps 800 --overlap-regions
  one --steps 600 --workers 2
  one --steps 600 --workers 3
done
