set -u
cd $GRAFT_REPO_ROOT
for fl in "" "--overlap-regions"; do python bench.py --mode chain --steps 800 $fl 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['ms_per_step'], {k:round(v*1e3,2) for k,v in c.items() if k.startswith('s_') and isinstance(v,float)}, c['s_each'], c['snv_records'], c['indel_tests'])"; done
