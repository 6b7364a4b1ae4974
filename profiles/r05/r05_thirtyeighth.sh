cd $GRAFT_REPO_ROOT
for t in 8 6 5 7 6; do python bench.py --config C4 --host-threads $t --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>gpurun_out/ht_$t.err | tail -1 > gpurun_out/ht_$t.json; python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/ht_$t.json').read()); print('$t host threads:', d['config']['workload'][:3], d['ms_per_step'], d['config']['vcf_sha256'][:12])
except Exception as e:
    print('$t host threads: FAILED', e); print(open('gpurun_out/ht_$t.err').read()[-3000:])
"; done
for t in 6 8; do python bench.py --config C5 --host-threads $t --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>gpurun_out/ht5_$t.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5 $t host threads:', d['ms_per_step'], d['config']['vcf_sha256'][:12])"; done
