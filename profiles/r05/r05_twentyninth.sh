# Round 5: the narrow-band reads with indels (IDAQ instantiation) beside the plain launches, on a side stream with scratch slots of their own
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_baq.py tests/test_gpu_chain.py tests/test_gpu_plpindel.py tests/test_gpu_bench_configs.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for c in C4 C5; do python bench.py --config $c --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], d['ms_per_step'], d['config']['vcf_sha256'][:12])"; done
python bench.py --mode baq --steps 100 --idaq 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('baq --idaq', d['ms_per_step'])"
python bench.py --mode baq --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('baq', d['ms_per_step'])"
python bench.py --mode chain --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chain', d['ms_per_step'])"
