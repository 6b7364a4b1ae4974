set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_knobs.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_plpindel.py tests/test_gpu_chain.py tests/test_gpu_configs.py tests/test_gpu_shard.py tests/test_gpu_srcq.py tests/test_gpu_stability.py tests/test_gpu_uniq.py 2>&1 | tail -3
python bench.py --mode chain --steps 600 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chain', d['ms_per_step'])"
