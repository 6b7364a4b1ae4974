cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
python -m pytest tests/test_gpu_baq.py tests/test_gpu_chain.py tests/test_gpu_plpindel.py -q -x > gpurun_out/r03c/baq_tests.log 2>&1; echo "baq tests rc=$?"; tail -5 gpurun_out/r03c/baq_tests.log
python bench.py --mode baq --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('baq ms', d['ms_per_step'], d['value'])"
python bench.py --mode baq --idaq --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('baq idaq ms', d['ms_per_step'], d['value'])"
bash profiles/baq_profile.sh r03c > /dev/null 2>&1; cat gpurun_out/baq_prof_r03c/stats.md 2>/dev/null | head -20; ls gpurun_out | head -30
python bench.py --mode chain --steps 300 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('chain ms', d['ms_per_step'], {k:c[k] for k in c if k.startswith('s_')})"
python -m pytest tests/test_gpu_shard.py tests/test_gpu_configs.py tests/test_gpu_stability.py tests/test_gpu_uniq.py -q -x > gpurun_out/r03c/new_tests.log 2>&1; echo "new tests rc=$?"; tail -30 gpurun_out/r03c/new_tests.log
