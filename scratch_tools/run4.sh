cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
python -m pytest tests/test_gpu_baq.py -q -x > gpurun_out/r03d/baq_tests.log 2>&1; echo "baq tests rc=$?"; tail -3 gpurun_out/r03d/baq_tests.log
bash profiles/baq_profile.sh r03d > /dev/null 2>&1; head -6 gpurun_out/r03d_rocprof_stats.md | cut -c1-200
bash profiles/baq_profile.sh r03d_idaq --idaq > /dev/null 2>&1; head -8 gpurun_out/r03d_idaq_rocprof_stats.md | cut -c1-200
python -m pytest tests/test_gpu_shard.py tests/test_gpu_configs.py tests/test_gpu_stability.py tests/test_gpu_uniq.py -q -x > gpurun_out/r03d/new_tests.log 2>&1; echo "new tests rc=$?"; tail -30 gpurun_out/r03d/new_tests.log | cut -c1-300
