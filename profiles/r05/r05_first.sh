# round 5, first GPU session: parity of the new count kernels, then the overlap A/B
set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py tests/test_gpu_shim.py -x -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r05_first_tests.txt
for v in "LFQ_COUNT_PERSIST=5" "LFQ_COUNT_WAVES_PER_WG=16"; do
  env $v python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "not c3_full" 2>&1 | tail -3 >> gpurun_out/r05_first_tests.txt
done
cat gpurun_out/r05_first_tests.txt
bash profiles/ab_overlap.sh C3 60 > gpurun_out/r05_ab_overlap_C3.txt 2>&1
cat gpurun_out/r05_ab_overlap_C3.txt
bash profiles/ab_overlap.sh C2 100 > gpurun_out/r05_ab_overlap_C2.txt 2>&1
cat gpurun_out/r05_ab_overlap_C2.txt
