set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r05_gpu_tests_a.txt
cat gpurun_out/r05_gpu_tests_a.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err
tail -c 3000 gpurun_out/r05_bench_a.json
