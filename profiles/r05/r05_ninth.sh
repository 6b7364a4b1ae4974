set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_uniq_binding.py tests/test_gpu_baq.py -x -q -p no:cacheprovider 2>&1 | tail -4
python bench.py --config C4 > gpurun_out/r05_c4_a.json 2> gpurun_out/r05_c4_a.err; tail -c 2500 gpurun_out/r05_c4_a.json; tail -3 gpurun_out/r05_c4_a.err
python bench.py --config C5 > gpurun_out/r05_c5_a.json 2> gpurun_out/r05_c5_a.err; tail -c 2500 gpurun_out/r05_c5_a.json; tail -3 gpurun_out/r05_c5_a.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-check --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_alone'], d['roofline']['step'], d['dp']['span_ms'], d['dp']['valu_busy'])"
