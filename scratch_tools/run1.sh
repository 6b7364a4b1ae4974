set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q -k "c3_full or deep_tail or bench_gpus" -s > gpurun_out/r03a/new_tests.log 2>&1; echo "new tests rc=$?" >> gpurun_out/r03a/new_tests.log
tail -15 gpurun_out/r03a/new_tests.log
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03a/pytest.log
tail -8 gpurun_out/r03a/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03a/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats'])
print(json.dumps(d['config'].get('vcf_concordance'), indent=1)[:1500])
print(d['roofline']['frac'], d['config']['kernel_ms'])
PY
