cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -q -s > gpurun_out/r03b/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03b/pytest.log
grep -E "passed|failed|FAILED|rc=|deep tail|C3 full|log p" gpurun_out/r03b/pytest.log | cut -c1-600
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03b/bench.json 2> gpurun_out/r03b/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats'])
print(json.dumps(d['config'].get('vcf_concordance'))[:1200])
PY
