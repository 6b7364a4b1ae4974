cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -q > gpurun_out/r03f/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03f/pytest.log
tail -12 gpurun_out/r03f/pytest.log | cut -c1-300
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats']['ms_per_step_median'])
s=d['config']['secondary']
print({k:(v.get('columns_per_s'), v.get('ms_per_batch'), v.get('s_mean'), v.get('s_per_region'), v.get('nt_layout'), v.get('frac_of_pcie'), v.get('error')) for k,v in s.items()})
print(d['config']['vcf_concordance']['identical'], d['config']['vcf_concordance']['columns_compared'], d['config']['vcf_concordance']['records_compared'])
PY
