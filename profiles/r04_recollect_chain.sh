# after the last chain / genome changes of round 4: C4, C5 and the three chain modes again
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04b.jsonl
python bench.py --config C4 2>/dev/null | tail -1 > gpurun_out/r04_bench_c4_line.json
python bench.py --config C5 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_line.json
python bench.py --mode chain --steps 400 2>/dev/null | tail -1 >> gpurun_out/r04b.jsonl
python bench.py --mode chain --steps 800 --overlap-regions 2>/dev/null | tail -1 >> gpurun_out/r04b.jsonl
python bench.py --mode chain --steps 600 --workers 2 2>/dev/null | tail -1 >> gpurun_out/r04b.jsonl
python - <<'PY'
import json
for f in ("gpurun_out/r04_bench_c4_line.json", "gpurun_out/r04_bench_c5_line.json"):
    d = json.loads(open(f).read())
    print(f, round(d["ms_per_step"], 1), d["value"], d["config"]["vcf_sha256"][:10], d["config"]["host_threads_per_rank"])
for l in open("gpurun_out/r04b.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][-60:], round(d["ms_per_step"], 2), d["config"].get("reads_per_s"))
PY
