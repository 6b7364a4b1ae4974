# Round 5, lean count kernel (decision planes only, 15 instructions per 8 observations): parity first, then the bench lines.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r05_lean_gpu_tests.txt
tail -3 gpurun_out/r05_lean_gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_lean_bench_driver_form.json 2> gpurun_out/r05_lean_bench_driver_form.err
tail -c 1500 gpurun_out/r05_lean_bench_driver_form.json
rm -f gpurun_out/r05_lean_other.jsonl
for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 >> gpurun_out/r05_lean_other.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_lean_other.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:40], d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"), d["config"].get("kernel_ms"))
PY
