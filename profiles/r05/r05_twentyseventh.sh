# Round 5: C4 / C5 with the read sets resident in HBM when the timed region starts (the measurement rule), the PCIe-inclusive rate beside it
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --config C4 2> gpurun_out/r05_c4.err | tail -1 > gpurun_out/r05_bench_c4_line.json
python bench.py --config C5 2> gpurun_out/r05_c5.err | tail -1 > gpurun_out/r05_bench_c5_line.json
tail -3 gpurun_out/r05_c4.err gpurun_out/r05_c5.err
python - <<'PY'
import json
for f in ("c4","c5"):
    d=json.loads(open("gpurun_out/r05_bench_%s_line.json"%f).read())
    c=d["config"]
    print(f, d["ms_per_step"], d["value"], c["ms_per_step_with_upload"], c["value_with_upload"], c["vcf_sha256"][:16], c["vcf_lines"], (d["roofline"] or {}).get("avg_launch_ms"), (d.get("cpu_baseline") or {}).get("value"))
PY
python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_configs.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
