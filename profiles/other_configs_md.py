"""r0N_other_configs.jsonl (profiles/r02_collect.sh) -> the markdown table of profiles/r0N_other_configs.md.
    python profiles/other_configs_md.py gpurun_out/r02_other_configs.jsonl > profiles/r02_other_configs.md"""
import json
import sys

NAMES = ["C2 (10^6 columns x 1000)", "C5 shard (3.75 x 10^6 columns x 200)", "C4, SNV part (4.6 x 10^6 columns x 500)",
         "--mode host-abi (200 k columns x 1000 from host memory)",
         "--mode chain (regions of 2 M reads x 150 bp -> VCF, --call-indels, BAQ on; 1 step = 1 region)",
         "--mode chain --workers 2 (two region workers = processes on the one GPU)",
         "--mode chain --workers 3", "--mode baq (400 K reads x 150 bp)"]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
if rnd != "r02":            # round 3 on: one more chain line (one worker, regions overlapped) and the IDAQ run of --mode baq
    NAMES = NAMES[:5] + ["--mode chain --overlap-regions (one worker; region k + 1 started before region k is finished)"] \
            + NAMES[5:] + ["--mode baq --idaq"]
if rnd not in ("r02", "r03"):   # round 4 on: the depth-200 / depth-500 shapes are labelled as what they are, C4 / C5 are the genome runs
    NAMES = ["C2 (10^6 columns x 1000)", "custom shape: 3.75 x 10^6 columns x 200, SNV-only, resident tracks",
             "custom shape: 4.6 x 10^6 columns x 500, SNV-only, resident tracks",
             "--config C4 (4.6 Mb x 500x genome of reads -> VCF, --call-indels, 32 bins; 1 step = the genome)",
             "--config C5 (29 Mb of BED targets x 200x, reads -> VCF, 32 bins; 1 step = the genome)",
             "--mode host-abi (200 k columns x 1000 from host memory)",
             "--mode chain (regions of 2 M reads x 150 bp -> VCF, --call-indels, BAQ on; 1 step = 1 region)",
             "--mode chain --overlap-regions (one worker; region k + 1 started before region k is finished)",
             "--mode chain --workers 2 (two region workers = processes on the one GPU)",
             "--mode baq (400 K reads x 150 bp)", "--mode baq --idaq"]
if rnd not in ("r02", "r03", "r04", "r05"):     # round 6: no two-worker chain line
    NAMES = [n for n in NAMES if "--workers 2" not in n]
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
print("# bench.py on the other configurations (1 MI355X; round %s).  C2 / depth 200 / depth 500: --steps 60 --warmup 5 --no-cpu-baseline --no-pmc" % rnd[1:].lstrip("0"))
print("# --no-secondary, pipelined two-context loop; host-abi, chain, baq: the --mode runs.  The headline configuration (C3) is in %s_bench_line.json." % rnd)
print()
print("| run | ms/step | value | count ms | scan ms | DP ms (light / mid / big chains) |")
print("|---|---|---|---|---|---|")
for name, l in zip(NAMES, lines):
    d = json.loads(l)
    k = d.get("config", {}).get("kernel_ms") or {}
    def g(key):
        v = k.get(key)
        return "" if v is None else ("%.3f" % v if v < 0.2 else "%.2f" % v)
    dp = ""
    if k:
        dp = "%s (%s / %s / %s)" % (g("ms_dp"), g("ms_dp_light"), g("ms_dp_mid"), g("ms_dp_big"))
    print("| %s | %.2f | %.3g %s | %s | %s | %s |" % (name, d["ms_per_step"], d["value"], d["unit"], g("ms_count"), g("ms_scan"), dp))
print()
print("Raw JSON lines:")
print()
print("```")
for l in lines:
    print(l.rstrip())
print("```")
