set -u
rm -f gpurun_out/r02_other_configs.jsonl
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/profile.sh r02 > /dev/null 2>&1
bash profiles/run_pmc.sh r02 > gpurun_out/r02_run_pmc.log 2>&1
python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench.err
tail -c 3000 gpurun_out/r02_bench_line.json
for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl
done
python bench.py --mode host-abi --steps 100 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl
python bench.py --mode chain --steps 300 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl
python bench.py --mode chain --steps 600 --workers 2 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl
python bench.py --mode chain --steps 600 --workers 3 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl
python bench.py --mode baq --steps 100 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl
ls gpurun_out | head -40
