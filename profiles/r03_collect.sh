# Round 3: everything the committed profiles/r03_* files come from.  From the repo root on the GPU box:
#     bash profiles/r03_collect.sh        (writes gpurun_out/r03_*; copy what is to be kept into profiles/)
set -u
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r03_other_configs.jsonl
bash profiles/profile.sh r03 > /dev/null 2>&1
bash profiles/run_pmc.sh r03 > gpurun_out/r03_run_pmc.log 2>&1
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err
tail -c 2500 gpurun_out/r03_bench_line.json
for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
done
python bench.py --mode host-abi --steps 100 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
python bench.py --mode chain --steps 400 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
python bench.py --mode chain --steps 800 --overlap-regions 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
python bench.py --mode chain --steps 600 --workers 2 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
python bench.py --mode chain --steps 600 --workers 3 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
python bench.py --mode baq --steps 100 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
python bench.py --mode baq --steps 100 --idaq 2>/dev/null | tail -1 >> gpurun_out/r03_other_configs.jsonl
python profiles/other_configs_md.py gpurun_out/r03_other_configs.jsonl r03 > gpurun_out/r03_other_configs.md 2>/dev/null
bash profiles/baq_profile.sh r03_baq > /dev/null 2>&1
bash profiles/baq_profile.sh r03_baq_idaq --idaq > /dev/null 2>&1
bash profiles/baq_pmc.sh r03_baq > gpurun_out/r03_baq_pmc.log 2>&1
# kernel timeline of one region of the reads -> VCF chain
out=$R/gpurun_out/prof_r03_chain; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $R/bench.py --mode chain --steps 300 > $out/bench.log 2>&1)
python profiles/chain_timeline.py $out > gpurun_out/r03_chain_timeline.txt 2>&1
ls gpurun_out | grep r03
