# Round 4: everything the committed profiles/r04_* files come from.  From the repo root on the GPU box:
#     bash profiles/r04_collect.sh        (writes gpurun_out/r04_*; copy what is to be kept into profiles/)
set -u
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r04_other_configs.jsonl
bash profiles/profile.sh r04 > /dev/null 2>&1
bash profiles/run_pmc.sh r04 > gpurun_out/r04_run_pmc.log 2>&1
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err
tail -c 1200 gpurun_out/r04_bench_line.json
for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 >> gpurun_out/r04_other_configs.jsonl
done
python bench.py --config C4 2>/dev/null | tail -1 > gpurun_out/r04_bench_c4_line.json
python bench.py --config C5 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_line.json
cat gpurun_out/r04_bench_c4_line.json gpurun_out/r04_bench_c5_line.json >> gpurun_out/r04_other_configs.jsonl
python bench.py --mode host-abi --steps 100 2>/dev/null | tail -1 >> gpurun_out/r04_other_configs.jsonl
python bench.py --mode chain --steps 400 2>/dev/null | tail -1 >> gpurun_out/r04_other_configs.jsonl
python bench.py --mode chain --steps 800 --overlap-regions 2>/dev/null | tail -1 >> gpurun_out/r04_other_configs.jsonl
python bench.py --mode chain --steps 600 --workers 2 2>/dev/null | tail -1 >> gpurun_out/r04_other_configs.jsonl
python bench.py --mode baq --steps 100 2>/dev/null | tail -1 >> gpurun_out/r04_other_configs.jsonl
python bench.py --mode baq --steps 100 --idaq 2>/dev/null | tail -1 >> gpurun_out/r04_other_configs.jsonl
python profiles/other_configs_md.py gpurun_out/r04_other_configs.jsonl r04 > gpurun_out/r04_other_configs.md 2>/dev/null
bash profiles/baq_profile.sh r04_baq > /dev/null 2>&1
bash profiles/baq_pmc.sh r04_baq > gpurun_out/r04_baq_pmc.log 2>&1
# kernel timeline of one region of the reads -> VCF chain, and how busy the GPU is in a C4 run
out=$R/gpurun_out/prof_r04_chain; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $R/bench.py --mode chain --steps 300 > $out/bench.log 2>&1)
python profiles/chain_timeline.py $out > gpurun_out/r04_chain_timeline.txt 2>&1
out=$R/gpurun_out/prof_r04_c4; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $out -o trace -- python $R/bench.py --config C4 --steps 4 --warmup 1 > $out/bench.log 2>&1)
python profiles/gpu_busy.py $out 0.6 > gpurun_out/r04_c4_gpu_busy.md 2>&1
ls gpurun_out | grep r04
# kernel timelines of one C2 and one C3 step (no pipelining: one step's kernels at a time)
for cfg in C2 C3; do
  out=$R/gpurun_out/prof_r04_tl_$cfg; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $out -o t -- python $R/bench.py --config $cfg --steps 6 --warmup 3 --repeats 1 --no-pipeline --no-cpu-baseline --no-pmc --no-secondary --no-full-check > /dev/null 2>&1)
  python profiles/timeline.py $out > gpurun_out/r04_timeline_$cfg.txt 2>&1
done
# the shared-wavefront count kernel: counters at C2, rate against the size of the launch
CTRS="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS FETCH_SIZE WRITE_SIZE" bash profiles/pmc_kernels.sh r04shallow lfq_count --config C2 > gpurun_out/r04_count_shallow_pmc.md 2>&1
bash profiles/count_vs_columns.sh > gpurun_out/r04_count_vs_columns.txt 2>&1
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" > gpurun_out/r04_gpu_tests.txt
ls gpurun_out | grep r04
