# Round 6: everything the committed profiles/r06_* files come from.  From the repo root on the GPU box:
#     bash profiles/r06_collect.sh        (writes gpurun_out/r06_*; copy what is to be kept into profiles/)
set -u
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r06_other_configs.jsonl
# the driver's form of the run first (fresh box, --steps 20 --warmup 5): the line that is quoted; it carries C2 / C4 / C5 as scalars
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line_driver_form.json 2> gpurun_out/r06_bench_driver_form.err ) 2> gpurun_out/r06_bench_driver_form.time
tail -c 400 gpurun_out/r06_bench_line_driver_form.json; tail -3 gpurun_out/r06_bench_driver_form.time
# kernel stats: the mode the warm-up picks (eight batches queued without a gate), and batch after batch (gate end) for a clean timeline
bash profiles/profile.sh r06 --no-other-configs > /dev/null 2>&1
bash profiles/profile.sh r06_gate_end --in-flight 4 --gate end --no-other-configs > /dev/null 2>&1
for m in "r06" "r06_gate_end"; do
  python profiles/step_gaps.py gpurun_out/prof_$m 4 > gpurun_out/${m}_step_gaps.txt 2>&1
done
{ echo "# C3, eight batches queued, no gate (what the warm-up picks): kernel trace of profiles/profile.sh r06"; cat gpurun_out/r06_step_gaps.txt;
  echo; echo "# the same with gate end (batch after batch on the device)"; cat gpurun_out/r06_gate_end_step_gaps.txt;
  echo; echo "# timeline of one step, gate end (ms from the start of its count kernel)"; python profiles/timeline.py gpurun_out/prof_r06_gate_end; } > gpurun_out/r06_timeline_C3.txt 2>&1
bash profiles/run_pmc.sh r06 > gpurun_out/r06_run_pmc.log 2>&1
python bench.py --no-other-configs > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
tail -c 400 gpurun_out/r06_bench_line.json
python bench.py --config C2 --steps 60 --warmup 5 --no-secondary 2>/dev/null | tail -1 > gpurun_out/r06_bench_c2_line.json
python bench.py --config C4 2>/dev/null | tail -1 > gpurun_out/r06_bench_c4_line.json
python bench.py --config C5 2>/dev/null | tail -1 > gpurun_out/r06_bench_c5_line.json
cat gpurun_out/r06_bench_c2_line.json >> gpurun_out/r06_other_configs.jsonl
for cfg in "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
  python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 >> gpurun_out/r06_other_configs.jsonl
done
cat gpurun_out/r06_bench_c4_line.json gpurun_out/r06_bench_c5_line.json >> gpurun_out/r06_other_configs.jsonl
python bench.py --mode host-abi --steps 100 2>/dev/null | tail -1 >> gpurun_out/r06_other_configs.jsonl
python bench.py --mode chain --steps 400 2>/dev/null | tail -1 >> gpurun_out/r06_other_configs.jsonl
python bench.py --mode chain --steps 800 --overlap-regions 2>/dev/null | tail -1 >> gpurun_out/r06_other_configs.jsonl
python bench.py --mode baq --steps 100 2>/dev/null | tail -1 >> gpurun_out/r06_other_configs.jsonl
python bench.py --mode baq --steps 100 --idaq 2>/dev/null | tail -1 >> gpurun_out/r06_other_configs.jsonl
python profiles/other_configs_md.py gpurun_out/r06_other_configs.jsonl r06 > gpurun_out/r06_other_configs.md 2>/dev/null
bash profiles/baq_profile.sh r06_baq > /dev/null 2>&1
bash profiles/baq_pmc.sh r06_baq > gpurun_out/r06_baq_pmc.log 2>&1
# how busy the GPU is in a C4 run
out=$R/gpurun_out/prof_r06_c4; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $out -o trace -- python $R/bench.py --config C4 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate > $out/bench.log 2>&1)
python profiles/gpu_busy.py $out 1100 > gpurun_out/r06_c4_gpu_busy.md 2>&1
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|log p|records" > gpurun_out/r06_gpu_tests.txt
ls gpurun_out | grep r06
