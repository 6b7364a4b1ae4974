set -u
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
LFQ_TIMING=1 python bench.py --config C4 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/c4_timing.json 2> gpurun_out/c4_timing.err
grep "lfq timing" gpurun_out/c4_timing.err | sed 's/[0-9.]\+/N/g' | sort | uniq -c | sort -rn | head -20
echo ---
grep "lfq timing" gpurun_out/c4_timing.err | tail -60
out=$R/gpurun_out/prof_r05_c4; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $out -o trace -- python $R/bench.py --config C4 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline > $out/bench.log 2>&1)
python profiles/gpu_busy.py $out 0.6
