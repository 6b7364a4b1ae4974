# Round 5: DP kernels that fit beside two 12-wavefront count workgroups (320 registers, two slots per SIMD): combine as a
# 256-thread workgroup, the unsplit-big kernel with four wavefronts, prep at <= 128 registers -- built on the box; parity, then
# count workgroups of 12 / 16 wavefronts, four batches queued without a gate
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_dp_slim.txt; : > $out
rm -f lofreq_amd/csrc/build/lfq_dp.o
make -C lofreq_amd/csrc EXTRA="-DLFQ_COMB_THREADS=256 -DLFQ_HEAVY_WAVES=4 -DLFQ_DP512_WAVES=4" 2>&1 | grep -i "error" >> $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py tests/test_gpu_knobs.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2; do for w in 12 16; do
  echo "== LFQ_COUNT_WAVES_PER_WG=$w (round $rep)" >> $out
  LFQ_COUNT_WAVES_PER_WG=$w python bench.py --in-flight 4 --gate none --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=c['kernel_ms']
print(d['ms_per_step'], d['repeats']['ms_per_step_median'], 'count', k['ms_count'], 'dp', k['ms_dp'])" >> $out
done; done
cat $out
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/prof_ov3; rm -rf $o; mkdir -p $o
(cd /tmp && export TMPDIR=/tmp && LFQ_COUNT_WAVES_PER_WG=12 timeout 300 rocprofv3 --kernel-trace -d $o -o t -- python $R/bench.py --in-flight 4 --gate none --steps 16 --warmup 4 --repeats 1 --no-cpu-baseline --no-pmc --no-secondary --no-full-check > $o/bench.log 2>&1)
python profiles/overlap_timeline.py $o 8 > gpurun_out/r05_overlap3_C3_wg12.txt 2>&1
cat gpurun_out/r05_overlap3_C3_wg12.txt
