# Round 5: how busy the GPU is in a C4 run with resident read sets (and with the uploads, for comparison)
set -u
R=$GRAFT_REPO_ROOT
cd $R
for mode in "" "--upload-reads"; do
  tag=resident; [ -n "$mode" ] && tag=upload
  out=$R/gpurun_out/prof_r05_c4_$tag; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $out -o trace -- python $R/bench.py --config C4 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate $mode > $out/bench.log 2>&1)
  { echo "# C4 (--config C4 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate $mode, N = 1, four host threads, four distinct sets of reads) under rocprofv3 --kernel-trace; profiles/gpu_busy.py over the last 60 % of the trace"; echo; python profiles/gpu_busy.py $out 0.6; grep '^{' $out/bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench line of the traced run: ms_per_step', d['ms_per_step'])"; } > gpurun_out/r05_c4_gpu_busy_$tag.md 2>&1
done
head -24 gpurun_out/r05_c4_gpu_busy_resident.md | cut -c1-200; head -6 gpurun_out/r05_c4_gpu_busy_upload.md | cut -c1-200
