# Round 5, shared-wavefront count kernel with prefetch: chunks per lane and step (LFQ_COUNT_AHEAD), built on the box
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r05_shallow_ahead.txt; : > $out
for a in 2 3 6; do
  rm -f lofreq_amd/csrc/build/lfq_kernels.o
  make -C lofreq_amd/csrc EXTRA=-DLFQ_COUNT_AHEAD=$a 2>&1 | grep -i "error" >> $out
  echo "== LFQ_COUNT_AHEAD=$a" >> $out
  for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
    python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['workload'][:50], d['ms_per_step'], c['kernel_ms']['ms_count'], d['roofline']['kernel'], d['roofline']['frac'], (d['roofline'].get('kernel_alone') or {}).get('avg_launch_ms'))" >> $out
  done
done
cat $out
