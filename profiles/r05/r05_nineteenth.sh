# Round 5: kernel traces of the queued run in its steady state, C3 and C2, no gate / gate end
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for cfg in C3 C2; do for gate in none end; do
  out=$R/gpurun_out/prof_ov_${cfg}_$gate; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $out -o t -- python $R/bench.py --config $cfg --in-flight 4 --gate $gate --steps 16 --warmup 4 --repeats 1 --no-cpu-baseline --no-pmc --no-secondary --no-full-check > $out/bench.log 2>&1)
  { echo "# $cfg, four batches queued, gate $gate"; python profiles/overlap_timeline.py $out 8; } > gpurun_out/r05_overlap_${cfg}_$gate.txt 2>&1
done; done
head -30 gpurun_out/r05_overlap_C3_none.txt
