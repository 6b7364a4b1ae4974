# the shallow count kernel alone (no pipelining: nothing else on the device while it runs), per library build and shape:
# average launch duration from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  i=0
  for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
    i=$((i+1)); out=/tmp/ciso_${L}_$i; rm -rf $out; mkdir -p $out
    LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$L rocprofv3 --kernel-trace --stats -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py $cfg --steps 30 --warmup 3 --repeats 1 --no-pipeline --no-cpu-baseline --no-pmc --no-secondary --no-full-check > /dev/null 2>&1
    echo "== $L $cfg"
    python $GRAFT_REPO_ROOT/profiles/summarize_rocprof.py $(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1) | grep -E "lfq_count" | cut -c1-70,100-170
  done
done
