cd /tmp && export TMPDIR=/tmp
for sk in "light" "light,big" "light,mid" "mid,big"; do
  out=/tmp/iso_$(echo $sk | tr , _); mkdir -p $out
  LFQ_DEBUG_SKIP=$sk rocprofv3 --kernel-trace --stats -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pipeline --no-cpu-baseline --no-pmc --no-secondary > /dev/null 2>&1
  echo "== skip $sk"
  python $GRAFT_REPO_ROOT/profiles/summarize_rocprof.py $(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1) | grep -E "lfq_dp|strand" | cut -c1-60,100-160
done
