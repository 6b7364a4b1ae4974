cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_baq.py tests/test_gpu_plpindel.py tests/test_gpu_chain.py -q -x 2>&1 | tail -3
for L in liblofreq_amd.so; do
  cd /tmp; export TMPDIR=/tmp; out=$GRAFT_REPO_ROOT/gpurun_out/ab_$L; rm -rf $out; mkdir -p $out
  LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$L timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --mode baq --steps 100 $BAQ_ARGS > $out/log 2>&1
  cd $GRAFT_REPO_ROOT; echo "== $L"; python profiles/summarize_rocprof.py $(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1) | head -6 | tail -4 | cut -c1-140
done
