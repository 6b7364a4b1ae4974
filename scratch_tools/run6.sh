cd $GRAFT_REPO_ROOT
for L in liblofreq_amd.so liblofreq_amd_fwd.so liblofreq_amd_fwdns.so; do
  cd /tmp; export TMPDIR=/tmp; out=$GRAFT_REPO_ROOT/gpurun_out/ab_$L; mkdir -p $out
  LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$L timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --mode baq --steps 100 > $out/log 2>&1
  cd $GRAFT_REPO_ROOT; echo "== $L"; python profiles/summarize_rocprof.py $(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1) | head -4 | tail -2 | cut -c1-140
done
python -m pytest tests/test_gpu_baq.py -q -x 2>&1 | tail -2
