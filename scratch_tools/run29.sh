set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_pileup.py tests/test_gpu_chain.py tests/test_gpu_plpindel.py tests/test_gpu_configs.py 2>&1 | tail -3
for lib in liblofreq_amd.so liblofreq_amd_nostore.so; do
out=$GRAFT_REPO_ROOT/gpurun_out/prof_tiles; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode chain --steps 300 > $out/bench.log 2>&1)
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1)
echo "== $lib"
python profiles/summarize_rocprof.py $db | grep -i "pileup_tiles" | cut -c1-160
done
