set -u
cd $GRAFT_REPO_ROOT
for t in 1 64; do
export LFQ_PILEUP_TILES=$t
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_pileup.py tests/test_gpu_chain.py 2>&1 | tail -1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_tiles; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode chain --steps 300 > $out/bench.log 2>&1)
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1)
echo "== LFQ_PILEUP_TILES=$t"
python profiles/summarize_rocprof.py $db | grep -i "pileup_tiles" | cut -c1-160
done
