set -u
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_plpindel.py tests/test_gpu_chain.py tests/test_gpu_configs.py tests/test_gpu_pileup.py 2>&1 | tail -12
out=$GRAFT_REPO_ROOT/gpurun_out/prof_tiles; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode chain --steps 300 > $out/bench.log 2>&1)
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1)
python profiles/summarize_rocprof.py $db | grep -i "tiles\|plp_indel\|baq_reg" | cut -c1-170
tail -1 $out/bench.log | cut -c1-260
