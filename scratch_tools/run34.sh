set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_chain4
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $R/bench.py --mode chain --steps 300 > $out/bench.log 2>&1
python $R/profiles/chain_timeline.py $out > $R/gpurun_out/r03b_chain_timeline.txt 2>&1
tail -62 $R/gpurun_out/r03b_chain_timeline.txt
