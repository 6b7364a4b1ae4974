set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_chain3
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $R/bench.py --mode chain --steps 12 --warmup 3 > $out/bench.log 2>&1
python $R/profiles/chain_timeline.py $out > $R/gpurun_out/r03_chain_timeline.txt 2>&1
tail -1 $out/bench.log
cat $R/gpurun_out/r03_chain_timeline.txt | tail -80
cd $R
python bench.py --mode chain --steps 200 2>/dev/null | tail -1
