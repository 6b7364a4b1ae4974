set -u
cd $GRAFT_REPO_ROOT
LFQ_BAQ_WAVES=2 timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_baq.py 2>&1 | tail -2
for w in 1 2; do
LFQ_BAQ_WAVES=$w python bench.py --mode baq --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves', $w, 'ms/call', d['ms_per_step'])"
done
out=$GRAFT_REPO_ROOT/gpurun_out/prof_baq2; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && LFQ_BAQ_WAVES=2 timeout 300 rocprofv3 --kernel-trace --stats -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode baq --steps 100 > $out/bench.log 2>&1)
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | tail -1)
python profiles/summarize_rocprof.py $db | grep -i "baq_reg" | cut -c1-170
