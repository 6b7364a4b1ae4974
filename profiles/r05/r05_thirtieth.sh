# Round 5: the IDAQ group beside the plain BAQ launches (LFQ_BAQ_IDAQ_BESIDE=1, default) against behind them (0), same box
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for k in 1 0; do
  echo "== LFQ_BAQ_IDAQ_BESIDE=$k (round $rep)"
  export LFQ_BAQ_IDAQ_BESIDE=$k
  python bench.py --config C4 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-upload-rate 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C4', d['ms_per_step'])"
  python bench.py --mode baq --steps 100 --idaq 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('baq --idaq', d['ms_per_step'])"
  python bench.py --mode chain --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chain', d['ms_per_step'])"
  python bench.py --mode chain --steps 400 --overlap-regions 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chain overlapped', d['ms_per_step'])"
done; done
