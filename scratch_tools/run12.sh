cd $GRAFT_REPO_ROOT
python bench.py --mode host-abi --steps 200 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('host-abi', round(d['value']/1e6,2),'M cols/s', round(c['ms_per_batch'],2),'ms', round(c['effective_GBps'],1),'GB/s', round(c['frac_of_pcie'],3), c['records'])"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stability.py tests/test_gpu_indel.py tests/test_gpu_uniq.py -q -x 2>&1 | tail -3
