cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_chain.py tests/test_gpu_plpindel.py tests/test_gpu_pileup.py tests/test_gpu_configs.py tests/test_gpu_srcq.py -q -x 2>&1 | tail -3
python bench.py --mode chain --steps 600 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('chain ms', round(d['ms_per_step'],2), {k:round(c[k]*1e3,1) for k in c if k.startswith('s_') and isinstance(c[k], float)}, c['snv_records'], c['indel_tests'])"
python bench.py --mode chain --steps 600 --workers 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('chain 2 workers ms', round(d['ms_per_step'],2))"
