cd $GRAFT_REPO_ROOT
LFQ_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --cols 250000 --steps 20 --warmup 3 --no-secondary 2> gpurun_out/two_ranks.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['n_gpus'], d['value'], d['ms_per_step'], c['pipeline'], c['rccl_ranks'], c['exchange_backend'], c['records_per_step'], d['repeats']['ms_per_step_min'], d['repeats']['ms_per_step_max'])"
tail -3 gpurun_out/two_ranks.err
LFQ_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --cols 250000 --steps 20 --warmup 3 --scaling strong --no-secondary 2> gpurun_out/two_ranks_s.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['n_gpus'], d['scaling'], d['value'], d['ms_per_step'], c['pipeline'], c['records_per_step'])"
tail -3 gpurun_out/two_ranks_s.err
python bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('C2', d['value'], d['ms_per_step'], c['pipeline'], c['vcf_identical'], c['columns_compared'], c['max_dlogp_device_vs_80bit_truth'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['kernel_alone'])"
