# A/B of library builds on whole steps: C2 and C3, step / count / DP span per variant
for L in "$@"; do
  for cfg in "--config C2 --steps 60" "--config C3 --steps 40"; do
    LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$L python bench.py $cfg --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernel_ms']; print('$L', '$cfg', 'count', round(k['ms_count'],3), 'dp', round(k['ms_dp'],3), 'chains', round(k['ms_dp_light'],3), round(k['ms_dp_mid'],3), round(k['ms_dp_big'],3), 'step', round(d['ms_per_step'],3), 'min', round(d['repeats']['ms_per_step_min'],3), 'recs', d['config']['records_per_step'])"
  done
done
