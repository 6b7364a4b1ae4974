# A/B of library builds on the shallow count kernel: count kernel ms and step ms on C2 / 200x / 500x per variant
for L in "$@"; do
  for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500"; do
    LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$L python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernel_ms']; print('$L', '$cfg', 'count', round(k['ms_count'],4), 'dp', round(k['ms_dp'],3), 'step', round(d['ms_per_step'],3), 'min', round(d['repeats']['ms_per_step_min'],3), 'frac', round(d['roofline']['frac'],3))"
  done
done
