for kn in "X=0" "LFQ_COUNT_LPG8_BELOW=5000" "LFQ_COUNT_LPG4_BELOW=5000"; do
  for cfg in "--config C2" "--cols 4600000 --depth 500" "--cols 3750000 --depth 200"; do
    env $kn python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernel_ms']; print('$kn', '$cfg', 'count', round(k['ms_count'],4), 'frac', round(d['roofline']['frac'],3), d['roofline']['kernel'])"
  done
done
