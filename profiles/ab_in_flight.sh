# one batch in flight / two (the second one's count kernel gated on the first one's DP tail) / two, ungated -- the last
# variant needs a build with the A/B switch LFQ_NO_TAIL_WAIT in tail_wait() (lfq_api.hip), which was removed after this run
for cfg in "--config C2" "--cols 3750000 --depth 200" "--cols 4600000 --depth 500" "--config C3 --steps 60"; do
  for v in "1 X=0" "2 X=0" "2 LFQ_NO_TAIL_WAIT=1"; do
    set -- $v
    env $2 python bench.py $cfg --in-flight $1 --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['kernel_ms']; r=d['repeats']; print('$cfg', 'in-flight $1 $2', round(d['ms_per_step'],3), round(r['ms_per_step_min'],3), round(r['ms_per_step_max'],3), 'count', round(k['ms_count'],3), 'dp', round(k['ms_dp'],3), d['config']['records_per_step'])"
  done
done
