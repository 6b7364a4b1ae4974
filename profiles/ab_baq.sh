# A/B of library builds on the BAQ bench: prints kernel-inclusive wall ms per variant
for L in "$@"; do
  LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$L python bench.py --mode baq --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],3))"
done
