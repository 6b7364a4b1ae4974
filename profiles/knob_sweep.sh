#!/bin/bash
# The -m gpu suite under the A/B knobs that select fallback paths (the default configuration never takes them on the
# test shapes).  From the repo root on the GPU box: bash profiles/knob_sweep.sh
for kv in "LFQ_SPLIT_POOL_CELLS=0" "LFQ_SPLIT_POOL_CELLS=60000" "LFQ_LIGHT_KERNEL=wave" \
          "LFQ_SCREEN_ROUNDS=1" "LFQ_SINGLE_STREAM=1" "LFQ_SEG_BUDGET_MID=1 LFQ_SEG_BUDGET_BIG=1" "LFQ_SEG_MAX=2" \
          "LFQ_COUNT_MULTI_BELOW=0" "LFQ_COUNT_MULTI_BELOW=1000000" "LFQ_NO_SB_PRECOMPUTE=1" "LFQ_HOST_THREADS=1" \
          "LFQ_BAQ_LDS=0" "LFQ_BAQ_SCRATCH_MB=64" "LFQ_PILEUP_ATOMIC=1" "LFQ_INDEL_HOST_PACK=1"; do
  r=$(env $kv python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "$kv: $r"
done
