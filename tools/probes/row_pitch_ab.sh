#!/bin/bash
# interleaved A/B: host store / HBM cache rows at their natural pitch (1204 B at the Reddit width) against 128-byte / 16-byte aligned pitches
for rep in 1 2 3; do
  for frac in 0.5 -1; do
    for m in natural aligned; do
      if [ $m = natural ]; then unset DGLL_HOST_ROW_ALIGN DGLL_CACHE_ROW_ALIGN; else export DGLL_HOST_ROW_ALIGN=128 DGLL_CACHE_ROW_ALIGN=16; fi
      timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline --mb-cache-frac $frac 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cache $frac pitch $m', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','loader_host_ms_per_batch')})"
    done
  done
done
