#!/bin/bash
# interleaved A/B of the round's loading stage: before (zero-copy reduction, one loading stream, 128-workgroup uploads) against the defaults
# (uncached rows staged by 20 workgroups, two alternating streams, 16-workgroup uploads), 50 % cache and the reference's capacity rule
run() { timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline --mb-cache-frac $2 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 cache $2', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','loader_host_ms_per_batch','cache_miss_rate')})" || tail -3 /tmp/mb_err.log; }
for rep in 1 2 3; do
  for frac in 0.5 -1 0.25; do
    DGLL_LOADER_STAGE_MISSES=0 DGLL_LOADER_STREAMS=1 DGLL_LOADER_UPLOAD_BLOCKS=128 run "before  " $frac
    run "defaults" $frac
  done
done
