#!/bin/bash
# interleaved A/B at the 50 % cache: the outermost hop's uncached rows read zero-copy by its reduction (stage 0) against fetched into HBM
# ahead of it (stage 1), on one and two alternating loading streams
for rep in 1 2 3; do
  for st in 0 1; do
    for n in 1 2; do
      DGLL_LOADER_STAGE_MISSES=$st DGLL_LOADER_STREAMS=$n timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline $@ 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage $st streams $n', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','loader_host_ms_per_batch','cache_miss_rate')})" || tail -5 /tmp/mb_err.log
    done
  done
done
