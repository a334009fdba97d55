#!/bin/bash
run() { timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','loader_host_ms_per_batch')})" || tail -3 /tmp/mb_err.log; }
for rep in 1 2; do
  for ub in 16 32; do
    for n in 2 3; do
      for sb in 16 20 24 32; do
        DGLL_LOADER_UPLOAD_BLOCKS=$ub DGLL_LOADER_STREAMS=$n DGLL_LOADER_STAGE_BLOCKS=$sb run "all hops staged: upload $ub wg, stage $sb wg, $n streams"
      done
    done
  done
done
