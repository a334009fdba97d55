#!/bin/bash
# the grid cap of the loading stage's HBM kernels (hop gathers, outermost-hop reduction) beside the training step: workgroups per CU
for rep in 1 2; do
for bpc in 2 3 4 6 8; do
  for frac in 0.5 -1; do
    timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline --mb-cache-frac $frac --mb-loader-blocks-per-cu $bpc 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks/cu $bpc cache $frac', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95')})" || tail -3 /tmp/mb_err.log
  done
done
done
