#!/bin/bash
# interleaved A/B of the mini-batch bench: the in-place native loads on one loading stream against two / three alternating ones
# (DGLL_LOADER_STREAMS), at the 50 % cache and with the reference's capacity rule.   usage: bash tools/loader_streams_ab.sh [reps]
REPS=${1:-3}
for i in $(seq 1 $REPS); do
  for frac in 0.5 -1; do
    for n in 1 2 3; do
      DGLL_LOADER_STREAMS=$n timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline --mb-cache-frac $frac 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cache $frac streams $n', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','consumer_host_ms_per_batch','loader_host_ms_per_batch')})"
    done
  done
done
