#!/bin/bash
# interleaved A/B of the mini-batch bench: batches written in place into static input sets (the default) against the copy path (--mb-copy-inputs)
for i in 1 2 3; do
  for m in inplace copy; do
    if [ $m = inplace ]; then F=""; else F="--mb-copy-inputs"; fi
    python bench.py --full-line --workload minibatch --no-cpu-baseline $F 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','consumer_host_ms_per_batch','loader_host_ms_per_batch')}, d['consumer_step'][:60], d['consumer_step'][-40:])"
    grep -i "capture of the sampled step failed" /tmp/mb_err.log | cut -c1-300
  done
done
