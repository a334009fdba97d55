for rep in 1 2; do
for bpc in 1 2 3 4 6 8; do
  for n in 1 2; do
    DGLL_LOADER_STREAMS=$n timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline --mb-cache-frac 0.5 --mb-loader-blocks-per-cu $bpc 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks/cu $bpc streams $n', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','loader_host_ms_per_batch')})"
  done
done
done
