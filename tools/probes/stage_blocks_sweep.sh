#!/bin/bash
# the fetch of the uncached rows (stage_rows_kernel): workgroups (= PCIe reads in flight) x loading streams x kernel arguments in device memory
run() { timeout -k 30 400 python bench.py --full-line --workload minibatch --no-cpu-baseline 2>/tmp/mb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', {k:round(d[k],3) for k in ('batches_per_s','gpu_side_ms_per_batch','gpu_side_ms_per_batch_p95','loader_host_ms_per_batch')})" || tail -3 /tmp/mb_err.log; }
for rep in 1 2; do
  for ka in 0 1; do
    if [ $ka = 1 ]; then export HIP_FORCE_DEV_KERNARG=1; else unset HIP_FORCE_DEV_KERNARG; fi
    DGLL_LOADER_STAGE_MISSES=0 DGLL_LOADER_STREAMS=1 run "dev-kernarg $ka zero-copy reduction, 1 stream "
    for n in 1 2; do
      for sb in 8 16 32 64 128; do
        DGLL_LOADER_STAGE_MISSES=1 DGLL_LOADER_STREAMS=$n DGLL_LOADER_STAGE_BLOCKS=$sb run "dev-kernarg $ka staged, $sb workgroups, $n stream(s)"
      done
    done
  done
done
