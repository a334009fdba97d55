#!/bin/bash
# rocprofv3 evidence for bench.py's roofline, collected on THE SAME command bench.py's default run uses (same build,
# same graph, same kernels): one --kernel-trace --stats pass (per-kernel durations) and one --pmc pass per counter
# group (kernel-trace only, as gpurun requires; FETCH_SIZE and WRITE_SIZE cannot share a pass -- TCC has 4 slots).
#   usage (on the GPU box, from the repo root):  bash tools/pmc_bench.sh <tag> [bench.py args, e.g. --workload gat]
# writes gpurun_out/<tag>/{kernel_stats.csv, bench_under_rocprof.json, pmc_<group>.csv}; tools/pmc_parse.py turns the
# counter files into profiles/traffic.json entries (stamped with the build they were collected on).
set -u
# every profiled run is bounded (PMC_TIMEOUT seconds, default 900): a counter pass that hangs (seen once: rmat27, WRITE_SIZE) must not
# hold the box until gpurun's own limit
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc}
shift || true
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# --other-workloads off: under rocprofv3 the program must not start child processes (the default line's other_workloads are children)
BENCH_ARGS="--full-line --steps 3 --warmup 2 --no-cpu-baseline --calibrate --other-workloads off $*"
rm -rf /tmp/kt_$TAG
if [ -z "${PMC_ONLY:-}" ]; then
# the kernel-trace pass runs bench.py's default timed region only (no calibration launches, no extra graphs), so that the
# AverageNs of each kernel instantiation is directly the figure bench.py's HIP events report for that launch kind
timeout -k 30 ${PMC_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o p -- python3 $R/bench.py --full-line --steps 10 --warmup 3 --no-cpu-baseline --no-extra-graphs --other-workloads off $* > $OUT/bench_under_rocprof.json 2> $OUT/kernel_trace.log
rc=$?
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pmc_bench.sh: the kernel-trace pass overran PMC_TIMEOUT (exit $rc): stopping, no counter passes" >&2; exit $rc; fi
f=$(find /tmp/kt_$TAG -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv
fi
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  if [ -n "${PMC_ONLY:-}" ] && [ "$grp" != "$PMC_ONLY" ]; then continue; fi
  tag=$(echo $grp | tr ' ' '_')
  rm -rf /tmp/pmc_${TAG}_$tag
  timeout -k 30 ${PMC_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${TAG}_$tag -o p -- python3 $R/bench.py $BENCH_ARGS --no-extra-graphs > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.log
  rc=$?
  # a pass that overran was killed (TERM, then KILL after 30 s: -k); its profiled child may still hold the GPU, so the remaining passes
  # would run next to it and read skewed counters -- stop here instead
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pmc_bench.sh: counter pass $grp overran PMC_TIMEOUT (exit $rc): remaining passes skipped" >&2; ls -la $OUT; exit $rc; fi
  f=$(find /tmp/pmc_${TAG}_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|spmm_csr_kernel|spmm_csr_flat_kernel|spmm_rowslot|spmm_long_finalize|gat2_kernel|gat_long_finalize|gat_fwd_kernel|gat_bwd|gemm_bf16|gradw_" $f > $OUT/pmc_$tag.csv
done
ls -la $OUT
