#!/bin/bash
# Copy the summaries of tools/pmc_bench.sh runs into profiles/ (tracked) and rebuild profiles/traffic.json.
#   usage: bash tools/collect_profiles.sh <sage-tag> <gat-tag> [round] [rmat27-tag] [f32-tag]      e.g.  r06p_sage r06p_gat r06 r06p_rmat r06p_f32
set -eu
SAGE=$1; GAT=$2; R=${3:-r03}; RMAT=${4:-}; F32=${5:-}
python tools/pmc_parse.py gpurun_out/$GAT --round $R --write | tail -8
python tools/pmc_parse.py gpurun_out/$SAGE --round $R --write | tail -6
cp gpurun_out/$SAGE/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp gpurun_out/$SAGE/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp gpurun_out/$GAT/kernel_stats.csv profiles/${R}_gat_kernel_stats.csv
cp gpurun_out/$GAT/bench_under_rocprof.json profiles/${R}_gat_bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum_TCC_MISS_sum; do
  cp gpurun_out/$SAGE/pmc_$c.csv profiles/${R}_pmc_$c.csv
  cp gpurun_out/$GAT/pmc_$c.csv profiles/${R}_gat_pmc_$c.csv
done
# the scaling tools' logs, the default line and the step trace of the same call (gpurun_out/<round>p/), when present
E=gpurun_out/${R}p
if [ -n "$RMAT" ]; then
  python tools/pmc_parse.py gpurun_out/$RMAT --round $R --write | tail -4
  cp gpurun_out/$RMAT/kernel_stats.csv profiles/${R}_rmat27_kernel_stats.csv
  cp gpurun_out/$RMAT/bench_under_rocprof.json profiles/${R}_rmat27_bench_under_rocprof.json
  for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum_TCC_MISS_sum; do cp gpurun_out/$RMAT/pmc_$c.csv profiles/${R}_rmat27_pmc_$c.csv; done
fi
if [ -n "$F32" ]; then      # the reference's own arithmetic (fp32 storage, dgll/__init__.py:1): bench.py --dtype f32
  python tools/pmc_parse.py gpurun_out/$F32 --round $R --write | tail -6
  cp gpurun_out/$F32/kernel_stats.csv profiles/${R}_f32_kernel_stats.csv
  cp gpurun_out/$F32/bench_under_rocprof.json profiles/${R}_f32_bench_under_rocprof.json
  for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum_TCC_MISS_sum; do cp gpurun_out/$F32/pmc_$c.csv profiles/${R}_f32_pmc_$c.csv; done
fi
for f in scaling_model.log gat_scaling_model.log rmat27_scaling_model.log scaling_trace_n8.log gat_trace_n4.log gat_trace_n1.log partition_stats.log step_trace.log; do
  [ -f $E/$f ] && grep -v "amdgpu.ids" $E/$f > profiles/${R}_$f
done
[ -f $E/bench_default.json ] && cp $E/bench_default.json profiles/${R}_bench_default.json
[ -f $E/bench_default_line.json ] && cp $E/bench_default_line.json profiles/${R}_bench_default_line.json
grep -c Cijk profiles/${R}_bench_kernel_stats.csv profiles/${R}_gat_kernel_stats.csv || true
