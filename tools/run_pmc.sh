#!/bin/bash
# rocprofv3 PMC passes for the dominant kernel, one counter group per run (kernel-trace only, as gpurun requires).
# usage (on the GPU box, from the repo root): bash tools/run_pmc.sh <outdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$tag -o p -- python3 $R/tools/pmc_spmm.py --reps 3 $PMC_ARGS > $OUT/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|spmm_csr_kernel" $f > $OUT/pmc_$tag.csv
done
ls -la $OUT
