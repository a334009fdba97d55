#!/bin/bash
# PMC passes over the MFMA transform kernels (kernel-trace only, one counter group per run)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-mfma_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf /tmp/mp_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/mp_$i -o p -- python3 $R/tools/mfma_pmc.py > $OUT/run_$i.log 2>&1
  f=$(find /tmp/mp_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|gemm_bf16" $f > $OUT/pmc_$i.csv
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        k="res" if "res_kernel" in r["Kernel_Name"] else "nt4"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["dur_ns"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in agg.items():
    print("==",k)
    for c,vals in sorted(v.items()):
        print("   %-40s %.4g" % (c, sum(vals)/len(vals)))
PY
