#!/bin/bash
# counters of the bf16 3x3 kernel on tools/time_conv3x3_bf16.py (run through gpurun), separate passes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_c3
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_c3 -- python $ROOT/tools/time_conv3x3_bf16.py 16 > /tmp/pmc_c3.log 2>&1
  python - <<PY
import csv, glob, collections
fs = glob.glob("/tmp/pmc_c3/*/*counter_collection.csv")
if not fs:
    print("no counters for: $set"); raise SystemExit
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "k_conv3x3_bf16" in r["Kernel_Name"]:
        agg[(r["Counter_Name"], r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
big = max((int(k[1]) for k in agg if k[1].isdigit()), default=0)
print({k[0]: round(sum(v) / len(v)) for k, v in agg.items() if k[1] == str(big)}, "grid", big)
PY
done
