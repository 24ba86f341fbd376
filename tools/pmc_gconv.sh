#!/bin/bash
# SQ / TA / TCP counters of k_gconv3x3 on one layer shape (ONLY="layer3 s1"), separate --pmc passes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/pmc
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pmc_gc$i
  PMC=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_gc$i -- python $ROOT/tools/time_gconv.py > /tmp/pmc_gc$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_gc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_gconv3x3" in r["Kernel_Name"] and "Cijk" not in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$ROOT/gpurun_out/pmc/gconv_pmc.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s %.4g" % (c, v))
PY
