#!/bin/bash
# Collect HBM traffic counters of the head kernels on an MI355X (run through gpurun).
# Separate --pmc passes (FETCH_SIZE uses 3 of 4 TCC slots, WRITE_SIZE 2), --kernel-trace only,
# as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes.  Output: gpurun_out/pmc/head_pmc.json
#   traffic_bytes = 2 * FETCH_SIZE(KB) * 1024   (gfx950: FETCH_SIZE reads exactly 1/2 of a wide
#                                                coalesced stream; MI355X_MICROARCH.md "HBM")
#                 +     WRITE_SIZE(KB) * 1024   (uncalibrated per the guide; reported as counted)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
B=${1:-8}; KIND=${2:-D}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- \
      python $ROOT/tools/time_head.py $B $KIND 5 > /dev/null 2>&1
done
mkdir -p $ROOT/gpurun_out/pmc
python - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ia::" in r["Kernel_Name"] and r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][c + "_KB_per_launch"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
for k, d in out.items():
    d["traffic_bytes_per_launch"] = int(2 * d.get("FETCH_SIZE_KB_per_launch", 0) * 1024
                                        + d.get("WRITE_SIZE_KB_per_launch", 0) * 1024)
res = {"batch": $B, "inputs": "$KIND", "workload": "tools/time_head.py (head kernels, 800x1344)",
       "correction": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts 1/2)",
       "stage_kernels": [k for k in out if k.startswith(("ia::k_rowmax_filter", "ia::k_sel_final", "ia::k_gather_nhwc"))],
       "note": "stage_kernels = the launches of ia_decode_stage / ia_get_bboxes' decode stage (SURVEY 8d); "
               "tools/time_head.py also runs the separate kernels of the stage-wise C-ABI",
       "kernels": out}
json.dump(res, open("$ROOT/gpurun_out/pmc/head_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
