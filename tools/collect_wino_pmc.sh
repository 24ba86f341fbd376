#!/bin/bash
# HBM traffic counters of the Winograd transforms on the head layer (tools/time_wino.py), separate
# --pmc passes as tools/collect_pmc.sh.  Output: gpurun_out/pmc/wino_pmc.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/wpmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/wpmc_$c -- \
      python $ROOT/tools/time_wino.py > /tmp/wpmc_$c.log 2>&1
done
tail -1 /tmp/wpmc_WRITE_SIZE.log
mkdir -p $ROOT/gpurun_out/pmc
python - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/wpmc_%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_wino" in r["Kernel_Name"] and r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][c + "_KB_per_launch"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
for k, d in out.items():
    d["traffic_bytes_per_launch"] = int(2 * d.get("FETCH_SIZE_KB_per_launch", 0) * 1024
                                        + d.get("WRITE_SIZE_KB_per_launch", 0) * 1024)
T, C = 11440, 512
res = {"workload": "tools/time_wino.py: head layer, batch 8, 800x1344, both towers (T = 11440 tiles, 512 channels)",
       "algorithmic_bytes_per_launch": (16 + 36) * 4 * T * C,
       "correction": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts 1/2)",
       "kernels": out}
json.dump(res, open("$ROOT/gpurun_out/pmc/wino_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
