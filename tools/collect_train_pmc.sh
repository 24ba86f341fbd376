#!/bin/bash
# HBM traffic counters of the training-side kernels: k_relu_bwd_colsum on the R-50 activation
# shapes (tools/time_colsum.py) and the channels-last loss kernels (tools/time_headloss.py 4 nhwc),
# separate --pmc passes as tools/collect_pmc.sh.  Output: gpurun_out/pmc/train_pmc.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tpmc_a_$c /tmp/tpmc_b_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tpmc_a_$c -- python $ROOT/tools/time_colsum.py > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tpmc_b_$c -- python $ROOT/tools/time_headloss.py 4 nhwc > /dev/null 2>&1
done
mkdir -p $ROOT/gpurun_out/pmc
python - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for tag, pat in (("a", ("k_relu_bwd_colsum",)), ("b", ("k_focal_nhwc", "k_box_nhwc"))):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("/tmp/tpmc_%s_%s/*/*counter_collection.csv" % (tag, c))[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(p in r["Kernel_Name"] for p in pat) and r["Counter_Name"] == c:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k][c + "_KB_total"] = sum(v)
            out[k]["launches"] = len(v)
for k, d in out.items():
    d["traffic_bytes_total"] = int(2 * d.get("FETCH_SIZE_KB_total", 0) * 1024 + d.get("WRITE_SIZE_KB_total", 0) * 1024)
    d["traffic_bytes_per_launch"] = d["traffic_bytes_total"] // max(d["launches"], 1)
# algorithmic bytes of the same launches
shapes = [(4, 128, 100, 168), (4, 512, 100, 168), (4, 256, 50, 84), (4, 1024, 50, 84), (4, 512, 25, 42),
          (4, 2048, 25, 42), (4, 256, 100, 168), (4, 720, 100, 168), (4, 256, 25, 42)]
el = sum(b * c * h * w for b, c, h, w in shapes)
n_each = out.get("ia::k_relu_bwd_colsum", {}).get("launches", 0) // (2 * len(shapes)) if out.get("ia::k_relu_bwd_colsum") else 0
alg_colsum = n_each * el * 4 * (3 + 1)          # with mask: 3 floats / element, without: 1
pix = 4 * 22400
alg_focal_fwd, alg_focal_bwd = pix * 720 * 4, pix * 720 * 8
res = {"workload": "tools/time_colsum.py (9 activation shapes of the R-50 iteration, with / without ReLU mask) and "
                   "tools/time_headloss.py 4 nhwc (loss part, batch 4)",
       "correction": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts 1/2)",
       "algorithmic_bytes": {"k_relu_bwd_colsum_total": alg_colsum, "k_focal_nhwc<false>_per_launch": alg_focal_fwd,
                             "k_focal_nhwc<true>_per_launch": alg_focal_bwd},
       "kernels": out}
json.dump(res, open("$ROOT/gpurun_out/pmc/train_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
