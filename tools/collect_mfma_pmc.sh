#!/bin/bash
# MFMA utilisation of the library GEMMs / MIOpen convolutions inside bench.py steps (run through
# gpurun).  One --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA pipe is busy, summed over
# SIMDs) and GRBM_GUI_ACTIVE (busy cycles of the kernel), --kernel-trace only.
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128)   (SQ: summed over 1024 SIMDs; GRBM: over 8 XCDs;
#   cross-checked against the achieved TFLOP/s of the head GEMM: 0.78 vs 125/157 = 0.80)
# Output: gpurun_out/pmc/mfma_pmc.json (per kernel name, averaged over its launches).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
python $ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-train --no-other-configs > /dev/null 2>&1      # warm MIOpen's find db
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- \
    python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train --no-other-configs > /tmp/pmc_mfma.log 2>&1
mkdir -p $ROOT/gpurun_out/pmc
python - <<PY
import csv, glob, json, collections
f = glob.glob("/tmp/pmc_mfma/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
rows = list(csv.DictReader(open(f)))
# only the dispatches of the last three steps (marker: the step's last kernel): the first steps carry
# MIOpen's find trials and the per-shape candidate timing of the library GEMMs
marks = sorted({int(r["Dispatch_Id"]) for r in rows if "k_finalize(" in r["Kernel_Name"]})
lo, hi = (marks[-4], marks[-1]) if len(marks) >= 4 else (-1, 1 << 62)
for r in rows:
    if lo < int(r["Dispatch_Id"]) <= hi:
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    mf, ga = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(d["GRBM_GUI_ACTIVE"])
    if mf <= 0:
        continue
    out[k] = {"launches": len(d["GRBM_GUI_ACTIVE"]), "mfma_busy_cycles": mf, "gui_active_cycles": ga,
              "mfma_util": mf / (ga * 128.0) if ga else None}
res = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128), summed over the launches of the last three steps of bench.py --steps 3 --warmup 2",
       "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles"]))}
json.dump(res, open("$ROOT/gpurun_out/pmc/mfma_pmc.json", "w"), indent=1)
for k, v in list(res["kernels"].items())[:14]:
    print("%-92s n=%3d util=%.3f" % (k, v["launches"], v["mfma_util"] or 0))
PY
