#!/bin/bash
# counters of ONE variant of the bf16 3x3 kernel on the P3 map of config 3 (256 -> 256, 100 x 168, batch
# 16), separate passes (run through gpurun).  usage: pmc_conv3x3_variant.sh <name> [ENV=VALUE ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
cat > /tmp/one_conv3.py <<PY
import os, sys
sys.path.insert(0, os.path.join("$ROOT", 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import ops
B, H, W = 16, 100, 168
x = torch.randn(B, 256, H, W, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device='cuda') * 0.03).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
b = torch.randn(256, device='cuda')
wp = ops.conv3x3_bf16_pack(w)
for _ in range(30):
    ops.conv3x3_bf16(x, wp, b, 256, relu=True)
torch.cuda.synchronize()
PY
echo "== $name ($*)"
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_c3
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_c3 -- python /tmp/one_conv3.py > /tmp/pmc_c3.log 2>&1
  python - <<PY
import csv, glob, collections
fs = glob.glob("/tmp/pmc_c3/*/*counter_collection.csv")
if not fs:
    print("no counters for: $set"); raise SystemExit
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "k_conv3x3_bf16" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: round(sum(v[5:]) / max(1, len(v[5:]))) for k, v in agg.items()}
ks = glob.glob("/tmp/pmc_c3/*/*kernel_trace.csv")
if ks:
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(ks[0])) if "k_conv3x3_bf16" in r["Kernel_Name"]]
    out["avg_ns"] = round(sum(d[5:]) / max(1, len(d[5:])))
print(out)
PY
done
