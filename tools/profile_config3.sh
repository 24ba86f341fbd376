#!/bin/bash
# BASELINE config 3 per-GPU shape: R-101, bf16, 16 images of 800x1344, fused channels-last path.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
cat > /tmp/c3.py <<PY
import sys, os, time
sys.path.insert(0, os.path.join("$ROOT", 'iou-aware-single-stage-object-detector_amd')); sys.path.insert(0, "$ROOT")
import torch, bench, iouaware
from iouaware.config import ConfigDict
from iouaware.fuse import fuse_inference
torch.backends.cudnn.benchmark = True
from iouaware import ops
ops.gemm_tuning('all')
cfg = ConfigDict(bench.MODEL); cfg.backbone.update(dict(depth=101))
torch.manual_seed(0)
m = iouaware.build_detector(cfg, test_cfg=ConfigDict(bench.TEST_CFG)).cuda().eval()
fuse_inference(m, winograd=True)
m = m.to(memory_format=torch.channels_last).to(torch.bfloat16)
B = 16
x = torch.randn(B, 3, 800, 1344, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
metas = bench.metas(B)
with torch.no_grad():
    for _ in range(3): out = m.simple_test_device(x, metas, rescale=True)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): out = m.simple_test_device(x, metas, rescale=True)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
print('R-101 bf16 B=16: %.1f ms/step  %.1f img/s' % (dt * 1e3, B / dt))
PY
python /tmp/c3.py > /tmp/c3warm.log 2>&1; tail -1 /tmp/c3warm.log
rm -rf /tmp/pc3
rocprofv3 --kernel-trace --output-format csv -d /tmp/pc3 -- python /tmp/c3.py > /tmp/c3.log 2>&1
tail -1 /tmp/c3.log
mkdir -p $ROOT/gpurun_out/profile
python $ROOT/tools/summarize_trace.py /tmp/pc3/*/*_kernel_trace.csv --steps 4 --marker "k_lazy_greedy" --top 40 > $ROOT/gpurun_out/profile/config3_step_summary.txt
head -44 $ROOT/gpurun_out/profile/config3_step_summary.txt | cut -c1-160
