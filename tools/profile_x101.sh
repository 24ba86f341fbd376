#!/bin/bash
# per-step kernel summary of the X-101-64x4d configuration (BASELINE config 4)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
cat > /tmp/x101.py <<PY
import sys, os
sys.path.insert(0, os.path.join("$ROOT", "iou-aware-single-stage-object-detector_amd")); sys.path.insert(0, "$ROOT")
import torch, bench, iouaware
from iouaware.config import ConfigDict
from iouaware.fuse import fuse_inference
torch.backends.cudnn.benchmark = True
from iouaware import ops
ops.gemm_tuning('all')
cfg = ConfigDict(bench.MODEL); cfg.backbone.update(dict(type='ResNeXt', depth=101, groups=64, base_width=4))
torch.manual_seed(0)
m = iouaware.build_detector(cfg, test_cfg=ConfigDict(bench.TEST_CFG)).cuda().eval()
fuse_inference(m, winograd=True)
m = m.to(memory_format=torch.channels_last)
x = torch.randn(8, 3, 800, 1344, device='cuda').contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(int(sys.argv[1])):
        m.simple_test_device(x, bench.metas(8), rescale=True)
torch.cuda.synchronize()
PY
python /tmp/x101.py 3 > /dev/null 2>&1
rm -rf /tmp/px
rocprofv3 --kernel-trace --output-format csv -d /tmp/px -- python /tmp/x101.py 7 > /tmp/px.log 2>&1
python $ROOT/tools/summarize_trace.py /tmp/px/*/*_kernel_trace.csv --steps 3 --marker "ia::k_lazy_greedy" --top 30 > /tmp/x101_sum.txt; mkdir -p $ROOT/gpurun_out/profile; cp /tmp/x101_sum.txt $ROOT/gpurun_out/profile/x101_step_summary.txt; (head -12; tail -10) < /tmp/x101_sum.txt | cut -c1-150
