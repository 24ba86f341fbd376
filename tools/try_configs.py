"""Scratch: the other BASELINE configurations run through the product path (build, fuse, forward,
post-processing): config 3 (bf16, 16 images), config 4 (X-101-64x4d), R-101."""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
sys.path.insert(0, ROOT)
import torch, bench, iouaware
from iouaware.config import ConfigDict
from iouaware.fuse import fuse_inference
torch.backends.cudnn.benchmark = True
from iouaware import ops
ops.gemm_tuning('all')
def run(name, backbone, B, dtype):
    cfg = ConfigDict(bench.MODEL); cfg.backbone.update(backbone)
    torch.manual_seed(0)
    m = iouaware.build_detector(cfg, test_cfg=ConfigDict(bench.TEST_CFG)).cuda().eval()
    fuse_inference(m, winograd=True)
    m = m.to(memory_format=torch.channels_last).to(dtype)
    x = torch.randn(B, 3, 800, 1344, device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
    metas = bench.metas(B)
    with torch.no_grad():
        for _ in range(3): out = m.simple_test_device(x, metas, rescale=True)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5): out = m.simple_test_device(x, metas, rescale=True)
        torch.cuda.synchronize(); dt = (time.time() - t) / 5
    print('%-28s B=%2d %s: %.1f ms/step  %.1f img/s  num %s' % (name, B, str(dtype)[6:], dt * 1e3, B / dt, out[3][:3].tolist()))
run('R-50 fp32', {}, 8, torch.float32)
run('R-50 bf16 (config 3)', {}, 16, torch.bfloat16)
run('R-101 fp32', dict(depth=101), 8, torch.float32)
run('X-101-64x4d fp32 (config 4)', dict(type='ResNeXt', depth=101, groups=64, base_width=4), 8, torch.float32)
run('R-50 fp32 B=1 (config 1)', {}, 1, torch.float32)
