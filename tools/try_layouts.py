"""Scratch A/B: whole inference step under NCHW vs channels-last convolutions (same process)."""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd')); sys.path.insert(0, ROOT)
import torch, bench
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
x = torch.randn(8, 3, 800, 1344, device=dev)
xcl = x.contiguous(memory_format=torch.channels_last)
def run(st, n=10, head_only=False):
    with torch.no_grad():
        for _ in range(3): st.step()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(n):
            if head_only: st.model.forward_head(st.imgs)
            else: st.step()
        torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
a = bench.Stepper(bench.build_model(dev, fuse=True, channels_last=False), x, 1)
b = bench.Stepper(bench.build_model(dev, fuse=True, channels_last=True), xcl, 1)
for rep in range(3):
    print('NCHW  step %.2f ms  convs %.2f ms | NHWC  step %.2f ms  convs %.2f ms' % (
        run(a), run(a, head_only=True), run(b), run(b, head_only=True)))
