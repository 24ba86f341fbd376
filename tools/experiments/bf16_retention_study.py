"""How chaotic is the twin-retention measure of the bf16 full-size contract?  R-101 fixture at
800 x 1344: reference fp32 detections vs (a) torch's own bf16 evaluation, three runs, (b) the fused
bf16 path, for several IoU thresholds and rank cuts; and (c) the fused path with the input perturbed
by one bf16 ulp of noise (a different, equally valid rounding pattern)."""
import os, sys, copy
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np, torch
import test_gpu_e2e as T
import synth
from iouaware.fuse import fuse_inference
f = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_backbone_r101_full.npz'))
m = T._build(dict(depth=101))
with torch.no_grad():
    synth.e2e_fill_state(m.state_dict(), int(f['weight_seed']))
m = m.cuda()
x, meta = T._img(f, 'module')
want = T._split(f['result_cat'], f['result_counts'])
def ret(res):
    return [T._twin_retention(want, res, 0.3, t)[1] for t in (0.5, 0.7, 0.85)]
with torch.no_grad():
    eager = copy.deepcopy(m).to(torch.bfloat16)
    xb = x.to(torch.bfloat16)
    for i in range(3):
        print('torch bf16 run %d: twins at IoU > 0.5 / 0.7 / 0.85:' % i, ret(eager(return_loss=False, rescale=True, img=[xb], img_meta=[[meta]])), flush=True)
    del eager
    fuse_inference(m, winograd=True)
    mb = m.to(memory_format=torch.channels_last).to(torch.bfloat16)
    xc = xb.contiguous(memory_format=torch.channels_last)
    print('fused bf16:', ret(mb(return_loss=False, rescale=True, img=[xc], img_meta=[[meta]])), flush=True)
    g = torch.Generator(device='cuda').manual_seed(1)
    for i in range(4):
        noise = (torch.rand(x.shape, device='cuda', generator=g) - 0.5) * 2 ** -8 * x.abs().cuda()
        xn = (x.cuda() + noise).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        print('fused bf16, input + half a bf16 ulp of noise #%d:' % i, ret(mb(return_loss=False, rescale=True, img=[xn], img_meta=[[meta]])), flush=True)
