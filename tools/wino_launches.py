"""per-launch durations / algorithmic GB/s of the Winograd transforms inside one bench step"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import torch, bench
from iouaware import winograd
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda', 0)
model = bench.build_model(dev, channels_last=True)
imgs = torch.randn(bench.BATCH, 3, bench.PAD_H, bench.PAD_W, device=dev).contiguous(memory_format=torch.channels_last)
st = bench.Stepper(model, imgs, 1)
for _ in range(3): st.step()
winograd.TIMING = []
st.step(); torch.cuda.synchronize()
rec = [(k, e0.elapsed_time(e1) * 1e3, b) for k, e0, e1, b in winograd.TIMING]
winograd.TIMING = None
tot = 0
for k, us, b in rec:
    tot += us
    print('%-4s %8.1f us  %7.1f MB  %6.0f GB/s' % (k, us, b / 1e6, b / us / 1e3))
print('total %.1f us' % tot)
