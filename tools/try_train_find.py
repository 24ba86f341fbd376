import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'iou-aware-single-stage-object-detector_amd'))
from iouaware import train_fuse, winograd_train
train_fuse.WGRAD = os.environ.get('WGRAD', train_fuse.WGRAD)
winograd_train.DU = os.environ.get('DU', winograd_train.DU)
t = time.time()
r = bench.train_record(torch.device('cuda', 0), find=bool(int(sys.argv[1])), loss_part=not os.environ.get('TRAIN_ONLY'), channels_last=bool(os.environ.get('CL')), fuse=not os.environ.get('NOFUSE'))
print('benchmark', sys.argv[1], 'total %.1f s' % (time.time() - t), r['value'], 'img/s', r['ms_per_iter'], 'ms/iter loss part', r['loss_part_ms'])
