"""CPU: the host-side mirror of the reference's registry / module API
(SURVEY 8b): configs load unchanged, type names resolve, parameter names and
shapes equal the reference's (checkpoint compatibility), error behaviour of the
builder, and -- when the reference tree is present (build container) -- equal
conv outputs for equal weights."""
import glob
import json
import os

import numpy as np

import pytest
import torch
import torch.nn as nn

import iouaware
from iouaware import registry
from iouaware.config import Config, ConfigDict

REF_CFG_DIR = '/root/reference/configs/iou_aware_single_stage_detector'
HERE = os.path.dirname(os.path.abspath(__file__))

R50_MODEL = dict(
    type='RetinaNet', pretrained=None,
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                  frozen_stages=1, style='pytorch'),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
              add_extra_convs=True, num_outs=5),
    bbox_head=dict(type='IoUawareRetinaHead', num_classes=81, in_channels=256, stacked_convs=4,
                   feat_channels=256, octave_base_scale=4, scales_per_octave=3,
                   anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[8, 16, 32, 64, 128],
                   target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
                   loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                                 loss_weight=1.0),
                   loss_bbox=dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0)))
TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_thr=0.5),
                max_per_img=100)


def model_cfg(**backbone):
    m = ConfigDict(R50_MODEL)
    m.backbone.update(backbone)
    return m


def test_configdict_behaviour():
    c = ConfigDict(TEST_CFG)
    assert c.get('nms_pre', -1) == 1000 and c.get('missing', -1) == -1
    assert c.score_thr == 0.05 and c.nms.iou_thr == 0.5
    n = c.nms.copy()
    assert n.pop('type', 'nms') == 'nms' and 'type' in c.nms and 'type' not in n
    with pytest.raises(AttributeError):
        c.nope


def test_registry_errors():
    with pytest.raises(KeyError, match='is not in the backbone registry'):
        registry.build_backbone(dict(type='NoSuchNet'))
    with pytest.raises(TypeError):
        registry.build_backbone(dict(type=3))
    with pytest.raises(TypeError):
        registry.BACKBONES.register_module(int)
    with pytest.raises(KeyError, match='already registered'):
        registry.BACKBONES.register_module(iouaware.backbones.ResNet)
    seq = registry.build([dict(type='FocalLoss', use_sigmoid=True), dict(type='SmoothL1Loss')],
                         registry.LOSSES)
    assert isinstance(seq, nn.Sequential) and len(seq) == 2
    with pytest.raises(AssertionError):
        registry.build_loss(dict(type='FocalLoss', use_sigmoid=False))


def test_state_dict_matches_reference_names_and_shapes():
    want = json.load(open(os.path.join(HERE, 'golden', 'state_dict_keys.json')))
    variants = {
        'iou_aware_retinanet_r50_fpn_1x_4gpu': model_cfg(),
        'iou_aware_retinanet_r101_fpn_1x_4gpu': model_cfg(depth=101),
        'iou_aware_retinanet_x101_32x4d_fpn_1x_4gpu': model_cfg(type='ResNeXt', depth=101, groups=32,
                                                                base_width=4),
    }
    for name, cfg in variants.items():
        m = iouaware.build_detector(cfg, train_cfg=None, test_cfg=ConfigDict(TEST_CFG))
        got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert got == want[name], name


def test_x101_64x4d_backbone_builds():          # BASELINE config 4
    m = iouaware.build_detector(model_cfg(type='ResNeXt', depth=101, groups=64, base_width=4),
                                test_cfg=ConfigDict(TEST_CFG))
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 95.89) < 0.01
    conv2 = m.backbone.layer1[0].conv2
    assert conv2.groups == 64 and conv2.in_channels == 256


def test_init_and_freeze_semantics():
    torch.manual_seed(0)
    m = iouaware.build_detector(model_cfg(), test_cfg=ConfigDict(TEST_CFG))
    # zero_init_residual: last BN gamma of every bottleneck is 0 (resnet.py:498-503)
    assert float(m.backbone.layer2[1].bn3.weight.abs().sum()) == 0.0
    # retina_cls bias = -log(99)
    assert torch.allclose(m.bbox_head.retina_cls.bias, torch.full((720,), -4.59512))
    # frozen_stages=1: stem + layer1 not trainable, BN in eval even in train mode
    assert not m.backbone.conv1.weight.requires_grad
    assert not m.backbone.layer1[0].conv1.weight.requires_grad
    assert m.backbone.layer2[0].conv1.weight.requires_grad
    m.train()
    assert not m.backbone.layer3[0].bn1.training and not m.backbone.bn1.training
    assert m.bbox_head.cls_convs[0].conv.weight.requires_grad


def test_head_forward_shapes_and_level_count():
    m = iouaware.build_detector(model_cfg(), test_cfg=ConfigDict(TEST_CFG)).eval()
    with torch.no_grad():
        cls, reg, iou = m.forward_head(torch.randn(2, 3, 128, 160))
    assert [tuple(t.shape) for t in cls] == [(2, 720, 16, 20), (2, 720, 8, 10), (2, 720, 4, 5),
                                             (2, 720, 2, 3), (2, 720, 1, 2)]
    assert [t.shape[1] for t in reg] == [36] * 5 and [t.shape[1] for t in iou] == [9] * 5


def test_get_bboxes_refuses_cpu_tensors():
    """no CPU fallback: the post-conv path only exists as gfx950 kernels"""
    from iouaware._lib import IouAwareLibraryError
    m = iouaware.build_detector(model_cfg(), test_cfg=ConfigDict(TEST_CFG)).eval()
    meta = [dict(img_shape=(128, 160, 3), scale_factor=1.0, pad_shape=(128, 160, 3),
                 ori_shape=(128, 160, 3), flip=False)]
    with torch.no_grad():
        outs = m.forward_head(torch.randn(1, 3, 128, 160))
    with pytest.raises(IouAwareLibraryError, match='no CPU'):
        m.bbox_head.get_bboxes(*outs, None, None, meta, m.test_cfg, True)


def test_nms_ops_accept_and_reject_like_the_reference():
    """type errors like the reference's; CPU tensors / ndarrays are ACCEPTED like the reference's
    (nms_wrapper.py:27-45) -- staged to the ROCm device (tests/test_gpu_native_ops.py); on a host
    without a device they fail loudly: there is no CPU compute path in this build"""
    from iouaware import nms_op, _lib
    if not torch.cuda.is_available():
        for call in (lambda: nms_op.nms(torch.zeros(3, 5), 0.5),
                     lambda: nms_op.nms(np.zeros((3, 5), np.float32), 0.5),
                     lambda: nms_op.multiclass_nms(torch.zeros(3, 4), torch.ones(3, 3), 0.05,
                                                   dict(type='nms', iou_thr=0.5))):
            with pytest.raises(_lib.IouAwareLibraryError, match='no CPU compute path'):
                call()
    # empty inputs never reach a kernel (nms_wrapper.py:38-39, bbox_nms.py:58-60)
    d, i = nms_op.nms(torch.zeros(0, 5), 0.5)
    assert d.shape == (0, 5) and i.dtype == torch.long and i.numel() == 0
    with pytest.raises(ValueError):                       # nms_wrapper.py:66
        nms_op.soft_nms(torch.zeros(1, 5), 0.5, method='quadratic')
    with pytest.raises(TypeError):                        # nms_wrapper.py:59-62
        nms_op.soft_nms([1, 2, 3], 0.5)
    with pytest.raises(TypeError):
        nms_op.nms([1, 2, 3], 0.5)
    with pytest.raises(AttributeError):                   # getattr(nms_wrapper, type), bbox_nms.py:31
        nms_op.multiclass_nms(torch.zeros(1, 4), torch.zeros(1, 3), 0.05,
                              dict(type='hard_nms', iou_thr=0.5))


@pytest.mark.skipif(not os.path.isdir(REF_CFG_DIR), reason='reference tree absent (GPU box)')
def test_reference_configs_load_unchanged_and_convs_match():
    import sys
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    import ref_shim
    for f in sorted(glob.glob(REF_CFG_DIR + '/*.py')):
        cfg = Config.fromfile(f)
        cfg.model.pretrained = None              # reference tools/test.py:138
        m = iouaware.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        assert type(m).__name__ == 'RetinaNet'
        assert cfg.test_cfg.nms.iou_thr == 0.5 and cfg.dist_params.backend == 'nccl'
    # same weights -> identical backbone/FPN/head outputs on CPU (R-50 config)
    ref_shim.install()
    # the compat alias must not be what we import here: this is the real reference
    from mmdet.models import build_detector as ref_build
    rcfg = ref_shim.load_config(REF_CFG_DIR + '/iou_aware_retinanet_r50_fpn_1x_4gpu.py')
    rcfg.model['pretrained'] = None
    torch.manual_seed(1)
    ref = ref_build(rcfg.model, train_cfg=rcfg.train_cfg, test_cfg=rcfg.test_cfg).eval()
    # zero_init_residual makes the backbone nearly an identity; perturb so the test bites
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    cfg = Config.fromfile(REF_CFG_DIR + '/iou_aware_retinanet_r50_fpn_1x_4gpu.py')
    cfg.model.pretrained = None
    mine = iouaware.build_detector(cfg.model, test_cfg=cfg.test_cfg).eval()
    info = iouaware.checkpoint.load_state_dict(mine, ref.state_dict(), strict=True)
    assert not info['missing'] and not info['unexpected']
    x = torch.randn(1, 3, 96, 128)
    with torch.no_grad():
        a = ref.bbox_head(ref.extract_feat(x))
        b = mine.forward_head(x)
    for ta, tb in zip(a, b):
        for u, v in zip(ta, tb):
            assert torch.equal(u, v)


class _Opaque(object):
    """an arbitrary Python object inside a checkpoint: what the safe unpickler refuses"""


def test_checkpoint_loads_with_the_safe_unpickler_by_default(tmp_path):
    """reference checkpoints are tensors + a plain meta dict; anything else needs an
    explicit opt-in"""
    from iouaware import checkpoint
    m = iouaware.build_detector(model_cfg(), test_cfg=ConfigDict(TEST_CFG))
    sd = {'module.' + k: v + 1 for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    good = str(tmp_path / 'good.pth')
    torch.save(dict(state_dict=sd, meta=dict(epoch=12, iter=7, mmdet_version='0.6.0')), good)
    ck = checkpoint.load_checkpoint(m, good)
    assert ck['meta']['epoch'] == 12
    k0 = 'backbone.conv1.weight'
    assert torch.equal(m.state_dict()[k0], sd['module.' + k0])            # 'module.' stripped

    bad = str(tmp_path / 'bad.pth')
    torch.save(dict(state_dict=sd, meta=dict(obj=_Opaque())), bad)
    with pytest.raises(Exception):
        checkpoint.load_checkpoint(m, bad)
    assert checkpoint.load_checkpoint(m, bad, allow_pickle=True)['meta']['obj'] is not None


def test_multiclass_nms_wrapper_has_no_capacity_errors_left():
    """round 1-3 raised ValueError above IA_MAX_CANDIDATES boxes / IA_MAX_PER_IMG outputs; the
    reference takes any n and any max_num (bbox_nms.py:33-56), and so does the wrapper now (the
    per-class route, tests/test_gpu_native_ops.py).  What is left without a device: the loud
    failure of a build without a CPU compute path."""
    from iouaware import nms_op, _lib
    cfg = dict(type='nms', iou_thr=0.5)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.IouAwareLibraryError, match='no CPU compute path'):
            nms_op.multiclass_nms(torch.zeros(_lib.IA_MAX_CANDIDATES + 1, 4),
                                  torch.ones(_lib.IA_MAX_CANDIDATES + 1, 3), 0.05, cfg, 100)
    with pytest.raises(NotImplementedError):              # class-specific boxes: two-stage feature
        nms_op.multiclass_nms(torch.zeros(8, 12), torch.zeros(8, 4), 0.05, cfg, 10)


@pytest.mark.parametrize('name,backbone', [
    ('r101', dict(depth=101)),
    ('x101_32x4d', dict(type='ResNeXt', depth=101, groups=32, base_width=4)),
    ('x101_64x4d', dict(type='ResNeXt', depth=101, groups=64, base_width=4)),
])
def test_deeper_backbones_get_the_reference_fixture_weights(name, backbone):
    """the name-keyed weight fill gives this build's R-101 / ResNeXt-101 detectors bit for bit the
    state the reference detector had when tests/golden/e2e_backbone_*.npz was generated: same
    parameter / buffer names, shapes and order of creation"""
    import numpy as np
    import sys
    sys.path.insert(0, HERE)
    import synth
    f = np.load(os.path.join(HERE, 'golden', 'e2e_backbone_%s.npz' % name))
    torch.manual_seed(0)
    m = iouaware.build_detector(model_cfg(**backbone), test_cfg=ConfigDict(TEST_CFG)).eval()
    with torch.no_grad():
        synth.e2e_fill_state(m.state_dict(), int(f['weight_seed']))
    assert synth.checksum([v.numpy() for k, v in sorted(m.state_dict().items())]) == \
        int(f['weight_checksum'])


def test_training_fixture_frozen_and_trainable_sets():
    """tests/golden/train_e2e.npz: the parameters the REFERENCE left without a gradient
    (frozen_stages=1) and the ones it trained are exactly this build's, by name"""
    import numpy as np
    f = np.load(os.path.join(HERE, 'golden', 'train_e2e.npz'))
    torch.manual_seed(0)
    m = iouaware.build_detector(model_cfg(), test_cfg=ConfigDict(TEST_CFG)).train()
    frozen = sorted(k for k, p in m.named_parameters() if not p.requires_grad)
    trainable = [k for k, p in m.named_parameters() if p.requires_grad]
    assert frozen == sorted(f['frozen'].tolist())
    assert trainable == f['grad_names'].tolist()
    # norm_eval: every BatchNorm stays in eval mode under .train()
    assert all(not b.training for b in m.modules() if isinstance(b, nn.modules.batchnorm._BatchNorm))
