"""Load a reference-format checkpoint (mmcv.runner.load_checkpoint semantics,
used at reference tools/test.py:160 and resnet.py:481-484): a torch file that
is either a bare state_dict or a dict with a 'state_dict' entry (+ 'meta');
a leading 'module.' (DataParallel) prefix is stripped; non-strict by default,
mismatches are reported, not fatal."""
import torch


def load_state_dict(module, state_dict, strict=False, logger=None):
    own = module.state_dict()
    unexpected, mismatched = [], []
    for name, value in state_dict.items():
        if name not in own:
            unexpected.append(name)
            continue
        if own[name].shape != value.shape:
            mismatched.append((name, tuple(own[name].shape), tuple(value.shape)))
            continue
        own[name].copy_(value)
    missing = sorted(set(own) - set(state_dict))
    report = []
    if unexpected:
        report.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        report.append('missing keys in source state_dict: ' + ', '.join(missing))
    for name, a, b in mismatched:
        report.append('size mismatch for %s: model %s vs checkpoint %s' % (name, a, b))
    if report:
        msg = '\n'.join(report)
        if strict:
            raise RuntimeError(msg)
        if logger is not None:
            logger.warning(msg)
    return dict(missing=missing, unexpected=unexpected, mismatched=mismatched)


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None,
                    allow_pickle=False):
    """Reference checkpoints hold tensors and a plain 'meta' dict, so the safe unpickler
    (weights_only=True) is the default; allow_pickle=True opts into torch's full unpickler for
    files that carry other Python objects -- only for files you trust."""
    if filename.startswith(('modelzoo://', 'open-mmlab://', 'http://', 'https://')):
        raise IOError('no network in this build: download %s yourself and pass a local path'
                      % filename)
    ckpt = torch.load(filename, map_location=map_location or 'cpu',
                      weights_only=not allow_pickle)
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        sd = ckpt['state_dict']
    elif isinstance(ckpt, dict):
        sd = ckpt
    else:
        raise RuntimeError('No state_dict found in checkpoint file {}'.format(filename))
    if sd and all(k.startswith('module.') for k in sd):
        sd = {k[7:]: v for k, v in sd.items()}
    target = model.module if hasattr(model, 'module') else model
    load_state_dict(target, sd, strict, logger)
    return ckpt
