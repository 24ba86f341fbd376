"""Host-side anchor generation (reference mmdet/core/anchor/anchor_generator.py:4-84).

Only the 9 base anchors per level are ever materialised on the inference path:
the HIP kernels regenerate `base[a] + (x*stride, y*stride, x*stride, y*stride)`
from the anchor index.  grid_anchors / valid_flags exist for the training-target
code and for API parity; unlike the reference (default device 'cuda', :53,:72)
they take the device explicitly and default to CPU.
"""
import torch


class AnchorGenerator(object):
    def __init__(self, base_size, scales, ratios, scale_major=True, ctr=None):
        self.base_size = base_size
        self.scales = torch.as_tensor(scales, dtype=torch.float32).clone()
        self.ratios = torch.as_tensor(ratios, dtype=torch.float32).clone()
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        """ratio-major x scale when scale_major; centre 0.5*(base-1); torch.round
        (half to even) -- the values are integral, e.g. stride 8:
        [-19,-7,26,14], [-25,-10,32,17], ... (anchor_generator.py:18-43)."""
        w = h = float(self.base_size)
        if self.ctr is None:
            cx, cy = 0.5 * (w - 1), 0.5 * (h - 1)
        else:
            cx, cy = self.ctr
        hr = torch.sqrt(self.ratios)
        wr = 1 / hr
        if self.scale_major:
            ws = (w * wr[:, None] * self.scales[None, :]).reshape(-1)
            hs = (h * hr[:, None] * self.scales[None, :]).reshape(-1)
        else:
            ws = (w * self.scales[:, None] * wr[None, :]).reshape(-1)
            hs = (h * self.scales[:, None] * hr[None, :]).reshape(-1)
        half_w, half_h = 0.5 * (ws - 1), 0.5 * (hs - 1)
        return torch.stack([cx - half_w, cy - half_h, cx + half_w, cy + half_h], dim=-1).round()

    def grid_anchors(self, featmap_size, stride=16, device='cpu'):
        """(H*W*A, 4), x fastest: row (y*W + x)*A + a  (anchor_generator.py:53-70)."""
        base = self.base_anchors.to(device)
        fh, fw = int(featmap_size[0]), int(featmap_size[1])
        sx = (torch.arange(0, fw, device=device) * stride).to(base.dtype)
        sy = (torch.arange(0, fh, device=device) * stride).to(base.dtype)
        xx = sx.repeat(fh)
        yy = sy.view(-1, 1).expand(fh, fw).reshape(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1)
        return (base[None, :, :] + shifts[:, None, :]).reshape(-1, 4)

    def valid_flags(self, featmap_size, valid_size, device='cpu'):
        """uint8 (H*W*A): anchor positions inside the un-padded part of the feature map."""
        fh, fw = int(featmap_size[0]), int(featmap_size[1])
        vh, vw = int(valid_size[0]), int(valid_size[1])
        if not (vh <= fh and vw <= fw):
            raise AssertionError('valid size exceeds feature map')
        vx = torch.zeros(fw, dtype=torch.uint8, device=device)
        vy = torch.zeros(fh, dtype=torch.uint8, device=device)
        vx[:vw] = 1
        vy[:vh] = 1
        valid = (vx.repeat(fh) & vy.view(-1, 1).expand(fh, fw).reshape(-1))
        return valid[:, None].expand(valid.size(0), self.num_base_anchors).reshape(-1)
