"""CPU: bbox2result (reference transforms.py:148-166) and the COCO json export
(coco_utils.py:77-113) -- the data formats right after the hot path."""
import json

import numpy as np
import torch

from iouaware.bbox import bbox2result
from iouaware.coco_export import det2json, results2json, xyxy2xywh


def test_bbox2result_splits_by_label():
    dets = torch.tensor([[1., 2., 11., 22., .9], [0., 0., 5., 5., .8], [3., 3., 9., 9., .7]])
    labels = torch.tensor([2, 0, 2])
    res = bbox2result(dets, labels, 81)
    assert len(res) == 80 and res[0].shape == (1, 5) and res[2].shape == (2, 5) and res[1].shape == (0, 5)
    assert res[2].dtype == np.float32 and np.allclose(res[2][1], [3, 3, 9, 9, .7])
    empty = bbox2result(torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), 81)
    assert len(empty) == 80 and all(r.shape == (0, 5) and r.dtype == np.float32 for r in empty)


def test_coco_json(tmp_path):
    assert xyxy2xywh([10., 20., 19., 39.]) == [10., 20., 10., 20.]     # +1 convention
    res = [bbox2result(torch.tensor([[1., 2., 11., 22., .9]]), torch.tensor([5]), 81)]
    js = det2json([42], list(range(1, 81)), res)
    assert js == [dict(image_id=42, bbox=[1., 2., 11., 21.], score=0.8999999761581421,
                       category_id=6)]
    path = results2json([42], list(range(1, 81)), res, str(tmp_path / 'out'))
    assert path.endswith('out.bbox.json') and json.load(open(path)) == js
