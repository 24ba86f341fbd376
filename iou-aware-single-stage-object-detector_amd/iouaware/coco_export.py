"""Detections -> COCO result json: the on-disk format downstream of the hot path
(reference mmdet/core/evaluation/coco_utils.py:78-149 `xyxy2xywh`,
`proposal2json`, `det2json`, `results2json`).  Pure host code on the per-class ndarray lists
produced by `bbox2result`; pycocotools itself (evaluation) is out of scope.

Signatures follow the reference: functions take a `dataset` object of which only
`len(dataset)`, `dataset.img_ids` and `dataset.cat_ids` are read (coco_utils.py:89-113);
`ResultIndex` is the smallest such object for callers that have no CocoDataset.
Pinned on reference-generated json: tests/golden/e2e_*.npz (`coco_json`)."""
import json

import numpy as np


class ResultIndex(object):
    """img_ids: image id per result; cat_ids: category id per class index."""

    def __init__(self, img_ids, cat_ids):
        self.img_ids, self.cat_ids = list(img_ids), list(cat_ids)

    def __len__(self):
        return len(self.img_ids)


def xyxy2xywh(bbox):
    """[x1, y1, x2, y2] -> [x, y, w, h] with the reference's +1 width convention."""
    b = np.asarray(bbox).tolist()
    return [b[0], b[1], b[2] - b[0] + 1, b[3] - b[1] + 1]


def proposal2json(dataset, results):
    """results: list over images of (k,5) arrays; category 1 (coco_utils.py:88-100)."""
    out = []
    for idx in range(len(dataset)):
        bboxes = results[idx]
        for i in range(bboxes.shape[0]):
            out.append(dict(image_id=dataset.img_ids[idx], bbox=xyxy2xywh(bboxes[i]),
                            score=float(bboxes[i][4]), category_id=1))
    return out


def det2json(dataset, results):
    """results: list over images of lists over classes of (k,5) arrays (coco_utils.py:103-113)."""
    out = []
    for idx in range(len(dataset)):
        img_id = dataset.img_ids[idx]
        result = results[idx]
        for label in range(len(result)):
            bboxes = result[label]
            for i in range(bboxes.shape[0]):
                out.append(dict(image_id=img_id, bbox=xyxy2xywh(bboxes[i]),
                                score=float(bboxes[i][4]), category_id=dataset.cat_ids[label]))
    return out


def results2json(dataset, results, out_file):
    """dispatch on the result type and dump to `out_file` itself (coco_utils.py:140-149; the
    caller appends '.json', tools/test.py:187).  Mask results are outside the IoU-aware path."""
    if isinstance(results[0], list):
        js = det2json(dataset, results)
    elif isinstance(results[0], tuple):
        raise NotImplementedError('segm results: mask heads are outside the IoU-aware path')
    elif isinstance(results[0], np.ndarray):
        js = proposal2json(dataset, results)
    else:
        raise TypeError('invalid type of results')
    with open(out_file, 'w') as f:
        json.dump(js, f)
    return out_file
