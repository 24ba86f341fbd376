"""Detections -> COCO result json: the on-disk format downstream of the hot path
(reference mmdet/core/evaluation/coco_utils.py:77-113,139-149 `xyxy2xywh`,
`det2json`, `results2json`).  Pure host code on the per-class ndarray lists produced by
`bbox2result`; pycocotools itself (evaluation) is out of scope."""
import json

import numpy as np


def xyxy2xywh(bbox):
    """[x1, y1, x2, y2] -> [x, y, w, h] with the reference's +1 width convention."""
    b = np.asarray(bbox).tolist()
    return [b[0], b[1], b[2] - b[0] + 1, b[3] - b[1] + 1]


def det2json(img_ids, cat_ids, results):
    """img_ids: image id per result; cat_ids: category id per class index;
    results: list over images of lists over classes of (k,5) arrays."""
    out = []
    for img_id, result in zip(img_ids, results):
        for label, bboxes in enumerate(result):
            for i in range(bboxes.shape[0]):
                out.append(dict(image_id=img_id, bbox=xyxy2xywh(bboxes[i]),
                                score=float(bboxes[i][4]), category_id=cat_ids[label]))
    return out


def results2json(img_ids, cat_ids, results, out_file):
    """writes `<out_file>.bbox.json` like the reference and returns the path"""
    path = '{}.{}.json'.format(out_file, 'bbox')
    with open(path, 'w') as f:
        json.dump(det2json(img_ids, cat_ids, results), f)
    return path
