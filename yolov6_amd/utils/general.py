"""Box helpers (reference yolov6/utils/general.py:32-86).  Host-side tensor algebra used by
callers of the hot path (loss code, tests); the device kernels fuse the same arithmetic."""
import torch


def dist2bbox(distance, anchor_points, box_format='xyxy'):
    lt, rb = torch.split(distance, 2, -1)
    x1y1, x2y2 = anchor_points - lt, anchor_points + rb
    if box_format == 'xyxy':
        return torch.cat([x1y1, x2y2], -1)
    if box_format == 'xywh':
        return torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1)
    raise ValueError(box_format)


def bbox2dist(anchor_points, bbox, reg_max):
    x1y1, x2y2 = torch.split(bbox, 2, -1)
    return torch.cat([anchor_points - x1y1, x2y2 - anchor_points], -1).clip(0, reg_max - 0.01)


def xywh2xyxy(bboxes):
    bboxes[..., 0] = bboxes[..., 0] - bboxes[..., 2] * 0.5
    bboxes[..., 1] = bboxes[..., 1] - bboxes[..., 3] * 0.5
    bboxes[..., 2] = bboxes[..., 0] + bboxes[..., 2]
    bboxes[..., 3] = bboxes[..., 1] + bboxes[..., 3]
    return bboxes
