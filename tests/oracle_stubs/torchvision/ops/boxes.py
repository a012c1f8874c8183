"""Stub of torchvision.ops.boxes: hard NMS delegates to the oracle's restatement."""
import torch
from oracle.postprocess import batched_nms as _bnms, nms as _nms


def batched_nms(boxes, scores, idxs, iou_threshold: float):
    return _bnms(boxes, scores, idxs, iou_threshold)


def nms(boxes, scores, iou_threshold: float):
    return _nms(boxes, scores, iou_threshold)


def remove_small_boxes(boxes, min_size: float):
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    return torch.where((ws >= min_size) & (hs >= min_size))[0]
