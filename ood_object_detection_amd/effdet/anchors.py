"""Anchors, box decode and detection generation on the HIP post-processing kernels.

Mirrors the reference's effdet/anchors.py: `get_feat_sizes` (:175), `Anchors` (:191-302),
`decode_box_outputs` (:51), `clip_boxes_xyxy` (:88), `generate_detections` (:95-172).
Anchor generation is init-time host arithmetic (numpy float64 -> float32, like the reference);
everything per-image runs in libeffdet_hip.so - there is no CPU fallback.

`AnchorLabeler` (:305-438) labels anchors for training with `effdet_label_anchors`.
"""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

MIN_CLASS_SCORE = -5.0
_DUMMY_DETECTION_SCORE = -1e5


def get_feat_sizes(image_size: Tuple[int, int], max_level: int):
    feat_size = tuple(image_size)
    feat_sizes = [feat_size]
    for _ in range(1, max_level + 1):
        feat_size = ((feat_size[0] - 1) // 2 + 1, (feat_size[1] - 1) // 2 + 1)
        feat_sizes.append(feat_size)
    return feat_sizes


class Anchors(nn.Module):
    """RetinaNet-style multiscale anchors; `boxes` buffer is [N,4] float32 yxyx."""

    def __init__(self, min_level, max_level, num_scales, aspect_ratios, anchor_scale, image_size: Tuple[int, int]):
        super().__init__()
        self.min_level = min_level
        self.max_level = max_level
        self.num_scales = num_scales
        self.aspect_ratios = aspect_ratios
        if isinstance(anchor_scale, Sequence):
            assert len(anchor_scale) == max_level - min_level + 1
            self.anchor_scales = anchor_scale
        else:
            self.anchor_scales = [anchor_scale] * (max_level - min_level + 1)
        assert isinstance(image_size, Sequence) and len(image_size) == 2
        assert image_size[0] % 2 ** max_level == 0, 'Image size must be divisible by 2 ** max_level (128)'
        assert image_size[1] % 2 ** max_level == 0, 'Image size must be divisible by 2 ** max_level (128)'
        self.image_size = tuple(image_size)
        self.feat_sizes = get_feat_sizes(image_size, max_level)
        self.config = self._generate_configs()
        self.register_buffer('boxes', self._generate_boxes())

    @classmethod
    def from_config(cls, config, img_size=None, min_level=0):
        size = config.image_size if img_size is None else (img_size, img_size)
        return cls(config.min_level + min_level, config.max_level, config.num_scales, config.aspect_ratios,
                   config.anchor_scale, size)

    def _generate_configs(self):
        fs = self.feat_sizes
        cfgs = {}
        for level in range(self.min_level, self.max_level + 1):
            stride = (fs[0][0] // fs[level][0], fs[0][1] // fs[level][1])
            cfgs[level] = [(stride, octave / float(self.num_scales), aspect, self.anchor_scales[level - self.min_level])
                           for octave in range(self.num_scales) for aspect in self.aspect_ratios]
        return cfgs

    def _generate_boxes(self):
        per_level = []
        for _, cfgs in self.config.items():
            per_cfg = []
            for stride, octave_scale, aspect, anchor_scale in cfgs:
                base_x = anchor_scale * stride[1] * 2 ** octave_scale
                base_y = anchor_scale * stride[0] * 2 ** octave_scale
                if isinstance(aspect, Sequence):
                    ax, ay = aspect[0], aspect[1]
                else:
                    ax = np.sqrt(aspect)
                    ay = 1.0 / ax
                half_x, half_y = base_x * ax / 2.0, base_y * ay / 2.0
                xs = np.arange(stride[1] / 2, self.image_size[1], stride[1])
                ys = np.arange(stride[0] / 2, self.image_size[0], stride[0])
                xv, yv = np.meshgrid(xs, ys)
                xv, yv = xv.reshape(-1), yv.reshape(-1)
                per_cfg.append(np.stack([yv - half_y, xv - half_x, yv + half_y, xv + half_x], axis=1)[:, None, :])
            per_level.append(np.concatenate(per_cfg, axis=1).reshape(-1, 4))
        return torch.from_numpy(np.vstack(per_level)).float()

    def get_anchors_per_location(self):
        return self.num_scales * len(self.aspect_ratios)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(t, what):
    if not (torch.is_tensor(t) and t.device.type == 'cuda'):
        raise RuntimeError('%s must be a GPU tensor: the post-processing path is HIP only' % what)


def batched_detections(cls_topk, box_topk, anchor_boxes, indices, classes, img_scale=None, img_size=None,
                       max_det_per_image: int = 100, soft_nms: bool = False, box_all=None):
    """All images at once: returns (det [B,max_det,6] zero padded, count [B] int32, keep_src [B,max_det] int32).
    box_all (extension): the box head's packed [B, N, 4] output; the regressions are then read through `indices`
    (box_topk may be None), so the top-k need not have waited for the box head."""
    lib = _lib.load()
    _need_cuda(cls_topk, 'cls_outputs')
    B, k = indices.shape
    dev = cls_topk.device
    bsrc = box_all if box_all is not None else box_topk
    if cls_topk.dtype not in (torch.float32, torch.bfloat16) or bsrc.dtype not in (cls_topk.dtype, torch.float32):
        raise RuntimeError('cls outputs must be float32 or bfloat16, box outputs the same dtype or float32')
    dt = 0 if cls_topk.dtype == torch.float32 else (3 if bsrc.dtype == torch.float32 else 1)     # | 2: float32 box regressions
    cls_topk, bsrc = cls_topk.contiguous(), bsrc.contiguous()
    indices, classes = indices.contiguous(), classes.contiguous()
    anchors = anchor_boxes.to(device=dev, dtype=torch.float32).contiguous()
    sc = sz = None
    if img_scale is not None:
        sc = img_scale.to(device=dev, dtype=torch.float32).reshape(B).contiguous()
    if img_size is not None and img_scale is not None:
        sz = img_size.to(device=dev, dtype=torch.float32).reshape(B, 2).contiguous()
    f32 = dict(device=dev, dtype=torch.float32)
    i32 = dict(device=dev, dtype=torch.int32)
    boxes, scores = torch.empty(B, k, 4, **f32), torch.empty(B, k, **f32)
    cls_i, src, count, maxc = torch.empty(B, k, **i32), torch.empty(B, k, **i32), torch.empty(B, **i32), torch.empty(B, **f32)
    det, det_count, keep_src = torch.empty(B, max_det_per_image, 6, **f32), torch.empty(B, **i32), torch.empty(B, max_det_per_image, **i32)
    st = _stream(cls_topk)
    tail = (anchors.data_ptr(), indices.data_ptr(), classes.data_ptr(),
            sc.data_ptr() if (sc is not None and sz is not None) else None, sz.data_ptr() if sz is not None else None, B, k,
            boxes.data_ptr(), scores.data_ptr(), cls_i.data_ptr(), src.data_ptr(), count.data_ptr(), maxc.data_ptr())
    if box_all is not None and (bsrc.dim() != 3 or bsrc.shape[0] != B or bsrc.shape[2] != 4):
        raise ValueError('box_all must be [B, N, 4]')
    n_gather = bsrc.shape[1] if box_all is not None else 0
    scp = sc.data_ptr() if sc is not None else None
    if not soft_nms:
        # hard NMS: decode + threshold + NMS of every image in one launch
        _lib.check(lib.effdet_detections_hard(st, dt, cls_topk.data_ptr(), bsrc.data_ptr(), n_gather, *tail, 0.3, max_det_per_image,
                                              scp, det.data_ptr(), det_count.data_ptr(), keep_src.data_ptr()), 'effdet_detections_hard')
        return det, det_count, keep_src
    if box_all is not None:
        _lib.check(lib.effdet_decode_threshold_gather(st, dt, cls_topk.data_ptr(), bsrc.data_ptr(), n_gather, *tail),
                   'effdet_decode_threshold_gather')
    else:
        _lib.check(lib.effdet_decode_threshold(st, dt, cls_topk.data_ptr(), bsrc.data_ptr(), *tail), 'effdet_decode_threshold')
    _lib.check(lib.effdet_nms_soft(st, boxes.data_ptr(), scores.data_ptr(), cls_i.data_ptr(), src.data_ptr(),
                                   count.data_ptr(), maxc.data_ptr(), B, k, 1, 0.5, 0.3, 0.001, max_det_per_image,
                                   scp, det.data_ptr(), det_count.data_ptr(), keep_src.data_ptr()), 'effdet_nms_soft')
    return det, det_count, keep_src


def generate_detections(cls_outputs, box_outputs, anchor_boxes, indices, classes,
                        img_scale: Optional[torch.Tensor], img_size: Optional[torch.Tensor],
                        max_det_per_image: int = 100, soft_nms: bool = False):
    """Single-image API of the reference (effdet/anchors.py:95): returns [n, 6] (unpadded)."""
    assert box_outputs.shape[-1] == 4
    assert anchor_boxes.shape[-1] == 4
    assert cls_outputs.shape[-1] == 1
    sc = None if img_scale is None else torch.as_tensor(img_scale, device=cls_outputs.device).reshape(1)
    sz = None
    if img_scale is not None and img_size is not None:
        sz = torch.as_tensor(img_size, device=cls_outputs.device).reshape(1, 2)
    det, count, _ = batched_detections(cls_outputs.reshape(1, -1), box_outputs.reshape(1, -1, 4), anchor_boxes,
                                       indices.reshape(1, -1), classes.reshape(1, -1), sc, sz,
                                       max_det_per_image=max_det_per_image, soft_nms=soft_nms)
    return det[0, :int(count[0].item())]


class AnchorLabeler(object):
    """Labeler for multiscale anchor boxes (reference: effdet/anchors.py:305-438) on `effdet_label_anchors`:
    IoU similarity -> ArgMaxMatcher(match_threshold, match_threshold, force_match_for_each_row) -> class targets
    (label - 1, background -1) and FasterRcnnBoxCoder box targets, unpacked per pyramid level."""

    def __init__(self, anchors, num_classes: int, match_threshold: float = 0.5):
        self.anchors = anchors
        self.match_threshold = match_threshold
        self.num_classes = num_classes
        self.indices_cache = {}

    def _unpack(self, cls_t, box_t):
        B = cls_t.shape[0]
        A = self.anchors.get_anchors_per_location()
        cls_out, box_out, count = [], [], 0
        for level in range(self.anchors.min_level, self.anchors.max_level + 1):
            h, w = self.anchors.feat_sizes[level]
            steps = h * w * A
            cls_out.append(cls_t[:, count:count + steps].reshape(B, h, w, A))
            box_out.append(box_t[:, count:count + steps].reshape(B, h, w, A * 4))
            count += steps
        return cls_out, box_out

    def batch_label_anchors(self, gt_boxes, gt_classes, filter_valid=True, task_cls=None):
        if task_cls is not None:
            raise NotImplementedError('task_cls relabelling (effdet/anchors.py:397-404) is not built')
        lib = _lib.load()
        boxes = self.anchors.boxes
        if boxes.device.type != 'cuda':
            raise RuntimeError('AnchorLabeler needs its Anchors on the GPU (no CPU fallback)')
        dev = boxes.device
        B = len(gt_boxes)
        assert B == len(gt_classes)
        lens = [int(b.shape[0]) for b in gt_boxes]
        Mmax = max(lens + [1])
        gb = torch.zeros(B, Mmax, 4, dtype=torch.float32, device=dev)
        gc = torch.full((B, Mmax), -1, dtype=torch.int64, device=dev)
        if sum(lens):
            # one scatter for the whole batch (the lengths are host values): rows i * Mmax + j <- the j-th box of image i
            rows = torch.tensor([i * Mmax + j for i, m in enumerate(lens) for j in range(m)], dtype=torch.int64).to(dev)
            allb = torch.cat([b.to(device=dev, dtype=torch.float32).reshape(-1, 4) for b, m in zip(gt_boxes, lens) if m], 0)
            allc = torch.cat([c.to(device=dev, dtype=torch.int64).reshape(-1) for c, m in zip(gt_classes, lens) if m], 0)
            gb.view(-1, 4)[rows] = allb
            gc.view(-1)[rows] = allc if filter_valid else allc.clamp(min=0)
        N = boxes.shape[0]
        cls_t = torch.empty(B, N, dtype=torch.int64, device=dev)
        box_t = torch.empty(B, N, 4, dtype=torch.float32, device=dev)
        npos = torch.empty(B, dtype=torch.float32, device=dev)
        nbytes = lib.effdet_label_anchors_workspace_bytes(B, Mmax, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        anchors_f = boxes.float().contiguous()
        _lib.check(lib.effdet_label_anchors(_stream(boxes), anchors_f.data_ptr(), gb.data_ptr(), gc.data_ptr(), B, Mmax, N,
                                            float(self.match_threshold), cls_t.data_ptr(), box_t.data_ptr(), npos.data_ptr(),
                                            None, ws.data_ptr(), nbytes), 'effdet_label_anchors')
        cls_out, box_out = self._unpack(cls_t, box_t)
        return cls_out, box_out, npos

    def label_anchors(self, gt_boxes, gt_classes, filter_valid=True):
        cls_out, box_out, npos = self.batch_label_anchors([gt_boxes], [gt_classes], filter_valid=filter_valid)
        return [c[0] for c in cls_out], [b[0] for b in box_out], npos[0]
