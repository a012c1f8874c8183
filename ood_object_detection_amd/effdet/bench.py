"""Task wrappers `_post_process`, `DetBenchPredict`, `DetBenchTrain` (reference: effdet/bench.py).

`DetBenchPredict.forward` = EfficientDet forward -> top-k -> decode -> (soft-)NMS, all in HIP, plus the
per-anchor OOD scores gathered to the kept detections.  The fork's `generate_detections` returns ragged
per-image rows and then `torch.stack`s them (bench.py:76), which only works when every image yields the
same count; here the batch result is the zero-padded [B, max_det, 6] tensor and `last_count` holds the
number of valid rows per image (`ragged()` gives the reference's per-image views).
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import _lib
from .anchors import Anchors, batched_detections


def _post_process(cls_outputs: List[torch.Tensor], box_outputs: Optional[List[torch.Tensor]], num_levels: int,
                  num_classes: int, max_detection_points: int = 5000, anchor_max: Optional[torch.Tensor] = None):
    """Top-k over all class logits (effdet/bench.py:12-56); ties go to the lower flat index.

    `anchor_max` (extension): the [B, N] float32 per-anchor maximum logit the class head already produced
    (`model.ood_max_logit`); the select then only scans the anchors that can reach the top k.  Results are
    identical with and without it.  `box_outputs=None` (extension) skips the box gather (returned box tensor is None).

    Accepts the per-level [B, A*C, H, W] / [B, A*4, H, W] lists.  When they are the engine's own
    NHWC-backed views the concatenation is free; other tensors are packed with one copy.
    Returns (cls [B,k,1], box [B,k,4], indices [B,k] int64, classes [B,k] int64)."""
    lib = _lib.load()
    c0 = cls_outputs[0]
    if c0.device.type != 'cuda':
        raise RuntimeError('_post_process runs on the GPU only (no CPU fallback)')
    B = c0.shape[0]
    cls_all = _packed(cls_outputs, num_levels, num_classes)
    box_all = _packed(box_outputs, num_levels, 4) if box_outputs is not None else None
    # a bf16 model's box head writes float32 regressions (engine.py): the select kernel gathers rows of the logits' dtype only,
    # so mixed dtypes take the box rows through the returned indices afterwards
    box_late = box_all if (box_all is not None and box_all.dtype != cls_all.dtype) else None
    if box_late is not None:
        box_all = None
    n_anchors = cls_all.shape[1]
    k = max_detection_points
    dt = 0 if cls_all.dtype == torch.float32 else 1
    out_cls = torch.empty(B, k, 1, dtype=cls_all.dtype, device=c0.device)
    out_box = torch.empty(B, k, 4, dtype=cls_all.dtype, device=c0.device) if box_all is not None else None
    idx = torch.empty(B, k, dtype=torch.int64, device=c0.device)
    cls_id = torch.empty(B, k, dtype=torch.int64, device=c0.device)
    if anchor_max is not None:
        if anchor_max.dtype != torch.float32 or tuple(anchor_max.shape) != (B, n_anchors) or anchor_max.device != c0.device:
            raise RuntimeError('anchor_max must be a float32 [B, N] tensor on the logits\' device')
        anchor_max = anchor_max.contiguous()
    ws_bytes = lib.effdet_topk_workspace_bytes(B, n_anchors)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=c0.device)
    st = torch.cuda.current_stream(c0.device).cuda_stream
    _lib.check(lib.effdet_topk_select(st, dt, cls_all.data_ptr(), anchor_max.data_ptr() if anchor_max is not None else None,
                                      B, n_anchors, num_classes, box_all.data_ptr() if box_all is not None else None, k,
                                      out_cls.data_ptr(), out_box.data_ptr() if out_box is not None else None, idx.data_ptr(), cls_id.data_ptr(),
                                      ws.data_ptr(), ws_bytes), 'effdet_topk_select')
    if box_late is not None:
        out_box = torch.gather(box_late, 1, idx.unsqueeze(-1).expand(B, k, 4))
    return out_cls, out_box, idx, cls_id


def _packed(outs, num_levels, width):
    """[B, A*width, H, W] per level -> one contiguous [B, N, width] tensor (a view when possible)."""
    B = outs[0].shape[0]
    base = outs[0]._base if outs[0]._base is not None else None
    if base is not None and base.dim() == 3 and base.shape[0] == B and base.shape[2] == width and base.is_contiguous():
        n = sum(o.shape[1] * o.shape[2] * o.shape[3] for o in outs[:num_levels]) // width
        if base.shape[1] == n and all(o._base is base for o in outs[:num_levels]) and outs[0].data_ptr() == base.data_ptr():
            return base
    if outs[0].dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError('head outputs must be float32 or bfloat16')
    return torch.cat([outs[l].permute(0, 2, 3, 1).reshape(B, -1, width) for l in range(num_levels)], 1).contiguous()


import contextlib
import threading

_fork = threading.local()


@contextlib.contextmanager
def forked_stream(stream):
    """`with forked_stream(s):` = `with torch.cuda.stream(s):` that also records that the code inside runs on a side stream
    forked from the caller's.  DetBenchPredict uses it for its concurrent sub-batches; callers that run a DetBenchPredict on
    their own side stream inside a hipGraph capture should use it too.  Why: forking again from a stream that is itself a
    fork of the capturing stream segfaults in hipStreamEndCapture on ROCm 7.2 (round-1 record gpurun_out/b2.log: the box
    head forked onto a third stream inside a captured half-batch) - a crash `except` cannot catch - so DetBenchPredict
    refuses to fork at depth > 0 while the stream is capturing."""
    _fork.depth = getattr(_fork, 'depth', 0) + 1
    try:
        with torch.cuda.stream(stream):
            yield
    finally:
        _fork.depth -= 1


class DetBenchPredict(nn.Module):
    """`streams` (extension, default: automatic): with two streams the batch is processed as two concurrent
    half-batches - images are independent, so results are identical - and the narrow, latency-bound launches of
    one half (SE gates, top-k sort, NMS: 64 workgroups on a 256-CU chip) overlap with the wide kernels of the
    other.  The halves run on shallow copies of the model (same parameters, own launch plans and buffers)."""

    def __init__(self, model, streams=None):
        super().__init__()
        self.model = model
        self.config = model.config
        self.num_levels = model.config.num_levels
        self.num_classes = model.config.num_classes
        self.anchors = Anchors.from_config(model.config)
        self.max_detection_points = model.config.max_detection_points
        self.max_det_per_image = model.config.max_det_per_image
        self.soft_nms = model.config.soft_nms
        self.streams = streams
        self.last_count = None
        self.last_ood = None
        self._replicas = None       # [(model copy, stream)], built on first use
        self._ood_buf = None

    def _one(self, model, x, img_scale, img_size):
        """One (sub-)batch: model -> top-k -> decode -> NMS -> OOD gather.  The top-k runs without the box gather and
        decode reads the box regressions from the box head's packed output through the top-k indices.  Class head, box
        head and top-k stay in launch order on one stream (see `forked_stream` for why nothing forks from here)."""
        lib = _lib.load()
        class_out, box_out = model(x)
        cls_topk, _, indices, classes = _post_process(
            class_out, None, num_levels=self.num_levels, num_classes=self.num_classes,
            max_detection_points=self.max_detection_points, anchor_max=model.ood_max_logit)
        B, k = indices.shape
        det, count, keep_src = batched_detections(
            cls_topk.reshape(B, k), None, self.anchors.boxes, indices, classes, img_scale, img_size,
            max_det_per_image=self.max_det_per_image, soft_nms=self.soft_nms, box_all=_packed(box_out, self.num_levels, 4))
        energy = torch.empty(B, self.max_det_per_image, dtype=torch.float32, device=x.device)
        maxlogit = torch.empty_like(energy)
        anchor = torch.empty(B, self.max_det_per_image, dtype=torch.int64, device=x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(lib.effdet_gather_ood(st, keep_src.data_ptr(), indices.data_ptr(), model.ood_energy.data_ptr(),
                                         model.ood_max_logit.data_ptr(), model.ood_energy.shape[1], B, k,
                                         self.max_det_per_image, energy.data_ptr(), maxlogit.data_ptr(), anchor.data_ptr()),
                   'effdet_gather_ood')
        return det, count, energy, maxlogit, anchor

    def _split_setup(self, x, n):
        """Shallow model copies (shared parameters) with engines that write their OOD rows into one [B, N] buffer."""
        import copy
        B, size = x.shape[0], (x.shape[2], x.shape[3])
        N = self.anchors.boxes.shape[0]
        key = (B, size, n, x.device, self.model.weights_token(), self.num_classes)
        if self._replicas is not None and self._replicas[0] == key and all(rep._engine is eng for rep, _, eng in self._replicas[1]):
            return self._replicas[1]
        e = torch.empty(B, N, dtype=torch.float32, device=x.device)
        m = torch.empty(B, N, dtype=torch.float32, device=x.device)
        self._ood_buf = (e, m)
        Bh = B // n
        reps = []
        for i in range(n):
            rep = self.model if i == 0 else copy.copy(self.model)
            eng = rep.prepare(Bh, size, ood_out=(e[i * Bh:(i + 1) * Bh], m[i * Bh:(i + 1) * Bh]))
            reps.append((rep, torch.cuda.Stream(x.device), eng))
        self._replicas = (key, reps)
        return reps

    def forward(self, x, img_info: Optional[Dict[str, torch.Tensor]] = None):
        # detections are not differentiable (top-k, NMS): the network always runs on the fused inference engine here, also
        # when the wrapped model was left in training mode or the caller did not enter torch.no_grad()
        with torch.no_grad():
            return self._forward(x, img_info)

    def _forward(self, x, img_info):
        if tuple(x.shape[2:]) != tuple(self.anchors.image_size):
            # the anchors follow the actual input size (the reference needs config.image_size == input size)
            self.anchors = Anchors(self.config.min_level, self.config.max_level, self.config.num_scales,
                                   self.config.aspect_ratios, self.config.anchor_scale, tuple(x.shape[2:])).to(x.device)
        if img_info is None:
            img_scale, img_size = None, None
        else:
            img_scale, img_size = img_info['img_scale'], img_info['img_size']
        B = x.shape[0]
        n = self.streams if self.streams is not None else (2 if B >= 16 and B % 2 == 0 else 1)
        if n <= 1 or B % n or x.device.type != 'cuda':
            det, count, energy, maxlogit, anchor = self._one(self.model, x, img_scale, img_size)
            self.last_count = count
            self.last_ood = {'energy': energy, 'max_logit': maxlogit, 'anchor_energy': self.model.ood_energy,
                             'anchor_max_logit': self.model.ood_max_logit, 'anchor_index': anchor}
            return det
        if getattr(_fork, 'depth', 0) > 0 and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('DetBenchPredict(streams=%d) would fork sub-batch streams from a stream that is itself a fork inside '
                               'a hipGraph capture; nested forks crash hipStreamEndCapture on ROCm 7.2. Capture from the origin '
                               'stream, or construct DetBenchPredict(model, streams=1) for use on side streams.' % n)
        reps = self._split_setup(x, n)
        Bh = B // n
        cur = torch.cuda.current_stream(x.device)
        outs = []
        for i, (rep, stream, _) in enumerate(reps):
            sl = slice(i * Bh, (i + 1) * Bh)
            stream.wait_stream(cur)
            with forked_stream(stream):
                outs.append(self._one(rep, x[sl], None if img_scale is None else img_scale[sl],
                                      None if img_size is None else img_size[sl]))
        for _, stream, _ in reps:
            cur.wait_stream(stream)
        det = torch.cat([o[0] for o in outs], 0)
        self.last_count = torch.cat([o[1] for o in outs], 0)
        self.last_ood = {'energy': torch.cat([o[2] for o in outs], 0), 'max_logit': torch.cat([o[3] for o in outs], 0),
                         'anchor_energy': self._ood_buf[0], 'anchor_max_logit': self._ood_buf[1],
                         'anchor_index': torch.cat([o[4] for o in outs], 0)}
        return det

    def ragged(self, det):
        """Per-image [n_i, 6] views, the shape the reference's generate_detections returns."""
        counts = self.last_count.tolist()
        return [det[i, :n] for i, n in enumerate(counts)]


class DetBenchTrain(nn.Module):
    """Training bench (effdet/bench.py:106-145): forward + anchor labelling + detection loss, all on HIP.
    In training mode (`bench.train()`, backbone BN put in eval mode as pretrain.py:168-176 does) `self.model(x)` is the
    differentiable float32 forward of train_engine.py and `output['loss'].backward()` fills every parameter's `.grad`."""

    def __init__(self, model, create_labeler=True):
        super().__init__()
        from .anchors import AnchorLabeler
        from .loss import DetectionLoss
        self.model = model
        self.config = model.config
        self.num_levels = model.config.num_levels
        self.num_classes = model.config.num_classes
        self.anchors = Anchors.from_config(model.config)
        self.max_detection_points = model.config.max_detection_points
        self.max_det_per_image = model.config.max_det_per_image
        self.soft_nms = model.config.soft_nms
        self.anchor_labeler = None
        if create_labeler:
            self.anchor_labeler = AnchorLabeler(self.anchors, self.num_classes, match_threshold=0.5)
        self.loss_fn = DetectionLoss(model.config)

    def forward(self, x, target: Dict[str, torch.Tensor]):
        class_out, box_out = self.model(x)
        if self.anchor_labeler is None:
            assert 'label_num_positives' in target
            cls_targets = [target['label_cls_%d' % l] for l in range(self.num_levels)]
            box_targets = [target['label_bbox_%d' % l] for l in range(self.num_levels)]
            num_positives = target['label_num_positives']
        else:
            cls_targets, box_targets, num_positives = self.anchor_labeler.batch_label_anchors(
                target['bbox'], target['cls'])
        loss, class_loss, box_loss = self.loss_fn(class_out, box_out, cls_targets, box_targets, num_positives)
        output = {'loss': loss, 'class_loss': class_loss, 'box_loss': box_loss}
        if not self.training:
            cls_topk, box_topk, indices, classes = _post_process(
                class_out, box_out, num_levels=self.num_levels, num_classes=self.num_classes,
                max_detection_points=self.max_detection_points)
            B, k = indices.shape
            det, count, _ = batched_detections(
                cls_topk.reshape(B, k), box_topk, self.anchors.boxes, indices, classes,
                target.get('img_scale'), target.get('img_size'),
                max_det_per_image=self.max_det_per_image, soft_nms=self.soft_nms)
            output['detections'] = det
            output['detection_counts'] = count
        return output


def unwrap_bench(model):
    if hasattr(model, 'module'):
        return unwrap_bench(model.module)
    elif hasattr(model, 'model'):
        return unwrap_bench(model.model)
    return model
