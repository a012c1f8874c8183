"""Detection loss on the HIP kernel (reference: effdet/loss.py).

`loss_fn` / `DetectionLoss` keep the reference signatures (loss.py:224-298, :355-401).  The fork's
`new_focal_loss` ignores gamma (its modulating factor is commented out, loss.py:77-79), so the class loss is
alpha-weighted BCE-with-logits with optional label smoothing; `legacy_focal` / `jit_loss` are not built.
The op is differentiable w.r.t. the class / box head outputs (the kernel emits both the loss and d total / d
outputs); there is no CPU fallback.
"""
from typing import List, Tuple

import torch
import torch.nn as nn

from .. import _lib


class _DetectionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_all, box_all, cls_t, box_t, num_positives, alpha, delta, box_loss_weight, label_smoothing):
        lib = _lib.load()
        if cls_all.device.type != 'cuda':
            raise RuntimeError('DetectionLoss runs on the GPU only (no CPU fallback)')
        ctx.in_dtypes = (cls_all.dtype, box_all.dtype)
        if cls_all.dtype == torch.bfloat16 and box_all.dtype == torch.float32:
            cls_all = cls_all.float()                        # a bf16 inference model writes float32 box regressions (engine.py):
                                                             # the loss runs on the float32 kernel, the boxes keep their precision
        if cls_all.dtype != box_all.dtype or cls_all.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError('head outputs must both be float32 or bfloat16')
        B, N, C = cls_all.shape
        dev = cls_all.device
        cls_c, box_c = cls_all.contiguous(), box_all.contiguous()
        cls_t = cls_t.to(device=dev, dtype=torch.int64).contiguous()
        box_t = box_t.to(device=dev, dtype=torch.float32).contiguous()
        npos = num_positives.to(device=dev, dtype=torch.float32).contiguous()
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        g_cls, g_box = torch.empty_like(cls_c), torch.empty_like(box_c)
        nws = lib.effdet_detection_loss_workspace_floats(B, N, C)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.effdet_detection_loss(st, 0 if cls_c.dtype == torch.float32 else 1, cls_c.data_ptr(), box_c.data_ptr(),
                                             cls_t.data_ptr(), box_t.data_ptr(), npos.data_ptr(), B, N, C, alpha, delta,
                                             box_loss_weight, label_smoothing, out3.data_ptr(), g_cls.data_ptr(),
                                             g_box.data_ptr(), ws.data_ptr(), nws), 'effdet_detection_loss')
        ctx.save_for_backward(g_cls, g_box)
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        g_cls, g_box = ctx.saved_tensors
        return ((g_cls * g_total.to(g_cls.dtype)).to(ctx.in_dtypes[0]), (g_box * g_total.to(g_box.dtype)).to(ctx.in_dtypes[1]),
                None, None, None, None, None, None, None)


def _pack(outs, width):
    """per-level [B, A*width, H, W] -> [B, N, width] (a view when the tensors are the engine's own)"""
    B = outs[0].shape[0]
    base = outs[0]._base
    if base is not None and base.dim() == 3 and base.shape[0] == B and base.shape[2] == width and base.is_contiguous() \
            and all(o._base is base for o in outs):
        n = sum(o.shape[1] * o.shape[2] * o.shape[3] for o in outs) // width
        if base.shape[1] == n:
            return base
    return torch.cat([o.permute(0, 2, 3, 1).reshape(B, -1, width) for o in outs], 1)


def _pack_targets(ts, last):
    """per-level targets [B, H, W, A(*4)] -> [B, N] (last = 0) or [B, N, 4]: the labeler's own packed tensor when the levels are
    its views in order (AnchorLabeler._unpack), else a concatenation"""
    B = ts[0].shape[0]
    base = ts[0]._base
    if base is not None and base.is_contiguous() and base.shape[0] == B and all(t._base is base for t in ts) and \
            base.dim() == (3 if last else 2) and (not last or base.shape[2] == last):
        n, off, ok = base.shape[1], 0, True
        for t in ts:
            cnt = t[0].numel() // (last or 1)
            ok = ok and t.storage_offset() == off * (last or 1) and t.stride(0) == base.stride(0)
            off += cnt
        if ok and off == n:
            return base
    if last:
        return torch.cat([t.reshape(B, -1, last) for t in ts], 1)
    return torch.cat([t.reshape(B, -1) for t in ts], 1)


def loss_fn(cls_outputs: List[torch.Tensor], box_outputs: List[torch.Tensor], cls_targets: List[torch.Tensor],
            box_targets: List[torch.Tensor], num_positives: torch.Tensor, num_classes: int, alpha: float, gamma: float,
            delta: float, box_loss_weight: float, label_smoothing: float = 0.,
            legacy_focal: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if legacy_focal:
        raise NotImplementedError('legacy_focal loss is not built (the fork runs new_focal_loss)')
    B = cls_outputs[0].shape[0]
    cls_all = _pack(list(cls_outputs), num_classes)
    box_all = _pack(list(box_outputs), 4)
    cls_t = _pack_targets(list(cls_targets), 0)
    box_t = _pack_targets(list(box_targets), 4)
    total, parts = _DetectionLossFn.apply(cls_all, box_all, cls_t, box_t, num_positives, float(alpha), float(delta),
                                          float(box_loss_weight), float(label_smoothing))
    return total, parts[1], parts[2]


class DetectionLoss(nn.Module):
    __constants__ = ['num_classes']

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_classes = config.num_classes
        self.alpha = config.alpha
        self.gamma = config.gamma
        self.delta = config.delta
        self.box_loss_weight = config.box_loss_weight
        self.label_smoothing = config.label_smoothing
        self.legacy_focal = config.legacy_focal
        self.use_jit = config.jit_loss

    def forward(self, cls_outputs, box_outputs, cls_targets, box_targets, num_positives):
        return loss_fn(cls_outputs, box_outputs, cls_targets, box_targets, num_positives,
                       num_classes=self.num_classes, alpha=self.alpha, gamma=self.gamma, delta=self.delta,
                       box_loss_weight=self.box_loss_weight, label_smoothing=self.label_smoothing,
                       legacy_focal=self.legacy_focal)


# ------------------------------------------------------------------------------------------------------------------------
# Episode-level auxiliary losses the fork's scripts import next to DetectionLoss (infer.py:17, pretrain.py:15).  They act on
# the small per-episode tensors of the few-shot logic (similarity targets, anchor confidences), not on the detection hot
# path of SURVEY §8: plain differentiable tensor expressions on whatever device their inputs live on.
# ------------------------------------------------------------------------------------------------------------------------
def cosine_loss(input, target, margin=0., reduction='mean'):
    """effdet/loss.py:97-101: pairs labelled 1 are pulled to similarity 1, the others pushed below `margin`; hinge at 0; mean."""
    per = torch.where(target == 1., 1 - input, input - margin)
    return per.clamp(min=0.).mean()


def _signed_weight_sums(err, weights):
    ws = torch.sign(err) * weights
    return ws[ws > 0.].sum(), ws[ws < 0.].sum()


def smooth_l1_loss(input, target, beta: float = 1. / 9, weights=None, size_average: bool = False):
    """effdet/loss.py:121-154: L1 below beta = 1e-5, else quadratic inside |err| < beta and linear outside; optional weights.
    Returns the mean (size_average) or (sum, sum of positive signed weights, sum of negative signed weights)."""
    err = input - target
    a = err.abs()
    loss = a if beta < 1e-5 else torch.where(a < beta, 0.5 * a.pow(2) / beta, a - 0.5 * beta)
    pos = neg = None
    if weights is not None:
        if beta < 1e-5:                        # same failure as the reference, whose pure-L1 branch never defines `err` (:132-147)
            raise UnboundLocalError("local variable 'err' referenced before assignment (smooth_l1_loss: beta < 1e-5 with weights)")
        loss = loss * weights
        pos, neg = _signed_weight_sums(err, weights)
    if size_average:
        return loss.mean()
    if weights is None:
        raise UnboundLocalError('smooth_l1_loss(size_average=False) returns the weighted sign sums: pass weights (as the reference requires)')
    return loss.sum(), pos, neg


def l2_loss(input, target, beta: float = 1. / 9, weights=None, size_average: bool = False):
    """effdet/loss.py:156-168: squared error (optionally weighted), always the mean, plus the two signed weight sums."""
    err = input - target
    loss = err ** 2
    if weights is None:
        raise UnboundLocalError('l2_loss returns the weighted sign sums: pass weights (as the reference requires)')
    loss = loss * weights
    pos, neg = _signed_weight_sums(err, weights)
    return loss.mean(), pos, neg


class SupportLoss(nn.Module):
    """effdet/loss.py:404-439 -> class_loss_fn (:188-221) -> new_focal_loss (:49-95): per level, alpha-weighted (alpha given at
    call time, may be None) `loss_func(logits, targets)` ('ce': BCE with logits, 'mse'), label smoothing, divided by
    sum(num_positives) + 1, summed over everything.  cls_targets are dense float maps [B, A*C, H, W] like the outputs."""

    def __init__(self, config, loss_type):
        super().__init__()
        self.config = config
        self.num_classes = config.num_classes
        self.alpha, self.gamma, self.label_smoothing = config.alpha, config.gamma, config.label_smoothing
        if loss_type not in ('ce', 'mse'):
            raise ValueError("loss_type must be 'ce' or 'mse'")
        self.loss_type = loss_type

    def forward(self, cls_outputs, cls_targets, num_positives, alpha):
        import torch.nn.functional as F
        norm = num_positives.sum() + 1.0
        total = []
        for out, tgt in zip(cls_outputs, cls_targets):
            logits, t = out.permute(0, 2, 3, 1), tgt.permute(0, 2, 3, 1).to(out.dtype)
            factor = None if alpha is None else t * alpha + (1. - t) * (1. - alpha)
            if self.label_smoothing > 0.:
                t = t * (1. - self.label_smoothing) + .5 * self.label_smoothing
            per = F.binary_cross_entropy_with_logits(logits, t, reduction='none') if self.loss_type == 'ce' else F.mse_loss(logits, t, reduction='none')
            per = per / norm if factor is None else factor * per / norm
            total.append(per.sum())
        return torch.stack(total).sum()
