"""Soft-NMS entry points of the reference (effdet/soft_nms.py) on the HIP kernel."""
import torch

from .. import _lib


def _run(boxes, scores, idxs, method_gaussian, sigma, iou_threshold, score_threshold):
    lib = _lib.load()
    if boxes.device.type != 'cuda':
        raise RuntimeError('soft_nms runs on the GPU only (no CPU fallback)')
    n = scores.numel()
    dev = boxes.device
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.float32, device=dev)
    b = boxes.float().contiguous().reshape(1, n, 4)
    s = scores.float().contiguous().reshape(1, n)
    c = idxs.to(torch.int32).contiguous().reshape(1, n)
    src = torch.arange(n, dtype=torch.int32, device=dev).reshape(1, n)
    count = torch.tensor([n], dtype=torch.int32, device=dev)
    # class offset of batched_soft_nms: max coordinate + 1 (soft_nms.py:163-165); plain soft_nms: classes are 0
    maxc = b.max().reshape(1)
    max_det = n                                  # the reference returns every pick (soft_nms.py:88-112), not a top-100
    det = torch.empty(1, max_det, 6, dtype=torch.float32, device=dev)
    dc = torch.empty(1, dtype=torch.int32, device=dev)
    keep = torch.empty(1, max_det, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if n <= 8192:                                # candidates in registers
        _lib.check(lib.effdet_nms_soft(st, b.data_ptr(), s.data_ptr(), c.data_ptr(), src.data_ptr(), count.data_ptr(),
                                       maxc.data_ptr(), 1, n, 1 if method_gaussian else 0, sigma, iou_threshold,
                                       score_threshold, max_det, None, det.data_ptr(), dc.data_ptr(), keep.data_ptr()),
                   'effdet_nms_soft')
    else:                                        # any size: working scores in a scratch row
        scratch = torch.empty(1, n, dtype=torch.float32, device=dev)
        _lib.check(lib.effdet_nms_soft_large(st, b.data_ptr(), s.data_ptr(), c.data_ptr(), src.data_ptr(), count.data_ptr(),
                                             maxc.data_ptr(), 1, n, 1 if method_gaussian else 0, sigma, iou_threshold,
                                             score_threshold, max_det, None, det.data_ptr(), dc.data_ptr(), keep.data_ptr(),
                                             scratch.data_ptr()), 'effdet_nms_soft_large')
    m = int(dc.item())
    return keep[0, :m].long(), det[0, :m, 4].clone()


def soft_nms(boxes, scores, method_gaussian: bool = True, sigma: float = 0.5, iou_threshold: float = .5,
             score_threshold: float = 0.005):
    """soft_nms.py:42-112: (kept original indices in pick order, rescored scores), every pick, any n."""
    return _run(boxes, scores, torch.zeros_like(scores, dtype=torch.int32), method_gaussian, sigma, iou_threshold, score_threshold)


def batched_soft_nms(boxes, scores, idxs, method_gaussian: bool = True, sigma: float = 0.5,
                     iou_threshold: float = .5, score_threshold: float = 0.001):
    """soft_nms.py:115-169 (per-class via the coordinate-offset trick)."""
    return _run(boxes, scores, idxs, method_gaussian, sigma, iou_threshold, score_threshold)
