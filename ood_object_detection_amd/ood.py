"""OOD evaluation helpers on the device.

The class head emits per-anchor `energy = -logsumexp_c z` and `max_logit = max_c z`.  For the
in-distribution-vs-OOD experiment an image is scored by `max_a(-energy_a)` and the separation is reported as
AUROC; both run as HIP kernels so that nothing but two integers leaves the GPU."""
import torch

from . import _lib


def image_scores(anchor_energy: torch.Tensor) -> torch.Tensor:
    """[B, N] float32 per-anchor energies (model.ood_energy) -> [B] image-level scores max_a(-energy)."""
    if anchor_energy.device.type != 'cuda' or anchor_energy.dtype != torch.float32 or anchor_energy.dim() != 2:
        raise RuntimeError('expected a float32 [B, N] GPU tensor (no CPU fallback)')
    lib = _lib.load()
    e = anchor_energy.contiguous()
    out = torch.empty(e.shape[0], dtype=torch.float32, device=e.device)
    st = torch.cuda.current_stream(e.device).cuda_stream
    _lib.check(lib.effdet_ood_image_score(st, e.data_ptr(), e.shape[0], e.shape[1], out.data_ptr()), 'effdet_ood_image_score')
    return out


def auroc(in_dist_scores: torch.Tensor, ood_scores: torch.Tensor) -> float:
    """AUROC with the in-distribution images as the positive class (exact pair counting, ties count 1/2)."""
    for t in (in_dist_scores, ood_scores):
        if t.device.type != 'cuda' or t.dtype != torch.float32 or t.dim() != 1 or t.numel() == 0:
            raise RuntimeError('expected non-empty float32 1-d GPU tensors (no CPU fallback)')
    lib = _lib.load()
    pos, neg = in_dist_scores.contiguous(), ood_scores.contiguous()
    counts = torch.empty(2, dtype=torch.int64, device=pos.device)
    st = torch.cuda.current_stream(pos.device).cuda_stream
    _lib.check(lib.effdet_auroc_counts(st, pos.data_ptr(), neg.data_ptr(), pos.numel(), neg.numel(), counts.data_ptr()),
               'effdet_auroc_counts')
    gt, eq = counts.tolist()
    return (gt + 0.5 * eq) / (pos.numel() * neg.numel())
