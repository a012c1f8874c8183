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


def novelty_score(proj_embds: torch.Tensor, confs: torch.Tensor, proto_idx: torch.Tensor, dot_mult: float, dot_add: float,
                  sim_target: str = 'avg'):
    """The fork's own novelty score (infer.py:425-427, 465-471, 607-616), one HIP launch: for every anchor
    `sigmoid(dot_mult * (conf + dot_add)) * sim`, sim = mean ('avg') or max ('max') cosine similarity of its ProjectionNet
    embedding to the cluster prototypes `proto_idx` (the episode code's `max_idxs`).
    proj_embds [n, d] float32 (un-normalised ProjectionNet outputs), confs [n] anchor confidence logits.
    Returns dict(score, soft_thresh, sim), each [n] float32."""
    if proj_embds.device.type != 'cuda' or proj_embds.dtype != torch.float32 or proj_embds.dim() != 2:
        raise RuntimeError('expected float32 [n, d] GPU embeddings (no CPU fallback)')
    if sim_target not in ('avg', 'max'):
        raise ValueError("sim_target must be 'avg' or 'max' (infer.py FLAGS.sim_target)")
    lib = _lib.load()
    e = proj_embds.detach().contiguous()
    n, d = e.shape
    c = confs.detach().to(device=e.device, dtype=torch.float32).reshape(n).contiguous()
    pi = proto_idx.to(device=e.device, dtype=torch.int64).reshape(-1).contiguous()
    m = pi.numel()
    if m == 0 or m * d > 16384:
        raise ValueError('between 1 and 16384 / d prototypes')
    out = torch.empty(3, n, dtype=torch.float32, device=e.device)
    st = torch.cuda.current_stream(e.device).cuda_stream
    _lib.check(lib.effdet_novelty_score(st, e.data_ptr(), c.data_ptr(), pi.data_ptr(), n, d, m, float(dot_mult), float(dot_add),
                                        1 if sim_target == 'max' else 0, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()),
               'effdet_novelty_score')
    return {'score': out[0], 'soft_thresh': out[1], 'sim': out[2]}
