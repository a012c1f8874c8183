"""ORACLE (test infrastructure only - never imported by the product path).

`detection_loss` (below) restates the fork's loss_fn (effdet/loss.py:224-298) in differentiable torch ops - pinned
against the fixture the reference's own loss_fn produced (tests/golden/loss.npz) - so that, with oracle/model.py in
training-BN mode, torch autograd on the CPU is the checker for the HIP backward pass.

Restatement of the optimizer half of the reference's pretrain step (pretrain.py:272-276):
`torch.nn.utils.clip_grad_norm_(model.parameters(), 10.)` then `torch.optim.Adam(...).step()` (pretrain.py:179-185,
lr FLAGS.meta_lr = 1e-3, default betas / eps).  Pinned in tests/test_oracle_golden.py against torch's own
`clip_grad_norm_` + `torch.optim.Adam` on the CPU."""
import numpy as np
import torch


def clip_adam_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=10.0):
    """float32 numpy arrays (flat); returns (p', m', v', total_norm)."""
    p, g, m, v = [np.asarray(a, dtype=np.float32) for a in (p, g, m, v)]
    f = np.float32
    norm = None
    if max_norm is not None:
        norm = f(np.sqrt(np.sum(g.astype(np.float64) ** 2)))            # torch: norm of per-tensor norms
        coef = min(f(1.0), f(max_norm) / (norm + f(1e-6)))
        g = g * f(coef)
    m = m + f(1.0 - beta1) * (g - m)                                     # exp_avg.lerp_(grad, 1 - beta1)
    v = v * f(beta2) + f(1.0 - beta2) * g * g                            # exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = np.sqrt(1.0 - beta2 ** step)
    denom = np.sqrt(v) / f(bc2_sqrt) + f(eps)
    p = p - f(lr / bc1) * (m / denom)
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32), norm


def _one_hot(x, num_classes):
    """effdet/loss.py:182-186: negative entries give an all-zero row."""
    nn_ = (x >= 0).unsqueeze(-1)
    oh = torch.zeros(x.shape + (num_classes,), dtype=torch.float32)
    return oh.scatter(-1, x.unsqueeze(-1) * nn_, 1) * nn_


def detection_loss(cls_outputs, box_outputs, cls_targets, box_targets, num_positives, num_classes, alpha, delta,
                   box_loss_weight, label_smoothing=0.0):
    """loss_fn of the fork (effdet/loss.py:224-298) with new_focal_loss (:49-95; gamma unused - the modulating factor is
    commented out), huber_loss (:104-118) and _box_loss (:171-179).  cls_outputs[l] [B, A*C, H, W], box_outputs[l]
    [B, A*4, H, W], cls_targets[l] [B, H, W, A] int64, box_targets[l] [B, H, W, A*4].  Differentiable (torch autograd)."""
    import torch.nn.functional as F
    norm = num_positives.sum() + 1.0
    cls_losses, box_losses = [], []
    for l in range(len(cls_outputs)):
        t = cls_targets[l]
        oh = _one_hot(t, num_classes)
        bs, h, w, _, _ = oh.shape
        oh = oh.view(bs, h, w, -1)
        logits = cls_outputs[l].permute(0, 2, 3, 1)
        alpha_factor = oh * alpha + (1.0 - oh) * (1.0 - alpha)
        tt = oh * (1.0 - label_smoothing) + 0.5 * label_smoothing if label_smoothing > 0.0 else oh
        loss = (1 / norm) * alpha_factor * F.binary_cross_entropy_with_logits(logits, tt, reduction='none')
        loss = loss.view(bs, h, w, -1, num_classes) * (t != -2).unsqueeze(-1)
        cls_losses.append(loss.sum())
        bo = box_outputs[l].permute(0, 2, 3, 1)
        bt = box_targets[l]
        err = bo - bt
        abs_err = err.abs()
        quad = torch.clamp(abs_err, max=delta)
        lin = abs_err - quad
        hub = (0.5 * quad.pow(2) + delta * lin) * (bt != 0.0)
        box_losses.append(hub.sum() / (norm * 4.0))
    cls_loss = torch.stack(cls_losses, -1).sum(-1)
    box_loss = torch.stack(box_losses, -1).sum(-1)
    return cls_loss + box_loss_weight * box_loss, cls_loss, box_loss
