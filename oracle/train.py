"""ORACLE (test infrastructure only - never imported by the product path).

Restatement of the optimizer half of the reference's pretrain step (pretrain.py:272-276):
`torch.nn.utils.clip_grad_norm_(model.parameters(), 10.)` then `torch.optim.Adam(...).step()` (pretrain.py:179-185,
lr FLAGS.meta_lr = 1e-3, default betas / eps).  Pinned in tests/test_oracle_golden.py against torch's own
`clip_grad_norm_` + `torch.optim.Adam` on the CPU."""
import numpy as np


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
