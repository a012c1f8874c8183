"""Differentiable primitives on the training kernels of libeffdet_hip.so, closed under differentiation.

The MAML inner loop of the reference (infer.py:658) takes `torch.autograd.grad(supp_class_loss, class_net.parameters(),
create_graph=True)` and back-propagates the query loss THROUGH that gradient (second-order terms).  A single autograd node
with a hand-written backward (effdet/meta_grad.py) cannot provide that: its backward is not itself differentiable.  Here
every activation-sized operation of `MetaHead.forward` (effdet/efficientdet.py:636-695) is an autograd Function whose
backward is again made of Functions of this file, so autograd can differentiate the head to any order that the element-wise
kernels cover (SiLU: up to its second derivative, i.e. exactly the double backward MAML needs):

    Linear      y = A W^T (+ b)           backward: Linear(g, W^T), MmTN(g, A), ColSum(g)
    MmTN        P = Y^T X  (param sized)  backward: Linear(X, G), Linear(Y, G^T)
    DwConv      depthwise 3x3 / s1 SAME   backward: DwBwdDx(g, taps), DwBwdDw(g, x)
    DwBwdDx     conv-transpose by taps    backward: DwConv(gg, taps), DwBwdDw(G, gg)
    DwBwdDw     tap correlation [9, C]    backward: DwConv(X, h), DwBwdDx(G, h)
    ColSum / RowBcast / MulCh / ColDot / Mul / Add / Silu / SiluBwd / SiluBwd2

All tensors are contiguous float32 on one GPU; activations are [M, C] or [B, H, W, C] (NHWC).  Parameter-sized algebra
(transposes, the [C]-vectors of batch-norm statistics) stays in PyTorch, as in the pretrain step.  No CPU fallback.
"""
import torch

from .. import _lib
from ..train_engine import _Ops

_ops = {}


def ops_for(dev):
    key = (dev.type, dev.index)
    if key not in _ops:
        if dev.type != 'cuda':
            raise RuntimeError('the differentiable MetaHead runs on the GPU only (no CPU fallback)')
        _ops[key] = _Ops(dev)
    return _ops[key]


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, W, b):
        A, W = _c(A), _c(W)
        ctx.save_for_backward(A, W)
        ctx.has_b = b is not None
        return ops_for(A.device).gemm_nt(A.reshape(-1, W.shape[1]), W, None if b is None else _c(b))

    @staticmethod
    def backward(ctx, g):
        A, W = ctx.saved_tensors
        gA = Linear.apply(g, W.t().contiguous(), None).reshape(A.shape) if ctx.needs_input_grad[0] else None
        gW = MmTN.apply(g, A.reshape(-1, W.shape[1])) if ctx.needs_input_grad[1] else None
        gb = ColSum.apply(g) if ctx.has_b and ctx.needs_input_grad[2] else None
        return gA, gW, gb


class MmTN(torch.autograd.Function):
    """Y [M, N], X [M, K] -> Y^T X [N, K]"""
    @staticmethod
    def forward(ctx, Y, X):
        Y, X = _c(Y), _c(X)
        ctx.save_for_backward(Y, X)
        return ops_for(Y.device).gemm_tn(Y, X, Y.shape[1], X.shape[1])[0].clone()

    @staticmethod
    def backward(ctx, G):
        Y, X = ctx.saved_tensors
        gY = Linear.apply(X, G, None) if ctx.needs_input_grad[0] else None                       # X G^T   [M, N]
        gX = Linear.apply(Y, G.t().contiguous(), None) if ctx.needs_input_grad[1] else None      # Y G     [M, K]
        return gY, gX


class ColSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        X = _c(X)
        ctx.shape = X.shape
        return ops_for(X.device).col_reduce(0, X.reshape(-1, X.shape[-1])).clone()

    @staticmethod
    def backward(ctx, g):
        return RowBcast.apply(g, ctx.shape)


class RowBcast(torch.autograd.Function):
    """v [C] -> every row of a tensor of `shape` (..., C)"""
    @staticmethod
    def forward(ctx, v, shape):
        v = _c(v)
        if v.numel() % 4:                        # the 9-channel logit rows (one per anchor): not a shape of the element-wise kernels
            return v.expand(shape).contiguous()
        z = torch.zeros(shape, dtype=torch.float32, device=v.device)
        return ops_for(v.device).ew(3, z, v=(torch.ones_like(v), v, None, None))

    @staticmethod
    def backward(ctx, g):
        return ColSum.apply(g), None


class MulCh(torch.autograd.Function):
    """x * v[c]"""
    @staticmethod
    def forward(ctx, X, v):
        X, v = _c(X), _c(v)
        ctx.save_for_backward(X, v)
        return ops_for(X.device).ew(3, X, v=(v, None, None, None))

    @staticmethod
    def backward(ctx, g):
        X, v = ctx.saved_tensors
        return (MulCh.apply(g, v) if ctx.needs_input_grad[0] else None,
                ColDot.apply(g, X) if ctx.needs_input_grad[1] else None)


class ColDot(torch.autograd.Function):
    """sum over rows of a * b -> [C]"""
    @staticmethod
    def forward(ctx, A, B):
        A, B = _c(A), _c(B)
        ctx.save_for_backward(A, B)
        C = A.shape[-1]
        return ops_for(A.device).col_reduce(1, A.reshape(-1, C), B.reshape(-1, C)).clone()

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        return (MulCh.apply(B, g) if ctx.needs_input_grad[0] else None,
                MulCh.apply(A, g) if ctx.needs_input_grad[1] else None)


class Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B):
        A, B = _c(A), _c(B)
        ctx.save_for_backward(A, B)
        return ops_for(A.device).ew(10, A, B)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        return (Mul.apply(g, B) if ctx.needs_input_grad[0] else None, Mul.apply(g, A) if ctx.needs_input_grad[1] else None)


class Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B):
        return ops_for(A.device).add(_c(A), _c(B))

    @staticmethod
    def backward(ctx, g):
        return g, g


class Silu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Z):
        Z = _c(Z)
        ctx.save_for_backward(Z)
        return ops_for(Z.device).silu(Z)

    @staticmethod
    def backward(ctx, g):
        (Z,) = ctx.saved_tensors
        return SiluBwd.apply(Z, g)


class SiluBwd(torch.autograd.Function):
    """g * silu'(z)"""
    @staticmethod
    def forward(ctx, Z, G):
        Z, G = _c(Z), _c(G)
        ctx.save_for_backward(Z, G)
        return ops_for(Z.device).silu_bwd(Z, G)

    @staticmethod
    def backward(ctx, gg):
        Z, G = ctx.saved_tensors
        return (SiluBwd2.apply(Z, G, gg) if ctx.needs_input_grad[0] else None,
                SiluBwd.apply(Z, gg) if ctx.needs_input_grad[1] else None)


class SiluBwd2(torch.autograd.Function):
    """a * b * silu''(z): the end of what the kernels differentiate (third-order terms are not built)"""
    @staticmethod
    def forward(ctx, Z, A, B):
        Z, A, B = _c(Z), _c(A), _c(B)
        ctx.save_for_backward(Z, A, B)
        return ops_for(Z.device).ew(11, A, B, Z)

    @staticmethod
    def backward(ctx, g):
        Z, A, B = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError('third-order derivative of SiLU: the MetaHead is differentiable to second order (MAML)')
        return None, (SiluBwd2.apply(Z, g, B) if ctx.needs_input_grad[1] else None), (SiluBwd2.apply(Z, A, g) if ctx.needs_input_grad[2] else None)


def _dw_dims(X):
    if X.dim() != 4 or X.shape[-1] % 4:
        raise RuntimeError('depthwise primitives take [B, H, W, C] float32 tensors with C % 4 == 0')
    return X.shape


class DwConv(torch.autograd.Function):
    """depthwise 3x3, stride 1, TF-SAME (= symmetric pad 1); taps [9, C], t = ky * 3 + kx"""
    @staticmethod
    def forward(ctx, X, taps):
        X, taps = _c(X), _c(taps)
        B, H, W, C = _dw_dims(X)
        ctx.save_for_backward(X, taps)
        one, zero = torch.ones(C, dtype=torch.float32, device=X.device), torch.zeros(C, dtype=torch.float32, device=X.device)
        return ops_for(X.device).dw_fwd(X, taps, one, zero, 3, 1)

    @staticmethod
    def backward(ctx, g):
        X, taps = ctx.saved_tensors
        return (DwBwdDx.apply(g, taps) if ctx.needs_input_grad[0] else None,
                DwBwdDw.apply(g, X) if ctx.needs_input_grad[1] else None)


class DwBwdDx(torch.autograd.Function):
    """dx[p] = sum_t taps[t] G[p - off(t)]  (gradient of DwConv w.r.t. its input)"""
    @staticmethod
    def forward(ctx, G, taps):
        G, taps = _c(G), _c(taps)
        B, H, W, C = _dw_dims(G)
        ctx.save_for_backward(G, taps)
        o = ops_for(G.device)
        dx = o.new(B, H, W, C)
        _lib.check(o.lib.effdet_train_dwconv_bwd_dx(o.st(), G.data_ptr(), taps.data_ptr(), dx.data_ptr(), B, H, W, C, 3, 1),
                   'effdet_train_dwconv_bwd_dx')
        return dx

    @staticmethod
    def backward(ctx, gg):
        G, taps = ctx.saved_tensors
        return (DwConv.apply(gg, taps) if ctx.needs_input_grad[0] else None,
                DwBwdDw.apply(G, gg) if ctx.needs_input_grad[1] else None)


class DwBwdDw(torch.autograd.Function):
    """out[t] = sum_q G[q] X[q + off(t)]  (gradient of DwConv w.r.t. its taps), [9, C]"""
    @staticmethod
    def forward(ctx, G, X):
        G, X = _c(G), _c(X)
        B, H, W, C = _dw_dims(X)
        ctx.save_for_backward(G, X)
        o = ops_for(G.device)
        n = o.lib.effdet_train_dwconv_bwd_dw_workspace_floats(B, H, W, C, 3, 1)
        ws = o.ws(n)
        out = o.new(10, C)
        _lib.check(o.lib.effdet_train_dwconv_bwd_dw(o.st(), G.data_ptr(), X.data_ptr(), out.data_ptr(), B, H, W, C, 3, 1,
                                                    ws.data_ptr(), ws.numel(), 0), 'effdet_train_dwconv_bwd_dw')
        return out[:9].clone()

    @staticmethod
    def backward(ctx, h):
        G, X = ctx.saved_tensors
        h = _c(h)
        return (DwConv.apply(X, h) if ctx.needs_input_grad[0] else None,
                DwBwdDx.apply(G, h) if ctx.needs_input_grad[1] else None)


def batch_norm_train(x2d, weight, bias, running_mean, running_var, momentum=0.1, eps=1e-5):
    """F.batch_norm(x, running_mean, running_var, weight, bias, training=True) over the rows of x2d [M, C] (efficientdet.py:673),
    written with the primitives above so that it differentiates twice; the running buffers are updated in place like
    torch's (biased variance for the output, unbiased for the buffer)."""
    M = x2d.shape[0]
    mean = ColSum.apply(x2d) / M
    xc = Add.apply(x2d, RowBcast.apply(-mean, x2d.shape))
    var = ColDot.apply(xc, xc) / M
    rstd = torch.rsqrt(var + eps)
    with torch.no_grad():
        if running_mean is not None:
            running_mean.mul_(1.0 - momentum).add_(momentum * mean.detach())
            running_var.mul_(1.0 - momentum).add_(momentum * var.detach() * (M / max(M - 1, 1)))
    y = MulCh.apply(xc, weight * rstd)
    return Add.apply(y, RowBcast.apply(bias, x2d.shape))


def meta_head_forward(x_levels, conv_dw_rep, conv_pw_rep, conv_pb_rep, bn_rep_w, bn_rep_b, predict, predict_class, levels, R,
                      running_mu, running_std):
    """MetaHead.forward (effdet/efficientdet.py:664-683) on the primitives: x_levels = NHWC inputs of `levels`;
    returns (outputs, x_pred activations, class_outputs or None), all NHWC."""
    outs, preds, couts = [], [], ([] if predict_class is not None else None)
    taps = lambda w: w.reshape(w.shape[0], 9).t().contiguous()              # [F, 1, 3, 3] -> [9][F], differentiable
    for x, level in zip(x_levels, levels):
        B, H, W, F = x.shape
        t = x
        for r in range(R):
            d = DwConv.apply(t, taps(conv_dw_rep[r]))
            c = Linear.apply(d.reshape(-1, F), conv_pw_rep[r].reshape(conv_pw_rep[r].shape[0], F), conv_pb_rep[r])
            y = batch_norm_train(c, bn_rep_w[level * R + r], bn_rep_b[level * R + r], running_mu, running_std)
            t = Silu.apply(y).reshape(B, H, W, -1)
        xp = DwConv.apply(t, taps(predict[0]))
        A = predict[1].shape[0]
        outs.append(Linear.apply(xp.reshape(-1, F), predict[1].reshape(A, F), predict[2]).reshape(B, H, W, A))
        preds.append(xp)
        if predict_class is not None:
            Ac = predict_class[0].shape[0]
            couts.append(Linear.apply(xp.reshape(-1, F), predict_class[0].reshape(Ac, F), predict_class[1]).reshape(B, H, W, Ac))
    return outs, preds, couts
