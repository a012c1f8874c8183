"""Differentiable `MetaHead.forward` (reference: effdet/efficientdet.py:636-695) on the training kernels of libeffdet_hip.so:
the MAML inner / outer loop of infer.py:561-681 takes gradients of a loss on the MetaHead's outputs with respect to the head's
parameters (`torch.autograd.grad(..., model.class_net.parameters(), create_graph=True)`, :658) and - through the fast weights -
with respect to the BiFPN activations.

`MetaHeadFn` is one autograd node: forward keeps the activations, backward runs depthwise / pointwise / batch-statistics-BN /
SiLU backward kernels (csrc/train_net.hip, the operators of the pretrain step) and returns d loss / d (every weight in the
reference's fast-weight order) and d loss / d (every input level).  float32 only, like the rest of the training path.
FIRST ORDER: this node's backward is not itself differentiable, so with `create_graph=True` it would drop the second-order
terms of MAML.  `MetaHead.first_order = True` selects it (one node, fewer launches); the default differentiable path is the
composition of primitives in effdet/meta_ops.py, which autograd differentiates twice like the reference.
"""
import warnings

import torch

from ..train_engine import TrainEngine, _Ops


class _Conv(object):                       # what TrainEngine's _pw_* / _dw_* read of a conv module
    def __init__(self, weight, bias=None):
        self.weight, self.bias = weight, bias


class _BatchStatBN(object):
    """F.batch_norm(x, running_mu, running_std, w, b, training=True) (efficientdet.py:673): batch statistics, default momentum
    0.1 / eps 1e-5, the head's two running buffers shared by every layer and level (updated in place like the reference)."""
    training, momentum, eps = True, 0.1, 1e-5

    def __init__(self, weight, bias, running_mean, running_var, counter):
        self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked = weight, bias, running_mean, running_var, counter


class _HeadKernels(TrainEngine):
    """the conv / BN building blocks of TrainEngine without a model behind them"""

    def __init__(self, dev):
        self.dev = dev
        self.ops = _Ops(dev)
        self.lib = self.ops.lib
        self._ones = {}
        self._stage = None                       # never inside a TrainEngine stage: derived weights are computed on the spot
        self._tables = {}
        self.use_tables = False


_kernels = {}
_warned = [False]


def _engine(dev):
    key = (dev.type, dev.index)
    if key not in _kernels:
        _kernels[key] = _HeadKernels(dev)
    return _kernels[key]


class MetaHeadFn(torch.autograd.Function):
    """tensors = x[levels] (NHWC float32) + conv_dw[R] + conv_pw[R] + conv_pb[R] + predict[3] + bn_w[L*R] + bn_b[L*R]
    (+ predict_class[2]); outputs = out[levels] + x_pred[levels] (+ class_out[levels]), all NHWC."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        R, L, levels, both = meta['R'], meta['L'], meta['levels'], meta['both']
        nl = len(levels)
        xs = [t.contiguous() for t in tensors[:nl]]
        w = list(tensors[nl:])
        dw, pw, pb = w[:R], w[R:2 * R], w[2 * R:3 * R]
        pred = w[3 * R:3 * R + 3]
        bn_w = w[3 * R + 3:3 * R + 3 + R * L]
        bn_b = w[3 * R + 3 + R * L:3 * R + 3 + 2 * R * L]
        pcls = w[3 * R + 3 + 2 * R * L:] if both else None
        eng = _engine(xs[0].device)
        counter = torch.zeros((), dtype=torch.int64, device=xs[0].device)
        recs, outs, preds, couts = [], [], [], []
        for li, level in enumerate(levels):
            t = xs[li]
            lrec = dict(reps=[])
            for r in range(R):
                d, rdw = eng._dw_fwd(t, _Conv(dw[r]), 'dw%d.' % r)
                c, rpw = eng._pw_fwd(d, _Conv(pw[r], pb[r]), 'pw%d.' % r)
                bn = _BatchStatBN(bn_w[level * R + r], bn_b[level * R + r], meta['running_mu'], meta['running_std'], counter)
                (y, t), rbn = eng._bn_fwd(c, bn, 'bn%d_%d.' % (level, r), silu_out=True)
                lrec['reps'].append((rdw, rpw, rbn, y))
            d, rdw = eng._dw_fwd(t, _Conv(pred[0]), 'pred_dw.')
            o, rpw = eng._pw_fwd(d, _Conv(pred[1], pred[2]), 'pred_pw.')
            lrec['predict'] = (rdw, rpw)
            outs.append(o)
            preds.append(d)
            if both:
                co, rpc = eng._pw_fwd(d, _Conv(pcls[0], pcls[1]), 'pcls.')
                lrec['pcls'] = rpc
                couts.append(co)
            recs.append(lrec)
        ctx.meta, ctx.recs, ctx.nl = meta, recs, nl
        return tuple(outs + preds + couts)

    @staticmethod
    def backward(ctx, *gouts):
        meta, recs, nl = ctx.meta, ctx.recs, ctx.nl
        R, L, levels, both = meta['R'], meta['L'], meta['levels'], meta['both']
        if any(g is not None and g.requires_grad for g in gouts) and not _warned[0]:
            _warned[0] = True
            warnings.warn('MetaHead gradients are first order: create_graph=True keeps no second-order (MAML) terms')
        eng = _engine(recs[0]['predict'][1]['x'].device)
        ops = eng.ops
        grads, dxs = {}, []
        for li in range(nl):
            lrec = recs[li]
            g_out, g_pred = gouts[li], gouts[nl + li]
            g_cls = gouts[2 * nl + li] if both else None
            rdw, rpw = lrec['predict']
            dd = None
            if g_out is not None:
                dd = eng._pw_bwd(rpw, g_out.contiguous(), grads)
            if g_cls is not None:
                dc = eng._pw_bwd(lrec['pcls'], g_cls.contiguous(), grads)
                dd = dc if dd is None else ops.add(dd, dc)
            if g_pred is not None:
                dd = g_pred.contiguous() if dd is None else ops.add(dd, g_pred.contiguous())
            if dd is None:
                dxs.append(None)
                continue
            da = eng._dw_bwd(rdw, dd, grads)
            for (rdw_, rpw_, rbn, y) in reversed(lrec['reps']):
                dy = ops.silu_bwd(y, da)
                dc = eng._bn_bwd(rbn, dy, grads)
                dd = eng._pw_bwd(rpw_, dc, grads)
                da = eng._dw_bwd(rdw_, dd, grads)
            dxs.append(da)
        ctx.recs = None
        z = lambda name: grads.get(name)
        gw = [z('dw%d.weight' % r) for r in range(R)] + [z('pw%d.weight' % r) for r in range(R)] + [z('pw%d.bias' % r) for r in range(R)]
        gw += [z('pred_dw.weight'), z('pred_pw.weight'), z('pred_pw.bias')]
        gw += [z('bn%d_%d.weight' % (lev, r)) for lev in range(L) for r in range(R)]
        gw += [z('bn%d_%d.bias' % (lev, r)) for lev in range(L) for r in range(R)]
        if both:
            gw += [z('pcls.weight'), z('pcls.bias')]
        return (None,) + tuple(dxs) + tuple(gw)


def meta_head_train_forward(mh, x, conv_dw_rep, conv_pw_rep, conv_pb_rep, bn_rep_w, bn_rep_b, predict, predict_class, levels,
                            ret_activs, both):
    """the differentiable forward; returns what MetaHead.forward returns (NCHW-shaped views)"""
    x0 = x[levels[0]]
    if x0.device.type != 'cuda':
        raise RuntimeError('MetaHead runs on the GPU only (no CPU fallback)')
    tens = list(conv_dw_rep) + list(conv_pw_rep) + list(conv_pb_rep) + list(predict) + list(bn_rep_w) + list(bn_rep_b)
    if both:
        tens += list(predict_class)
    if any(t.dtype != torch.float32 for t in tens) or x0.dtype != torch.float32:
        raise RuntimeError('the differentiable MetaHead path is float32 (the reference trains in fp32); use torch.no_grad() for '
                           'bfloat16 inference')
    xs = [x[l].permute(0, 2, 3, 1) for l in levels]
    nl = len(levels)
    if not getattr(mh, 'first_order', False):
        from . import meta_ops
        o, a, c = meta_ops.meta_head_forward([t.contiguous() for t in xs], conv_dw_rep, conv_pw_rep, conv_pb_rep, bn_rep_w, bn_rep_b,
                                             predict, predict_class if both else None, list(levels), mh.num_layers,
                                             mh.running_mu, mh.running_std)
        res = tuple(o) + tuple(a) + (tuple(c) if both else ())
    else:
        meta = dict(R=mh.num_layers, L=mh.num_levels, levels=list(levels), both=both, running_mu=mh.running_mu, running_std=mh.running_std)
        res = MetaHeadFn.apply(meta, *(xs + tens))
    nchw = lambda t: t.permute(0, 3, 1, 2)
    outputs = [nchw(t) for t in res[:nl]]
    activs = [nchw(t) for t in res[nl:2 * nl]]
    if both:
        class_outputs = [nchw(t) for t in res[2 * nl:]]
        return (class_outputs, outputs, activs) if ret_activs else (class_outputs, outputs)
    return (outputs, activs) if ret_activs else outputs
