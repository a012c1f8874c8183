"""`MetaHead` (reference: effdet/efficientdet.py:569-695): the fork's functional class head for few-shot episodes.

`box_class_repeats` SeparableConv layers shared by the pyramid levels, each followed by `F.batch_norm(training=True)`
- statistics of the CURRENT batch, separately per level and layer - and Swish, then a depthwise 3x3 + pointwise
predict conv to one logit per anchor.  `fast_weights` (the MAML inner-loop copies, infer.py:561,681) replace the
module's own parameters in the reference's list order; `ret_activs` returns the predict layer's depthwise output.

HIP path: one `effdet_sepconv_meta` launch per layer for all levels, `effdet_bn_batch_stats` in between (the batch
statistics are folded into a per-level scale / shift that the next layer applies while loading its halo tile).
With grad mode on and trainable weights / inputs the forward is the differentiable float32 path (effdet/meta_ops.py: autograd
primitives on the training kernels, differentiable twice - the `create_graph=True` inner gradient of infer.py:658 and the outer
backward through it; `first_order = True` selects the single-node backward of effdet/meta_grad.py instead)."""
import math
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


class MetaHead(nn.Module):
    first_order = False     # True: single-node backward (effdet/meta_grad.py), drops MAML's second-order terms; default: meta_ops.py

    def __init__(self, config, pretrain_init=None, num_channels_flag=None):
        super().__init__()
        self.num_layers = config.box_class_repeats
        self.num_levels = config.num_levels
        in_channels = config.fpn_channels
        self.num_channels = num_channels = num_channels_flag or config.fpn_channels
        self.num_anchors = num_anchors = len(config.aspect_ratios) * config.num_scales
        if pretrain_init is None:
            raise ValueError('MetaHead is initialised from a class_net state dict (pretrain_init), as in infer.py:186-191')
        g = lambda k: nn.Parameter(pretrain_init[k].detach().clone())
        for l in range(self.num_layers):
            setattr(self, 'conv_dw%d' % l, g('class_net.conv_rep.%d.conv_dw.weight' % l))
        for l in range(self.num_layers):
            setattr(self, 'conv_pw%d' % l, g('class_net.conv_rep.%d.conv_pw.weight' % l))
        for l in range(self.num_layers):
            setattr(self, 'conv_pb%d' % l, g('class_net.conv_rep.%d.conv_pw.bias' % l))
        self.register_buffer('running_mu', torch.zeros(num_channels), persistent=False)     # touched by F.batch_norm in the
        self.register_buffer('running_std', torch.ones(num_channels), persistent=False)     # reference, never read back
        self.predict_dw = g('class_net.predict.conv_dw.weight')
        self.predict_pw = nn.Parameter(torch.randn((num_anchors, num_channels, 1, 1)) * ((1 / num_channels) ** 0.5))
        self.predict_pb = nn.Parameter(torch.full([num_anchors], -math.log((1 - 0.01) / 0.01)))
        for lev in range(self.num_levels):
            for rep in range(self.num_layers):
                setattr(self, 'bn_w%d%d' % (rep, lev), g('class_net.bn_rep.%d.%d.bn.weight' % (rep, lev)))
        for lev in range(self.num_levels):
            for rep in range(self.num_layers):
                setattr(self, 'bn_b%d%d' % (rep, lev), g('class_net.bn_rep.%d.%d.bn.bias' % (rep, lev)))
        assert in_channels == num_channels, 'the reference builds every layer from the class_net weights (F -> F)'

    # parameter lists in the reference's order (efficientdet.py:594-632)
    @property
    def conv_dw_rep(self): return [getattr(self, 'conv_dw%d' % l) for l in range(self.num_layers)]
    @property
    def conv_pw_rep(self): return [getattr(self, 'conv_pw%d' % l) for l in range(self.num_layers)]
    @property
    def conv_pb_rep(self): return [getattr(self, 'conv_pb%d' % l) for l in range(self.num_layers)]
    @property
    def predict(self): return [self.predict_dw, self.predict_pw, self.predict_pb]
    @property
    def bn_rep_w(self): return [getattr(self, 'bn_w%d%d' % (rep, lev)) for lev in range(self.num_levels) for rep in range(self.num_layers)]
    @property
    def bn_rep_b(self): return [getattr(self, 'bn_b%d%d' % (rep, lev)) for lev in range(self.num_levels) for rep in range(self.num_layers)]

    def add_head(self):
        self.predict_pw_sep = nn.Parameter(torch.randn((self.num_anchors, self.num_channels, 1, 1)) * ((1 / self.num_channels) ** 0.5))
        self.predict_pb_sep = nn.Parameter(torch.full([self.num_anchors], -math.log((1 - 0.01) / 0.01)))

    @property
    def predict_class(self): return [self.predict_pw_sep, self.predict_pb_sep]

    def forward(self, x: List[torch.Tensor], fast_weights: Optional[List[torch.Tensor]] = None, ret_activs=False,
                level_offset=0, heads='anch'):
        nl_, L = self.num_layers, self.num_levels
        if fast_weights is None:
            conv_dw_rep, conv_pw_rep, conv_pb_rep = self.conv_dw_rep, self.conv_pw_rep, self.conv_pb_rep
            bn_rep_w, bn_rep_b, predict = self.bn_rep_w, self.bn_rep_b, self.predict
        else:                                                         # efficientdet.py:645-652
            conv_dw_rep = fast_weights[:nl_]
            conv_pw_rep = fast_weights[nl_:2 * nl_]
            conv_pb_rep = fast_weights[2 * nl_:3 * nl_]
            predict = fast_weights[3 * nl_:3 * nl_ + 3]
            bn_rep_w = fast_weights[3 * nl_ + 3:3 * nl_ + 3 + nl_ * L]
            bn_rep_b = fast_weights[3 * nl_ + 3 + nl_ * L:]
            heads = 'class'
        both = heads == 'both' and hasattr(self, 'predict_pw_sep')     # FLAGS.separate_head in the reference
        levels = list(range(level_offset, len(x)))
        if not levels:
            return ([], []) if ret_activs else []
        x0 = x[levels[0]]
        if x0.device.type != 'cuda' or x0.dtype not in _DT:
            raise RuntimeError('MetaHead runs on the GPU in float32 / bfloat16 only (no CPU fallback)')
        if torch.is_grad_enabled() and any(t.requires_grad for t in list(x) + list(conv_dw_rep) + list(conv_pw_rep) + list(conv_pb_rep) +
                                           list(predict) + list(bn_rep_w) + list(bn_rep_b)):
            # the MAML inner / outer loop (infer.py:561-681): differentiable float32 path on the training kernels, gradients
            # (to second order) with respect to every (fast) weight and every input level
            from .meta_grad import meta_head_train_forward
            return meta_head_train_forward(self, x, conv_dw_rep, conv_pw_rep, conv_pb_rep, bn_rep_w, bn_rep_b, predict,
                                           self.predict_class if both else None, levels, ret_activs, both)
        lib = _lib.load()
        dev, dtype, dt = x0.device, x0.dtype, _DT[x0.dtype]
        B, F, A = x0.shape[0], self.num_channels, self.num_anchors
        nl = len(levels)
        hw = [(x[l].shape[2], x[l].shape[3]) for l in levels]
        import ctypes
        c_hw = (ctypes.c_int * (2 * nl))(*[v for pair in hw for v in pair])
        # inputs as NHWC memory: engine pyramid views are taken as they are, anything else is packed once
        keep, in_ptr, in_stride = [], [], []
        for l, (h, w) in zip(levels, hw):
            t = x[l].permute(0, 2, 3, 1)
            if not (t.stride(3) == 1 and t.stride(2) == F and t.stride(1) == w * F and t.stride(0) >= h * w * F and t.data_ptr() % 16 == 0):
                t = t.contiguous()
            keep.append(t)
            in_ptr.append(t.data_ptr()); in_stride.append(t.stride(0))
        P = sum(h * w for h, w in hw)
        offs = [sum(h * w for h, w in hw[:i]) for i in range(nl)]
        ybuf = [torch.empty(B, P, F, dtype=dtype, device=dev) for _ in range(2)]
        out = torch.empty(B, P, A, dtype=dtype, device=dev)
        tiles = lib.effdet_sepconv_tiles(dt, nl, c_hw, None)
        partial = torch.empty(B, tiles, 2, F, dtype=torch.float32, device=dev)
        sc = torch.empty(nl, F, dtype=torch.float32, device=dev)
        sh = torch.empty(nl, F, dtype=torch.float32, device=dev)
        rows = (ctypes.c_int * nl)(*range(nl))
        st = torch.cuda.current_stream(dev).cuda_stream
        es = x0.element_size()

        def ptrs(vals):
            return (ctypes.c_void_p * len(vals))(*vals)

        def lls(vals):
            return (ctypes.c_longlong * len(vals))(*vals)

        def taps(wdw):          # [F,1,3,3] -> [9][F] float32
            return wdw.detach().reshape(F, 9).t().contiguous().to(device=dev, dtype=torch.float32)

        cur_ptr, cur_stride, have_affine = ptrs(in_ptr), lls(in_stride), False
        for rep in range(nl_):
            yb = ybuf[rep % 2]
            dw = taps(conv_dw_rep[rep])
            pw = conv_pw_rep[rep].detach().reshape(F, F).to(device=dev, dtype=dtype).contiguous()
            pb = conv_pb_rep[rep].detach().to(device=dev, dtype=torch.float32).contiguous()
            optr = ptrs([yb.data_ptr() + o * F * es for o in offs])
            ostr = lls([P * F] * nl)
            _lib.check(lib.effdet_sepconv_meta(st, dt, B, nl, c_hw, cur_ptr, cur_stride,
                                               sc.data_ptr() if have_affine else None, sh.data_ptr() if have_affine else None,
                                               rows, 1 if have_affine else 0, dw.data_ptr(), pw.data_ptr(), pb.data_ptr(), F, F,
                                               optr, ostr, partial.data_ptr(), None, None), 'effdet_sepconv_meta')
            bw = torch.stack([bn_rep_w[l * nl_ + rep].detach().to(device=dev, dtype=torch.float32) for l in levels]).contiguous()
            bb = torch.stack([bn_rep_b[l * nl_ + rep].detach().to(device=dev, dtype=torch.float32) for l in levels]).contiguous()
            _lib.check(lib.effdet_bn_batch_stats(st, dt, partial.data_ptr(), B, nl, c_hw, F, bw.data_ptr(), bb.data_ptr(), rows,
                                                 1e-5, sc.data_ptr(), sh.data_ptr()), 'effdet_bn_batch_stats')
            keep += [dw, pw, pb, bw, bb]
            cur_ptr, cur_stride, have_affine = optr, ostr, True
        # predict: depthwise on act(bn(x)) -> x_pred -> pointwise to one logit per anchor
        want_pred = ret_activs or both
        xpred = [torch.empty(B, h * w, F, dtype=dtype, device=dev) for h, w in hw] if want_pred else None
        dw = taps(predict[0])
        pw = predict[1].detach().reshape(A, F).to(device=dev, dtype=dtype).contiguous()
        pb = predict[2].detach().to(device=dev, dtype=torch.float32).contiguous()
        _lib.check(lib.effdet_sepconv_meta(st, dt, B, nl, c_hw, cur_ptr, cur_stride,
                                           sc.data_ptr() if have_affine else None, sh.data_ptr() if have_affine else None,
                                           rows, 1 if have_affine else 0, dw.data_ptr(), pw.data_ptr(), pb.data_ptr(), F, A,
                                           ptrs([out.data_ptr() + o * A * es for o in offs]), lls([P * A] * nl), None,
                                           ptrs([t.data_ptr() for t in xpred]) if want_pred else None,
                                           lls([h * w * F for h, w in hw]) if want_pred else None), 'effdet_sepconv_meta')
        outputs = [out[:, o:o + h * w].view(B, h, w, A).permute(0, 3, 1, 2) for o, (h, w) in zip(offs, hw)]
        activs = [t.view(B, h, w, F).permute(0, 3, 1, 2) for t, (h, w) in zip(xpred, hw)] if want_pred else None
        if both:
            pws = self.predict_pw_sep.detach().reshape(A, F).to(device=dev, dtype=dtype).contiguous()
            pbs = self.predict_pb_sep.detach().to(device=dev, dtype=torch.float32).contiguous()
            class_outputs = []
            for t, (h, w) in zip(xpred, hw):
                co = torch.empty(B, h * w, A, dtype=dtype, device=dev)
                _lib.check(lib.effdet_pw_gemm_bn_act(st, dt, t.data_ptr(), B * h * w, F, pws.data_ptr(), A, None, pbs.data_ptr(), 0,
                                                     None, None, h * w, co.data_ptr(), 0, 0), 'effdet_pw_gemm_bn_act')
                class_outputs.append(co.view(B, h, w, A).permute(0, 3, 1, 2))
            return (class_outputs, outputs, activs) if ret_activs else (class_outputs, outputs)
        return (outputs, activs) if ret_activs else outputs
