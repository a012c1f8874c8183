"""`AnchorNet` and `ProjectionNet` (reference: effdet/efficientdet.py:697-830), the two small networks the fork's
few-shot episode code (infer.py) puts next to `MetaHead`.  Forward passes only, on the HIP kernels:

* `AnchorNet`: a HeadNet-shaped tower (SeparableConv + per-level BatchNorm (eval statistics) + Swish, then a
  SeparableConv to 9 outputs) -> `effdet_sepconv_fused`, one launch per layer for all levels.
* `ProjectionNet`: bias-free MLP with ReLU over `[features | anchor enc (8) | cell enc (28) | level enc (6)]`
  -> `effdet_pw_gemm_bn_act` (act = ReLU); the sinusoidal encoding tables are the reference's; `weighted_median`
  (per-dimension median weighted by anchor confidence) -> `effdet_weighted_median`.
The reference reads absl FLAGS for the layer counts; here they are constructor arguments with the flag defaults."""
import ctypes
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _lib
from .efficientdet import SeparableConv2d, _bn

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _arr(ct, vals):
    return (ct * len(vals))(*vals)


def _nhwc(t, F):
    """[B,F,H,W] logical tensor -> (tensor kept alive, data_ptr, image stride) of NHWC memory (zero-copy when possible)."""
    v = t.permute(0, 2, 3, 1)
    h, w = t.shape[2], t.shape[3]
    if not (v.stride(3) == 1 and v.stride(2) == F and v.stride(1) == w * F and v.stride(0) >= h * w * F and v.data_ptr() % 16 == 0):
        v = v.contiguous()
    return v, v.data_ptr(), v.stride(0)


class AnchorNet(nn.Module):
    def __init__(self, config, at_start=True, num_anch_layers=3, num_channels=88, alpha=None):
        super().__init__()
        self.config = config
        self.num_levels = config.num_levels
        self.alpha = alpha                                     # FLAGS.supp_alpha / inner_alpha in the reference
        F = config.fpn_channels
        if num_anch_layers == 1:
            self.conv_rep = nn.ModuleList()
            out_in = F
        else:
            self.conv_rep = nn.ModuleList([SeparableConv2d(F, num_channels, 3, bias=True)] +
                                          [SeparableConv2d(num_channels, num_channels, 3, bias=config.redundant_bias)
                                           for _ in range(num_anch_layers - 2)])
            out_in = num_channels
        self.bn_rep = nn.ModuleList()
        for _ in range(num_anch_layers - 1):
            self.bn_rep.append(nn.ModuleList([nn.Sequential(OrderedDict([('bn', _bn(config, num_channels))]))
                                              for _ in range(self.num_levels)]))
        self.anchor_out = SeparableConv2d(out_in, 9, 3, bias=True)
        from .efficientdet import _init_weight
        for n, m in self.named_modules():
            _init_weight(m, n)

    def forward(self, x):
        x0 = x[0]
        if x0.device.type != 'cuda' or x0.dtype not in _DT:
            raise RuntimeError('AnchorNet runs on the GPU in float32 / bfloat16 only (no CPU fallback)')
        if self.training and len(self.bn_rep):
            raise NotImplementedError('AnchorNet HIP path: BatchNorm in eval mode only (call .eval())')
        lib = _lib.load()
        dev, dtype, dt = x0.device, x0.dtype, _DT[x0.dtype]
        B, es = x0.shape[0], x0.element_size()
        nl = len(x)
        hw = [(t.shape[2], t.shape[3]) for t in x]
        P = sum(h * w for h, w in hw)
        offs = [sum(h * w for h, w in hw[:i]) for i in range(nl)]
        c_hw = _arr(ctypes.c_int, [v for p_ in hw for v in p_])
        c_mode = _arr(ctypes.c_int, [0] * nl)
        c_fw = _arr(ctypes.c_float, [0.0, 0.0, 0.0])
        st = torch.cuda.current_stream(dev).cuda_stream
        keep, ptrs, strides = [], [], []
        Fin = x0.shape[1]
        for t in x:
            v, p_, s_ = _nhwc(t, Fin)
            keep.append(v); ptrs.append(p_); strides.append(s_)

        def launch(conv, Fi, No, scale, shift, rows, post_act, in_ptrs, in_strides, out_t):
            taps = conv.conv_dw.weight.detach().reshape(Fi, 9).t().contiguous().to(device=dev, dtype=torch.float32)
            wq = conv.conv_pw.weight.detach().reshape(No, Fi).to(device=dev, dtype=dtype).contiguous()
            keep.extend([taps, wq, scale, shift])
            _lib.check(lib.effdet_sepconv_fused(
                st, dt, B, nl, c_hw, 1, _arr(ctypes.c_void_p, in_ptrs), _arr(ctypes.c_longlong, in_strides), c_hw, c_mode,
                0, c_fw, ctypes.c_float(1.0), 0, taps.data_ptr(), wq.data_ptr(), scale.data_ptr() if scale is not None else None,
                shift.data_ptr(), _arr(ctypes.c_int, rows), post_act, Fi, No,
                _arr(ctypes.c_void_p, [out_t.data_ptr() + o * No * es for o in offs]), _arr(ctypes.c_longlong, [P * No] * nl),
                0, 9, None, None, 0, None), 'effdet_sepconv_fused')

        cur_ptrs, cur_strides, Fi = ptrs, strides, Fin
        for conv, bns in zip(self.conv_rep, self.bn_rep):
            No = conv.conv_pw.weight.shape[0]
            ss, ts = [], []
            for l in range(nl):
                bn = bns[l].bn
                s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                t = bn.bias.detach().float() - bn.running_mean.detach().float() * s
                if conv.conv_pw.bias is not None:
                    t = t + conv.conv_pw.bias.detach().float() * s
                ss.append(s); ts.append(t)
            scale = torch.stack(ss).to(dev).contiguous(); shift = torch.stack(ts).to(dev).contiguous()
            y = torch.empty(B, P, No, dtype=dtype, device=dev)
            keep.append(y)
            launch(conv, Fi, No, scale, shift, list(range(nl)), 1, cur_ptrs, cur_strides, y)
            cur_ptrs = [y.data_ptr() + o * No * es for o in offs]
            cur_strides = [P * No] * nl
            Fi = No
        out = torch.empty(B, P, 9, dtype=dtype, device=dev)
        bias = self.anchor_out.conv_pw.bias.detach().float().reshape(1, 9).to(dev).contiguous()
        launch(self.anchor_out, Fi, 9, None, bias, [0] * nl, 0, cur_ptrs, cur_strides, out)
        return [out[:, o:o + h * w].view(B, h, w, 9).permute(0, 3, 1, 2) for o, (h, w) in zip(offs, hw)]


class ProjectionNet(nn.Module):
    def __init__(self, config, width, proj_depth=2, dot_mult=1.0, dot_add=0.0):
        super().__init__()
        self.dot_mult = nn.Parameter(torch.tensor(float(dot_mult)))
        self.dot_add = nn.Parameter(torch.tensor(float(dot_add)))

        def enc(step, count, freqs):                    # efficientdet.py:706-732
            locs = (torch.arange(start=-1., end=1., step=step) * 3.14159)[:count]
            rows = []
            for freq in range(freqs):
                rows.append(torch.sin(2 ** freq * locs))
                rows.append(torch.cos(2 ** freq * locs))
            return torch.stack(rows).transpose(0, 1).contiguous()

        self.register_buffer('anch_enc', enc(1 / 8, 9, 4), persistent=False)       # [9, 8]
        self.register_buffer('cell_enc', enc(1 / 64, 80, 7), persistent=False)     # [80, 14]
        self.register_buffer('lev_enc', enc(1 / 4, 5, 3), persistent=False)        # [5, 6]
        self.width = width
        d_in = config.fpn_channels + 8 + 28 + 6
        dims = [d_in] + [width] * (proj_depth - 1) + [int(width / 2)]
        layers = []
        for i in range(proj_depth):
            layers.append(nn.Linear(dims[i], dims[i + 1], bias=False))
            if i + 1 < proj_depth:
                layers.append(nn.ReLU())
        self.projection = nn.Sequential(*layers)

    def forward(self, x):
        """x [..., fpn_channels + 42] -> [..., width / 2]: bias-free Linear + ReLU chain as MFMA GEMMs."""
        if x.device.type != 'cuda' or x.dtype not in _DT:
            raise RuntimeError('ProjectionNet runs on the GPU in float32 / bfloat16 only (no CPU fallback)')
        lib = _lib.load()
        dev, dtype, dt = x.device, x.dtype, _DT[x.dtype]
        lead = x.shape[:-1]
        cur = x.reshape(-1, x.shape[-1])
        M = cur.shape[0]
        st = torch.cuda.current_stream(dev).cuda_stream
        linears = [m for m in self.projection if isinstance(m, nn.Linear)]
        for i, lin in enumerate(linears):
            K, N = lin.in_features, lin.out_features
            Kp = (K + 7) // 8 * 8                      # the GEMM walks K in 16-byte pieces: zero-pad 106 -> 112
            a = cur
            if Kp != K or not a.is_contiguous():
                a = torch.zeros(M, Kp, dtype=dtype, device=dev)
                a[:, :K] = cur
            w = torch.zeros(N, Kp, dtype=dtype, device=dev)
            w[:, :K] = lin.weight.detach().to(device=dev, dtype=dtype)
            zero = torch.zeros(N, dtype=torch.float32, device=dev)
            out = torch.empty(M, N, dtype=dtype, device=dev)
            _lib.check(lib.effdet_pw_gemm_bn_act(st, dt, a.data_ptr(), M, Kp, w.data_ptr(), N, None, zero.data_ptr(),
                                                 2 if i + 1 < len(linears) else 0, None, None, 0, out.data_ptr(), 0, 0),
                       'effdet_pw_gemm_bn_act')
            cur = out
        return cur.reshape(*lead, cur.shape[-1])

    def weighted_median(self, embds, confs):
        """efficientdet.py:748-760: per embedding dimension, the value at which the confidence mass reaches one half.
        embds [n, d], confs [n] (float32, GPU) -> (median [1, d], conf_sum)."""
        if embds.device.type != 'cuda' or embds.dtype != torch.float32 or confs.dtype != torch.float32 or embds.dim() != 2:
            raise RuntimeError('weighted_median expects float32 GPU tensors embds [n, d], confs [n]')
        n, d = embds.shape
        if n > 1024:
            raise NotImplementedError('weighted_median is built for up to 1024 anchors per object')
        lib = _lib.load()
        e, c = embds.detach().contiguous(), confs.detach().contiguous()
        med = torch.empty(1, d, dtype=torch.float32, device=embds.device)
        csum = torch.empty(1, dtype=torch.float32, device=embds.device)
        st = torch.cuda.current_stream(embds.device).cuda_stream
        _lib.check(lib.effdet_weighted_median(st, e.data_ptr(), c.data_ptr(), n, d, med.data_ptr(), csum.data_ptr()), 'effdet_weighted_median')
        return med, csum[0]
