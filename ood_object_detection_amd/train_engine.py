"""Forward-with-saved-activations and backward of the pretrain step on libeffdet_hip.so.

Reference: pretrain.py:226-236 (`model(x, mode='bb')` -> `model(feats, mode='fpn_and_head')` -> `loss_fn` ->
`qry_loss.backward()`), with the module semantics of timm's EfficientNet blocks and effdet/efficientdet.py:42-469.
float32 only (the reference trains in fp32).  BatchNorm follows each module's own `.training` flag like nn.BatchNorm2d:
batch statistics (+ running-stat update) where it is set - by default BiFPN and heads - and running statistics where
it is not (pretrain.py:168-176 puts the backbone's BN in eval mode; that is the only backbone mode built).
Stochastic depth of the timm backbone (`drop_path_rate` in backbone_args) IS applied while the backbone module is in training
mode (per-sample keep masks, timm's drop_path); with it off (rate 0 or backbone.eval()) the step is
deterministic, and bitwise reproducible (all reductions are fixed-order).

Division of labour: every activation-sized operation (anything O(B*H*W*C)) is a HIP kernel of csrc/train_net.hip (plus
the forward kernels effdet_dwconv_bn_act / effdet_maxpool_same / effdet_se_gate); PyTorch does parameter-sized glue only
(folding BN into conv weights, transposing weights, the closed-form chain rule from the kernels' raw sums to
d weight / d gamma / d beta / d edge_weights) and owns memory, streams and the autograd graph the two stage
functions (`BackboneFn`, `FpnHeadFn`) plug into, so `loss.backward()` fills `.grad` exactly like the reference.
There is no CPU fallback.
"""
import ctypes
import math

import torch

from . import _lib


def _same_out(n, s):
    return (n + s - 1) // s


_FPN_METHODS = {'fastattn': 0, 'attn': 1, 'sum': 2}


class _PrepOp(ctypes.Structure):               # csrc/train_net.hip PrepOp
    _fields_ = [('kind', ctypes.c_int), ('rows', ctypes.c_int), ('cols', ctypes.c_int), ('eps', ctypes.c_float)] + \
               [(n, ctypes.c_void_p) for n in ('src', 'gamma', 'beta', 'mean', 'var', 'dst0', 'dst1', 'dst2', 'scale', 'shift', 'rstd')]


class _GradOp(ctypes.Structure):               # csrc/train_net.hip GradOp
    _fields_ = [(n, ctypes.c_void_p) for n in ('dWext', 'W', 'scale', 'rstd', 'mean', 'dW', 'dgamma', 'dbeta')] + \
               [('N', ctypes.c_int), ('K', ctypes.c_int), ('transposed', ctypes.c_int), ('pad', ctypes.c_int)]


class _StageTables(object):
    """The parameter-sized work of one stage (backbone, or BiFPN + heads) as ONE launch per kind instead of one per conv:
    the derived weights of a step (BN folded into conv weights, transposed copies for the dX GEMMs, tap-major depthwise taps,
    the transposed SE expand weight) and the closed-form conv + BN parameter gradients.  The first forward / backward records
    what the stage needs (and computes it conv by conv); from then on `run_prep()` at the start of the forward refreshes every
    derived tensor - they live in persistent buffers - from the current parameters, and `run_grads()` at the end of the backward
    turns the raw GEMM sums into parameter gradients.  Keys are parameter names, so the stage must keep its module structure."""

    def __init__(self, ops):
        self.ops, self.lib = ops, ops.lib
        self.prep, self.grad = {}, {}              # key -> (op record, tensors kept alive)
        self.ptab = self.gtab = None
        self.frozen = False                        # True once the tables are on the device

    def _check_src(self, entry, src, extra=()):
        # `extra`: (recorded pointer, tensor) pairs of the other sources of the record (BN gamma / beta / running statistics)
        if entry[0].src != src.data_ptr() or any(ptr != t.data_ptr() for ptr, t in extra):
            raise RuntimeError('a parameter was re-allocated after the training engine recorded it; build a new TrainEngine')

    def transpose(self, key, src2d):
        """-> persistent [cols, rows] transpose of the contiguous 2-D parameter view src2d (None while recording: caller computes)"""
        if key in self.prep:
            self._check_src(self.prep[key], src2d)
            return self.prep[key][1][0]
        if self.frozen:
            raise RuntimeError('unrecorded derived weight %r' % (key,))
        rows, cols = src2d.shape
        dst = self.ops.new(cols, rows)
        dst.copy_(src2d.t())
        op = _PrepOp(0, rows, cols, 0.0, src2d.data_ptr(), None, None, None, None, dst.data_ptr(), None, None, None, None, None)
        self.prep[key] = (op, (dst, src2d))
        return dst

    def edge_weights(self, key, ewp, n, method, compute):
        """-> persistent {w0, w1, w2, den} of a BiFPN node's edge_weights parameter ('fastattn' / 'attn')"""
        if key in self.prep:
            self._check_src(self.prep[key], ewp)
            return self.prep[key][1][0]
        if self.frozen:
            raise RuntimeError('unrecorded derived weight %r' % (key,))
        wdev = compute()
        op = _PrepOp(2, n, 1, float(method), ewp.data_ptr(), None, None, None, None, wdev.data_ptr(), None, None, None, None, None)
        self.prep[key] = (op, (wdev, ewp))
        return wdev

    def fold(self, key, W, bn, want_wf, want_wft, want_wt, compute):
        if key in self.prep:
            op = self.prep[key][0]
            self._check_src(self.prep[key], W, ((op.gamma, bn.weight), (op.beta, bn.bias), (op.mean, bn.running_mean),
                                                (op.var, bn.running_var)))
            return self.prep[key][1][0]
        if self.frozen:
            raise RuntimeError('unrecorded derived weight %r' % (key,))
        out = compute()                             # (Wf, WfT, WT, scale, shift, rstd) from effdet_train_fold_bn: these buffers persist
        p = lambda t: None if t is None else t.data_ptr()
        N, K = W.shape
        op = _PrepOp(1, N, K, float(bn.eps), W.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                     bn.running_var.data_ptr(), p(out[0]), p(out[1]), p(out[2]), out[3].data_ptr(), out[4].data_ptr(), out[5].data_ptr())
        self.prep[key] = (op, (out, W, bn))
        return out

    def grad_entry(self, key, rec, transposed):
        """-> (dWext buffer the backward GEMM writes, dW, dgb, deferred) for conv `key`; deferred: run_grads() will fill dW / dgb"""
        if key in self.grad:
            op = self.grad[key][0]
            if op.W != rec['W'].data_ptr() or op.scale != rec['scale'].data_ptr() or op.rstd != rec['rstd'].data_ptr() or \
                    op.mean != rec['mean'].data_ptr():
                raise RuntimeError('a parameter was re-allocated after the training engine recorded it; build a new TrainEngine')
            return self.grad[key][1][:3] + (self.gtab is not None,)
        if self.gtab is not None:
            raise RuntimeError('unrecorded gradient %r' % (key,))
        N, K = rec['W'].shape
        dwext, dW, dgb = self.ops.new(N * K + N), self.ops.new(N, K), self.ops.new(2, N)
        op = _GradOp(dwext.data_ptr(), rec['W'].data_ptr(), rec['scale'].data_ptr(), rec['rstd'].data_ptr(), rec['mean'].data_ptr(),
                     dW.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), N, K, int(transposed), 0)
        self.grad[key] = (op, (dwext, dW, dgb, rec['W'], rec['scale'], rec['rstd'], rec['mean']))
        return dwext, dW, dgb, False

    @staticmethod
    def _upload(ops_list, cls, dev):
        arr = (cls * len(ops_list))(*ops_list)
        return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    def run_prep(self):
        if not self.prep:
            return
        if self.ptab is None:
            recs = [v[0] for v in self.prep.values()]
            self.ptab = self._upload(recs, _PrepOp, self.ops.dev)
            self.pmax = max(r.rows * r.cols for r in recs)
            self.frozen = True
        _lib.check(self.lib.effdet_train_prep_table(self.ops.st(), self.ptab.data_ptr(), len(self.prep), self.pmax),
                   'effdet_train_prep_table')

    def run_grads(self):
        if not self.grad:
            return
        if self.gtab is None:                        # end of the recording backward (it computed conv by conv): table for the next ones
            recs = [v[0] for v in self.grad.values()]
            self.gtab = self._upload(recs, _GradOp, self.ops.dev)
            self.gmax = max(r.N for r in recs)
            return
        _lib.check(self.lib.effdet_train_grads_table(self.ops.st(), self.gtab.data_ptr(), len(self.grad), self.gmax),
                   'effdet_train_grads_table')


class _Levels(object):
    """Geometry of a packed pyramid: the L levels' NHWC tensors [B, h_l, w_l, C] one behind the other (level-major rows)."""

    def __init__(self, B, hw):
        import ctypes
        self.B, self.L, self.hw = B, len(hw), list(hw)
        self.Hs = (ctypes.c_int * self.L)(*[h for h, _ in hw])
        self.Ws = (ctypes.c_int * self.L)(*[w for _, w in hw])
        self.rows = [B * h * w for h, w in hw]
        self.M = sum(self.rows)
        self.P = sum(h * w for h, w in hw)
        self.inv_m = (ctypes.c_float * self.L)(*[1.0 / r for r in self.rows])
        self.unbias = (ctypes.c_float * self.L)(*[float(r / max(r - 1, 1)) for r in self.rows])

    def split(self, t):
        """packed [M, C] -> per-level NHWC views"""
        out, o = [], 0
        for (h, w), r in zip(self.hw, self.rows):
            out.append(t[o:o + r].view(self.B, h, w, t.shape[-1]))
            o += r
        return out


class _Ops(object):
    """Thin tensor-level wrappers of the C ABI (float32 NHWC tensors on one GPU)."""

    def __init__(self, device):
        self.lib = _lib.load()
        self.dev = device
        self._ws = None
        self._retired = []
        # EFFDET_PAD_SYMMETRIC (1 << 24) or 0: the padding convention of the stage that is running (timm pad_type '' vs 'same'),
        # OR-ed into the selector argument of every entry point that pads; set by TrainEngine._in_stage
        self.pad = 0

    def st(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def ws(self, n):
        n = int(n)
        if n < 0:
            raise RuntimeError('workspace query rejected its arguments')
        if self._ws is None or self._ws.numel() < n:
            if self._ws is not None:
                self._retired.append(self._ws)          # a captured hipGraph may still point at it: never hand it back
            self._ws = self.new(max(n, 1 << 20))
        return self._ws

    # ---- GEMMs ----------------------------------------------------------------------------------
    def gemm_nt(self, A, W, bias=None, M=None, a_map=None, out=None, c_map=None, silu_out=False):
        """A [M,K] (dense, or a (ptr, rpi, img_stride, ld) map) x W[N,K]^T + bias -> out [M,N] (and silu(out) when asked)."""
        N, K = W.shape
        if a_map is None:
            M = A.numel() // K
            ap, am = A.data_ptr(), (0, 0, 0)
        else:
            ap, am = a_map[0], a_map[1:]
        if c_map is None:
            out = self.new(M, N)
            cp, cm = out.data_ptr(), (0, 0, 0)
        else:
            cp, cm = c_map[0], c_map[1:]
        out2 = self.new(M, N) if silu_out else None
        _lib.check(self.lib.effdet_train_gemm_nt(self.st(), ap, am[0], am[1], am[2], W.data_ptr(),
                                                 None if bias is None else bias.data_ptr(), cp, cm[0], cm[1], cm[2], M, K, N, 0,
                                                 None if out2 is None else out2.data_ptr()), 'effdet_train_gemm_nt')
        return (out, out2) if silu_out else out

    def gemm_tn(self, dY, X, N, K, M=None, y_map=None, x_map=None, out=None):
        """-> (dW [N,K], dsum [N]) = dY^T [X | 1]"""
        if y_map is None:
            M = dY.numel() // N
            yp, ym = dY.data_ptr(), (0, 0, 0)
        else:
            yp, ym = y_map[0], y_map[1:]
        if x_map is None:
            xp, xm = X.data_ptr(), (0, 0, 0)
        else:
            xp, xm = x_map[0], x_map[1:]
        n = self.lib.effdet_train_gemm_tn_workspace_floats(M, N, K)
        ws = self.ws(n)
        out = self.new(N * K + N) if out is None else out
        _lib.check(self.lib.effdet_train_gemm_tn(self.st(), yp, ym[0], ym[1], ym[2], xp, xm[0], xm[1], xm[2], M, N, K,
                                                 out.data_ptr(), ws.data_ptr(), ws.numel()), 'effdet_train_gemm_tn')
        return out[:N * K].view(N, K), out[N * K:]

    # ---- depthwise ------------------------------------------------------------------------------
    def dw_fwd(self, x, taps, scale, shift, k, s):
        B, H, W, C = x.shape
        y = self.new(B, _same_out(H, s), _same_out(W, s), C)
        _lib.check(self.lib.effdet_train_dwconv_fwd(self.st(), x.data_ptr(), y.data_ptr(), None, taps.data_ptr(), scale.data_ptr(),
                                                    shift.data_ptr(), None, B, H, W, C, k | self.pad, s), 'effdet_train_dwconv_fwd')
        return y

    def dw_fwd_train(self, x, taps, scale, shift, k, s):
        """conv_dw + folded BN -> (z, silu(z), SE pool partial rows of silu(z) [B, nblk, C], nblk) in one pass"""
        B, H, W, C = x.shape
        Ho, Wo = _same_out(H, s), _same_out(W, s)
        z, a = self.new(B, Ho, Wo, C), self.new(B, Ho, Wo, C)
        nblk = self.lib.effdet_train_dwconv_fwd_parts(H, W, C, k | self.pad, s)
        part = self.new(B, nblk, C)
        _lib.check(self.lib.effdet_train_dwconv_fwd(self.st(), x.data_ptr(), z.data_ptr(), a.data_ptr(), taps.data_ptr(),
                                                    scale.data_ptr(), shift.data_ptr(), part.data_ptr(), B, H, W, C, k | self.pad, s),
                   'effdet_train_dwconv_fwd')
        return z, a, part, nblk

    def gemm_nt_fused(self, A, W, bias=None, a_scale=None, a_rows=0, R=None, silu_out=False):
        """(A * a_scale[row // a_rows]) W^T + bias + R  (and silu of it when asked); dense rows"""
        N, K = W.shape
        M = A.numel() // K
        out = self.new(M, N)
        out2 = self.new(M, N) if silu_out else None
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self.lib.effdet_train_gemm_nt_fused(self.st(), A.data_ptr(), p(a_scale), a_rows, W.data_ptr(), p(bias), p(R),
                                                       out.data_ptr(), p(out2), M, K, N), 'effdet_train_gemm_nt_fused')
        return (out, out2) if silu_out else out

    def gemm_tn_scaled(self, dY, X, x_scale, x_rows, N, K, out=None):
        M = dY.numel() // N
        ws = self.ws(self.lib.effdet_train_gemm_tn_workspace_floats(M, N, K))
        out = self.new(N * K + N) if out is None else out
        _lib.check(self.lib.effdet_train_gemm_tn_scaled(self.st(), dY.data_ptr(), X.data_ptr(), x_scale.data_ptr(), x_rows, M, N, K,
                                                        out.data_ptr(), ws.data_ptr(), ws.numel()), 'effdet_train_gemm_tn_scaled')
        return out[:N * K].view(N, K), out[N * K:]

    def dw_bwd(self, dy, x, taps, k, s, z=None, out=None, cmajor=False):
        """-> dx, dtaps [k*k, C], dsum [C]   (taps already carry any folded BN scale); z: dx is multiplied by silu'(z)"""
        B, H, W, C = x.shape
        dx = self.new(B, H, W, C)
        if z is not None:
            _lib.check(self.lib.effdet_train_dwconv_bwd_dx_silu(self.st(), dy.data_ptr(), taps.data_ptr(), z.data_ptr(), dx.data_ptr(),
                                                                B, H, W, C, k | self.pad, s), 'effdet_train_dwconv_bwd_dx_silu')
        else:
            _lib.check(self.lib.effdet_train_dwconv_bwd_dx(self.st(), dy.data_ptr(), taps.data_ptr(), dx.data_ptr(), B, H, W, C, k | self.pad, s),
                       'effdet_train_dwconv_bwd_dx')
        n = self.lib.effdet_train_dwconv_bwd_dw_workspace_floats(B, H, W, C, k | self.pad, s)
        ws = self.ws(n)
        out = self.new(k * k + 1, C) if out is None else out.view(k * k + 1, C)
        _lib.check(self.lib.effdet_train_dwconv_bwd_dw(self.st(), dy.data_ptr(), x.data_ptr(), out.data_ptr(), B, H, W, C, k | self.pad, s,
                                                       ws.data_ptr(), ws.numel(), int(cmajor)), 'effdet_train_dwconv_bwd_dw')
        if cmajor:                                   # tap gradients in the parameter's [C, k*k] layout
            return dx, out.view(-1)[:k * k * C].view(C, k * k), out[k * k]
        return dx, out[:k * k], out[k * k]

    # ---- element-wise ---------------------------------------------------------------------------
    def ew(self, op, a, b=None, c=None, v=(None, None, None, None), s=(0.0, 0.0, 0.0, 0.0), hw=0, sdev=None, silu_out=False):
        """sdev: device float[4] replacing the scalars s (no host read-back: the step stays graph-capturable);
        silu_out: also return silu(result), produced in the same pass"""
        out = torch.empty_like(a)
        out2 = torch.empty_like(a) if silu_out else None
        C = a.shape[-1]
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self.lib.effdet_train_ew(self.st(), op, out.data_ptr(), a.data_ptr(), p(b), p(c), p(v[0]), p(v[1]), p(v[2]), p(v[3]),
                                            s[0], s[1], s[2], s[3], a.numel(), C, hw, p(sdev), p(out2)), 'effdet_train_ew(%d)' % op)
        return (out, out2) if silu_out else out

    def silu(self, z):
        return self.ew(0, z)

    def silu_bwd(self, z, da):
        return self.ew(1, z, da)

    def add(self, a, b):
        return self.ew(2, a, b)

    def col_reduce(self, mode, a, b=None, v=None, per_image=False, alpha=1.0):
        """a: [..., C]; -> alpha * [C] (or [B, C] per image)"""
        C = a.shape[-1]
        G = a.shape[0] if per_image else 1
        R = a.numel() // (C * G)
        n = self.lib.effdet_train_col_reduce_workspace_floats(G, R, C)
        ws = self.ws(n)
        out = self.new(G, 2, C) if mode == 4 else self.new(G, C)
        _lib.check(self.lib.effdet_train_col_reduce(self.st(), mode, a.data_ptr(), None if b is None else b.data_ptr(),
                                                    None if v is None else v.data_ptr(), G, R, C, out.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), alpha), 'effdet_train_col_reduce')
        return out if per_image else out[0]

    def spatial(self, op, x, aux=None):
        B, H, W, C = x.shape
        if op == 0:
            out = self.new(B, 2 * H, 2 * W, C)
            h, w = H, W
        elif op == 1:
            h, w = H // 2, W // 2
            out = self.new(B, h, w, C)
        else:
            out = self.new(B, H, W, C)
            h, w = H, W
        _lib.check(self.lib.effdet_train_spatial(self.st(), op | self.pad, x.data_ptr(), None if aux is None else aux.data_ptr(),
                                                 out.data_ptr(), B, h, w, C), 'effdet_train_spatial(%d)' % op)
        return out

    def maxpool(self, x):
        B, H, W, C = x.shape
        y = self.new(B, _same_out(H, 2), _same_out(W, 2), C)
        _lib.check(self.lib.effdet_maxpool_same(self.st(), 0 | self.pad, x.data_ptr(), 0, y.data_ptr(), 0, B, H, W, C), 'effdet_maxpool_same')
        return y

    # ---- whole-pyramid operators (csrc/train_levels.hip) ------------------------------------------
    def lv_ws(self, lv, C):
        return self.ws(self.lib.effdet_train_levels_workspace_floats(lv.B, lv.L, lv.Hs, lv.Ws, C))

    def lv_dw(self, lv, x, taps, flip=False):
        C = x.shape[-1]
        y = self.new(lv.M, C)
        _lib.check(self.lib.effdet_train_levels_dw(self.st(), x.data_ptr(), taps.data_ptr(), y.data_ptr(), lv.B, lv.L, lv.Hs, lv.Ws,
                                                   C, int(flip)), 'effdet_train_levels_dw')
        return y

    def lv_dw_bwd_dw(self, lv, dy, x, cmajor=False):
        """-> tap gradients [9, C] (cmajor: [C, 9], the parameter's layout)"""
        C = x.shape[-1]
        ws = self.lv_ws(lv, C)
        out = self.new(C, 9) if cmajor else self.new(9, C)
        _lib.check(self.lib.effdet_train_levels_dw_bwd_dw(self.st(), dy.data_ptr(), x.data_ptr(), out.data_ptr(), lv.B, lv.L, lv.Hs,
                                                          lv.Ws, C, ws.data_ptr(), ws.numel(), int(cmajor)), 'effdet_train_levels_dw_bwd_dw')
        return out

    def lv_col_reduce(self, lv, mode, a, b=None, v=None, pre=None, vscale=None):
        C = a.shape[-1]
        ws = self.lv_ws(lv, C)
        out = self.new(lv.L, 2, C) if mode == 4 else self.new(lv.L, C)
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self.lib.effdet_train_levels_col_reduce(self.st(), mode, a.data_ptr(), p(b), p(v), p(pre), vscale, lv.B, lv.L,
                                                           lv.Hs, lv.Ws, C, out.data_ptr(), ws.data_ptr(), ws.numel()),
                   'effdet_train_levels_col_reduce')
        return out

    def lv_bn_finalize(self, lv, sums, sq, bns, C):
        """-> [4, L, C]: mean, scale, shift, rstd of the L per-level BatchNorm layers `bns`"""
        import ctypes
        L = lv.L
        vp = ctypes.c_void_p * L
        out = self.new(4, L, C)
        nbt = vp(*[None if bn.num_batches_tracked is None else bn.num_batches_tracked.data_ptr() for bn in bns])
        _lib.check(self.lib.effdet_train_levels_bn_finalize(
            self.st(), None if sums is None else sums.data_ptr(), None if sq is None else sq.data_ptr(), L, C,
            vp(*[bn.weight.data_ptr() for bn in bns]), vp(*[bn.bias.data_ptr() for bn in bns]),
            vp(*[bn.running_mean.data_ptr() for bn in bns]), vp(*[bn.running_var.data_ptr() for bn in bns]), nbt,
            (ctypes.c_int * L)(*[int(bn.training) for bn in bns]), lv.inv_m, lv.unbias,
            (ctypes.c_float * L)(*[float(bn.momentum if bn.momentum is not None else 0.1) for bn in bns]),
            (ctypes.c_float * L)(*[float(bn.eps) for bn in bns]),
            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr()), 'effdet_train_levels_bn_finalize')
        return out

    def lv_bn_bwd_prep(self, lv, sums, rstd, C):
        """-> [4, L, C]: d gamma, d beta, v1, v3"""
        out = self.new(4, lv.L, C)
        _lib.check(self.lib.effdet_train_levels_bn_bwd_prep(self.st(), sums.data_ptr(), rstd.data_ptr(), lv.inv_m, lv.L, C,
                                                            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                                            out[3].data_ptr()), 'effdet_train_levels_bn_bwd_prep')
        return out

    def lv_ew(self, lv, op, a, b=None, pre=None, v=(None, None, None, None), train=None, silu_out=False):
        C = a.shape[-1]
        out = torch.empty_like(a)
        out2 = torch.empty_like(a) if silu_out else None
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self.lib.effdet_train_levels_ew(self.st(), op, out.data_ptr(), p(out2), a.data_ptr(), p(b), p(pre), p(v[0]), p(v[1]),
                                                   p(v[2]), p(v[3]), train, lv.B, lv.L, lv.Hs, lv.Ws, C), 'effdet_train_levels_ew(%d)' % op)
        return (out, out2) if silu_out else out

    def gemm_nt_levels(self, lv, A, W, bias=None, a_packed=False, out_packed=None, pk=(0, 0)):
        """rows of the packed pyramid x W[N,K]^T + bias; a_packed: A is the image-major head tensor (pk = floats per image, row
        stride); out_packed: write into that image-major tensor instead of returning dense level-major rows"""
        N, K = W.shape
        out = self.new(lv.M, N) if out_packed is None else out_packed
        _lib.check(self.lib.effdet_train_gemm_nt_levels(self.st(), A.data_ptr(), int(a_packed), W.data_ptr(),
                                                        None if bias is None else bias.data_ptr(), out.data_ptr(),
                                                        int(out_packed is not None), lv.B, lv.L, lv.Hs, lv.Ws, pk[0], pk[1], K, N, None),
                   'effdet_train_gemm_nt_levels')
        return out

    def gemm_tn_levels(self, lv, dY, X, N, K, y_packed=False, pk=(0, 0)):
        """-> (dW [N,K], dsum [N]) over the rows of the packed pyramid"""
        ws = self.ws(self.lib.effdet_train_gemm_tn_workspace_floats(lv.M, N, K))
        out = self.new(N * K + N)
        _lib.check(self.lib.effdet_train_gemm_tn_levels(self.st(), dY.data_ptr(), int(y_packed), X.data_ptr(), lv.B, lv.L, lv.Hs, lv.Ws,
                                                        pk[0], pk[1], N, K, out.data_ptr(), ws.data_ptr(), ws.numel()),
                   'effdet_train_gemm_tn_levels')
        return out[:N * K].view(N, K), out[N * K:]

    # ---- FpnCombine without materialised resampled inputs (csrc/train_fpn.hip) ------------------------
    @staticmethod
    def _fpn_srcs(ins):
        import ctypes
        n = len(ins)
        return ((ctypes.c_void_p * n)(*[t.data_ptr() for t in ins]), (ctypes.c_int * n)(*[t.shape[1] for t in ins]),
                (ctypes.c_int * n)(*[t.shape[2] for t in ins]))

    def fpn_weights(self, ewp, n, method):
        wdev = self.new(4)
        _lib.check(self.lib.effdet_train_fpn_weights(self.st(), None if ewp is None else ewp.data_ptr(), n, method, wdev.data_ptr()),
                   'effdet_train_fpn_weights')
        return wdev

    def fpn_combine(self, ins, wdev, method, H, W):
        B, C = ins[0].shape[0], ins[0].shape[-1]
        fused, act = self.new(B, H, W, C), self.new(B, H, W, C)
        sp, hs, ws_ = self._fpn_srcs(ins)
        _lib.check(self.lib.effdet_train_fpn_combine(self.st(), len(ins), sp, hs, ws_, method | self.pad, wdev.data_ptr(), fused.data_ptr(),
                                                     act.data_ptr(), B, H, W, C), 'effdet_train_fpn_combine')
        return fused, act

    def fpn_wgrad(self, ins, wdev, method, ewp, dact, fused):
        B, H, W, C = fused.shape
        n = len(ins)
        ws = self.ws(self.lib.effdet_train_fpn_dots_workspace_floats(B, H, W, C))
        out = self.new(n * C + n)
        sp, hs, ws_ = self._fpn_srcs(ins)
        _lib.check(self.lib.effdet_train_fpn_wgrad(self.st(), n, sp, hs, ws_, method | self.pad, wdev.data_ptr(), ewp.data_ptr(), dact.data_ptr(),
                                                   fused.data_ptr(), out.data_ptr(), out[n * C:].data_ptr(), B, H, W, C,
                                                   ws.data_ptr(), ws.numel()), 'effdet_train_fpn_wgrad')
        return out[n * C:]

    def fpn_input_bwd(self, idx, src, wdev, dact, fused, acc=None):
        B, H, W, C = fused.shape
        out = torch.empty_like(src)
        _lib.check(self.lib.effdet_train_fpn_input_bwd(self.st(), idx | self.pad, src.data_ptr(), src.shape[1], src.shape[2], wdev.data_ptr(),
                                                       dact.data_ptr(), fused.data_ptr(), None if acc is None else acc.data_ptr(),
                                                       out.data_ptr(), B, H, W, C), 'effdet_train_fpn_input_bwd')
        return out

    def reduce_rows(self, t):
        """[S, L] -> [L] in row order"""
        S, L = t.shape
        out = self.new(L)
        _lib.check(self.lib.effdet_train_reduce_mid(self.st(), t.data_ptr(), 1, S, L, out.data_ptr(), 0), 'effdet_train_reduce_mid')
        return out


class TrainEngine(object):
    """Stage functions `bb_forward/backward`, `fh_forward/backward` over one model (its parameters are read live, so an
    optimizer step needs no re-preparation)."""

    def __init__(self, model):
        p0 = model.backbone.conv_stem.weight
        if p0.device.type != 'cuda':
            raise RuntimeError('the training path needs the model on a GPU (cuda:N); there is no CPU fallback')
        if p0.dtype != torch.float32:
            raise RuntimeError('the training path is float32 (the reference trains in fp32; pretrain.py:226)')
        self.model = model
        self.signature = model.train_signature() if hasattr(model, 'train_signature') else None
        self.dev = p0.device
        self.ops = _Ops(self.dev)
        self.lib = self.ops.lib
        cfg = model.config
        self.cfg = cfg
        self.F = cfg.fpn_channels
        self.L = cfg.num_levels
        self.A = model.num_anchors
        # padding conventions (EFFDET_PAD_SYMMETRIC for timm's pad_type ''): the backbone's follows its name, BiFPN / heads config.pad_type
        self.bb_pad = (1 << 24) if getattr(model.backbone, 'pad_type', 'same') == '' else 0
        self.fpn_pad = (1 << 24) if cfg.pad_type == '' else 0
        self._ones = {}
        self._tables = {'bb': _StageTables(self.ops), 'fh': _StageTables(self.ops)}
        self._stage = None                          # the stage whose forward / backward is running (None: called from outside, e.g. meta_grad)
        import os
        self.use_tables = os.environ.get('EFFDET_TRAIN_TABLES', '1') != '0'     # 0: every derived weight / gradient conv by conv (debugging)
        self.direct_grad = False        # True: parameter gradients are added into existing `.grad`s by one multi-tensor launch

    def _const(self, C, v):
        key = (C, v)
        if key not in self._ones:
            self._ones[key] = torch.full((C,), float(v), dtype=torch.float32, device=self.dev)
        return self._ones[key]

    # =============================================================================================
    # conv (+BN) building blocks.  Every *_fwd returns (output, record); *_bwd(record, dy, grads) returns dx.
    # =============================================================================================
    def _fold(self, W, bn, want_wf, want_wft, want_wt, key=None):
        """effdet_train_fold_bn: (Wf, WfT, WT, scale, shift, rstd) of a conv weight [N, K] and its BatchNorm (running stats)"""
        N, K = W.shape
        ops = self.ops

        def compute():
            Wf = ops.new(N, K) if want_wf else None
            WfT = ops.new(K, N) if want_wft else None
            WT = ops.new(K, N) if want_wt else None
            vec = ops.new(3, N)
            p = lambda t: None if t is None else t.data_ptr()
            _lib.check(self.lib.effdet_train_fold_bn(ops.st(), W.data_ptr(), N, K, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                     bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps),
                                                     p(Wf), p(WfT), p(WT), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr()),
                       'effdet_train_fold_bn')
            return Wf, WfT, WT, vec[0], vec[1], vec[2]

        if self._stage is None or key is None:
            return compute()
        return self._tables[self._stage].fold(key, W, bn, want_wf, want_wft, want_wt, compute)

    def _transposed(self, key, src2d):
        """contiguous transpose of a 2-D parameter view: from the stage's table inside a stage, computed on the spot otherwise"""
        if self._stage is None or key is None:
            return src2d.t().contiguous()
        return self._tables[self._stage].transpose(key, src2d)

    def _grad_bufs(self, rec, transposed):
        """-> (buffer for the backward GEMM's raw sums [N*K + N], dW, dgb, deferred)"""
        if self._stage is None or not rec.get('table', True):
            N, K = rec['W'].shape
            return self.ops.new(N * K + N), self.ops.new(N, K), self.ops.new(2, N), False
        return self._tables[self._stage].grad_entry(rec['names'][0], rec, transposed)

    def _convbn_grads(self, rec, bufs, transposed, grads):
        """effdet_train_convbn_grads: d weight (parameter layout), d gamma, d beta from the raw sums of the backward GEMM (bufs[0]);
        deferred: the stage's table launch at the end of the backward fills dW / dgb instead"""
        N, K = rec['W'].shape
        dwext, dW, dgb, deferred = bufs
        if not deferred:
            _lib.check(self.lib.effdet_train_convbn_grads(self.ops.st(), dwext.data_ptr(), N, K, int(transposed), rec['W'].data_ptr(),
                                                          rec['scale'].data_ptr(), rec['rstd'].data_ptr(), rec['mean'].data_ptr(),
                                                          dW.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr()), 'effdet_train_convbn_grads')
        wn, gn, bn_ = rec['names']
        grads[wn] = dW.view(rec['wshape'])
        grads[gn] = dgb[0]
        grads[bn_] = dgb[1]

    def _pw_bneval_fwd(self, x, conv, bn, names, silu_out=False, gate=None, resid=None, table=True):
        """1x1 conv (no bias) + BN with running statistics, folded: z = x (scale*W)^T + shift (and a = silu(z) when asked).
        gate [B, K]: the SE gate, applied to x while the GEMM loads it; resid: the block's shortcut, added in the epilogue."""
        if bn.training:
            raise NotImplementedError('backbone BatchNorm in batch-statistics mode is not built.  For training put the backbone BN in '
                                      'eval mode as pretrain.py:168-176 does (model.backbone.apply(set_bn_eval)); for inference call '
                                      'model.eval() or run under torch.no_grad() (a module in training mode with grad enabled takes '
                                      'the differentiable path)')
        N = conv.weight.shape[0]
        W = conv.weight.detach().reshape(N, -1)
        Wf, WfT, _, scale, shift, rstd = self._fold(W, bn, True, True, False, key=names[0] if table else None)
        B, H, Wd, K = x.shape
        rec = dict(x=x, W=W, Wf=Wf, WfT=WfT, mean=bn.running_mean, rstd=rstd, scale=scale, names=names, wshape=conv.weight.shape,
                   gate=gate, hw=H * Wd, table=table)
        out = self.ops.gemm_nt_fused(x, Wf, shift, a_scale=gate, a_rows=H * Wd, R=resid, silu_out=silu_out)
        if silu_out:
            return (out[0].view(B, H, Wd, N), out[1].view(B, H, Wd, N)), rec
        return out.view(B, H, Wd, N), rec

    def _pw_bneval_bwd(self, rec, dz, grads, need_dx=True, resid=None):
        """-> d x (+ resid: the gradient that reaches x through the shortcut); for a gated conv d (x * gate)"""
        N, K = rec['Wf'].shape
        bufs = self._grad_bufs(rec, False)
        if rec['gate'] is not None:
            self.ops.gemm_tn_scaled(dz, rec['x'], rec['gate'], rec['hw'], N, K, out=bufs[0])
        else:
            self.ops.gemm_tn(dz, rec['x'], N, K, out=bufs[0])
        self._convbn_grads(rec, bufs, False, grads)
        if not need_dx:
            return None
        B, H, Wd, _ = dz.shape
        return self.ops.gemm_nt_fused(dz, rec['WfT'], R=resid).view(B, H, Wd, K)

    def _dw_bneval_fwd(self, x, conv, bn, k, s, names):
        """-> z (pre-activation), silu(z), SE pool partial rows, rows per image, record"""
        if bn.training:
            raise NotImplementedError('backbone BatchNorm in batch-statistics mode is not built (see pretrain.py:168-176)')
        C = conv.weight.shape[0]
        W = conv.weight.detach().reshape(C, k * k)
        _, taps_s, taps, scale, shift, rstd = self._fold(W, bn, False, True, True, key=names[0])
        z, a, part, nblk = self.ops.dw_fwd_train(x, taps, scale, shift, k, s)
        return z, a, part, nblk, dict(x=x, W=W, taps_s=taps_s, mean=bn.running_mean, rstd=rstd, scale=scale, k=k, s=s, names=names,
                                      wshape=conv.weight.shape)

    def _dw_bneval_bwd(self, rec, dz, grads, z_below=None):
        """z_below: pre-activation of the layer that produced this conv's input: d input is multiplied by silu'(z_below)"""
        k, s = rec['k'], rec['s']
        bufs = self._grad_bufs(rec, True)
        dx, _, _ = self.ops.dw_bwd(dz, rec['x'], rec['taps_s'], k, s, z=z_below, out=bufs[0])
        self._convbn_grads(rec, bufs, True, grads)
        return dx

    def _bn_fwd(self, c, bn, prefix, silu_out=False):
        """BatchNorm2d on a raw conv output c [..., C] following bn.training (batch vs running statistics);
        silu_out: returns ((y, silu(y)), record)."""
        C = c.shape[-1]
        M = c.numel() // C
        ops = self.ops
        vec = ops.new(3, C)
        mom = bn.momentum if bn.momentum is not None else 0.1
        if bn.training:
            # batch mean, then the centred second pass whose second stage also does the layer's bookkeeping
            mean = ops.col_reduce(0, c, alpha=1.0 / M)
            ws = ops.ws(self.lib.effdet_train_col_reduce_workspace_floats(1, M, C))
            _lib.check(self.lib.effdet_train_bn_var_finalize(ops.st(), c.data_ptr(), mean.data_ptr(), M, C, bn.weight.data_ptr(),
                                                             bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                             bn.num_batches_tracked.data_ptr(), float(mom), float(M / max(M - 1, 1)),
                                                             float(bn.eps), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                                             ws.data_ptr(), ws.numel()), 'effdet_train_bn_var_finalize')
        else:
            mean, var = bn.running_mean, bn.running_var
            _lib.check(self.lib.effdet_train_bn_finalize(ops.st(), mean.data_ptr(), var.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                         bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                         bn.num_batches_tracked.data_ptr(), C, 0, float(mom),
                                                         float(M / max(M - 1, 1)), float(bn.eps), vec[0].data_ptr(), vec[1].data_ptr(),
                                                         vec[2].data_ptr()), 'effdet_train_bn_finalize')
        scale, shift, rstd = vec[0], vec[1], vec[2]
        y = ops.ew(3, c, v=(scale, shift, None, None), silu_out=silu_out)
        if not bn.training:
            mean = mean.clone()                 # the record must not alias a buffer a later load_state_dict may overwrite
        return y, dict(c=c, mean=mean, rstd=rstd, scale=scale, train=bn.training, M=M, prefix=prefix)

    def _bn_bwd(self, rec, dy, grads):
        c, mean, rstd = rec['c'], rec['mean'], rec['rstd']
        ops = self.ops
        C = c.shape[-1]
        # sum(dy), sum(dy * (c - mean)) in one pass over dy; its second stage turns them into d gamma, d beta, v1, v3
        out = ops.new(4, C)
        R = c.numel() // C
        ws = ops.ws(self.lib.effdet_train_col_reduce_workspace_floats(1, R, C))
        _lib.check(self.lib.effdet_train_bn_bwd_sums(ops.st(), dy.data_ptr(), c.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, C,
                                                     out.data_ptr(), ws.data_ptr(), ws.numel()), 'effdet_train_bn_bwd_sums')
        self._acc(grads, rec['prefix'] + 'weight', out[0])
        self._acc(grads, rec['prefix'] + 'bias', out[1])
        if rec['train']:
            return ops.ew(6, dy, c, v=(rec['scale'], out[2], mean, out[3]))
        return ops.ew(3, dy, v=(rec['scale'], None, None, None))

    @staticmethod
    def _acc(grads, name, g):
        grads[name] = g if name not in grads else grads[name] + g

    def _pw_fwd(self, x, conv, prefix, c_map=None):
        """1x1 conv with optional bias, raw output (BN, if any, is applied by _bn_fwd)."""
        N = conv.weight.shape[0]
        W = conv.weight.detach().reshape(N, -1).contiguous()
        bias = None if conv.bias is None else conv.bias.detach().contiguous()
        B, H, Wd, K = x.shape
        if c_map is None:
            c = self.ops.gemm_nt(x, W, bias).view(B, H, Wd, N)
        else:
            self.ops.gemm_nt(x, W, bias, M=B * H * Wd, a_map=(x.data_ptr(), 0, 0, 0), c_map=c_map)
            c = None
        Wt = self._transposed(prefix + 'weight^T', W) if self._stage is not None else None
        return c, dict(x=x, W=W, Wt=Wt, prefix=prefix, wshape=conv.weight.shape, has_bias=bias is not None)

    def _pw_bwd(self, rec, dc, grads, y_map=None, need_dx=True):
        N, K = rec['W'].shape
        B, H, Wd, _ = rec['x'].shape
        M = B * H * Wd
        dW, dsum = self.ops.gemm_tn(dc, rec['x'], N, K, M=M, y_map=y_map)
        self._acc(grads, rec['prefix'] + 'weight', dW.reshape(rec['wshape']))
        if rec['has_bias']:
            self._acc(grads, rec['prefix'] + 'bias', dsum)
        if not need_dx:
            return None
        Wt = rec['Wt'] if rec.get('Wt') is not None else rec['W'].t().contiguous()
        if y_map is None:
            return self.ops.gemm_nt(dc, Wt).view(B, H, Wd, K)
        return self.ops.gemm_nt(None, Wt, M=M, a_map=y_map).view(B, H, Wd, K)

    def _dw_fwd(self, x, conv, prefix):
        """depthwise 3x3/s1 without BN (SeparableConv2d.conv_dw, efficientdet.py:66-69)"""
        C = conv.weight.shape[0]
        k = conv.weight.shape[-1]
        taps = self._transposed(prefix + 'weight^T' if self._stage is not None else None, conv.weight.detach().reshape(C, k * k))
        d = self.ops.dw_fwd(x, taps, self._const(C, 1.0), self._const(C, 0.0), k, 1)
        return d, dict(x=x, taps=taps, k=k, prefix=prefix, wshape=conv.weight.shape)

    def _dw_bwd(self, rec, dd, grads):
        k = rec['k']
        dx, dtaps, _ = self.ops.dw_bwd(dd, rec['x'], rec['taps'], k, 1, cmajor=True)
        self._acc(grads, rec['prefix'] + 'weight', dtaps.reshape(rec['wshape']))
        return dx

    # =============================================================================================
    # backbone
    # =============================================================================================
    def _se_fwd(self, a, part, nblk, se, R, prefix):
        """SqueezeExcite gate [B, C] from the depthwise kernel's pool partial rows (the multiply happens inside the project GEMM)"""
        B, H, W, C = a.shape
        W1 = se.conv_reduce.weight.detach().reshape(R, C).contiguous()
        b1 = se.conv_reduce.bias.detach().contiguous()
        W2t = self._transposed(prefix + 'conv_expand.weight^T', se.conv_expand.weight.detach().reshape(C, R))
        b2 = se.conv_expand.bias.detach().contiguous()
        gate, pool = self.ops.new(B, C), self.ops.new(B, C)
        _lib.check(self.lib.effdet_train_se_gate(self.ops.st(), part.data_ptr(), nblk, H * W, W1.data_ptr(), b1.data_ptr(),
                                                 W2t.data_ptr(), b2.data_ptr(), gate.data_ptr(), pool.data_ptr(), B, C, R),
                   'effdet_train_se_gate')
        return gate, dict(a=a, pool=pool, gate=gate, W1=W1, b1=b1, W2t=W2t, R=R)

    def _se_bwd(self, rec, dag, z, grads, prefix):
        """dag = d (a * gate) -> d z, z the pre-activation of a = silu(z)  (gate, pool and SiLU backward in one element-wise pass)"""
        a, gate, R = rec['a'], rec['gate'], rec['R']
        B, H, W, C = a.shape
        dgate = self.ops.col_reduce(1, dag, a, per_image=True)
        ds = self.ops.new(B, C)
        L = 2 * R * C + R + C
        pg = self.ops.new(B, L)
        _lib.check(self.lib.effdet_train_se_bwd(self.ops.st(), rec['pool'].data_ptr(), H * W, gate.data_ptr(), dgate.data_ptr(),
                                                rec['W1'].data_ptr(), rec['b1'].data_ptr(), rec['W2t'].data_ptr(), ds.data_ptr(),
                                                pg.data_ptr(), B, C, R), 'effdet_train_se_bwd')
        g = self.ops.reduce_rows(pg)
        grads[prefix + 'conv_reduce.weight'] = g[:R * C].reshape(R, C, 1, 1)
        grads[prefix + 'conv_reduce.bias'] = g[R * C:R * C + R]
        grads[prefix + 'conv_expand.weight'] = g[R * C + R:2 * R * C + R].reshape(R, C).t().reshape(C, R, 1, 1)
        grads[prefix + 'conv_expand.bias'] = g[2 * R * C + R:]
        return self.ops.ew(12, dag, c=z, v=(gate, ds, None, None), s=(1.0 / (H * W), 0.0, 0.0, 0.0), hw=H * W)

    # The stage functions proper.  Inside them `self._stage` names the stage, and the parameter-sized work (derived weights,
    # conv + BN parameter gradients) goes through that stage's tables: one launch per kind (see _StageTables).
    def _in_stage(self, stage, fn, *args, **kw):
        self._stage = stage if self.use_tables else None
        self.ops.pad = self.bb_pad if stage == 'bb' else self.fpn_pad
        try:
            return fn(*args, **kw)
        finally:
            self._stage = None
            self.ops.pad = 0

    def bb_forward(self, x):
        """x: [B,3,H,W] float32 (normalised) or uint8 (raw; loader normalisation applied).  -> (feats NHWC list, saved)"""
        def run():
            if self.use_tables:
                self._tables['bb'].run_prep()
            return self._bb_forward(x)
        return self._in_stage('bb', run)

    def bb_backward(self, dfeats, saved):
        """dfeats: d loss / d feature maps (NHWC, None allowed) -> {param name (relative to backbone): grad}"""
        def run():
            grads = self._bb_backward(dfeats, saved)
            if self.use_tables:
                self._tables['bb'].run_grads()
            stem_c = self.model.backbone.arch[0]
            g = grads['conv_stem.weight']                          # [C0, 32] patch layout -> [C0, 3, 3, 3]
            grads['conv_stem.weight'] = g.reshape(stem_c, 32)[:, :27].reshape(stem_c, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
            return grads
        return self._in_stage('bb', run)

    def fh_forward(self, feats, want_cls=True, want_box=True):
        """feats: backbone feature maps (NHWC).  -> (cls_all [B,N,C], box_all [B,N,4], saved)"""
        def run():
            if self.use_tables:
                self._tables['fh'].run_prep()
            return self._fh_forward(feats, want_cls, want_box)
        return self._in_stage('fh', run)

    def fh_backward(self, g_cls, g_box, saved, need_dfeats=True):
        """-> (d feats list (NHWC), {param name: grad})"""
        return self._in_stage('fh', self._fh_backward, g_cls, g_box, saved, need_dfeats)

    def _bb_forward(self, x):
        """x: [B,3,H,W] float32 (normalised) or uint8 (raw; loader normalisation applied).  -> (feats NHWC list, saved)"""
        bb = self.model.backbone
        ops = self.ops
        if x.device != self.dev:
            raise RuntimeError('input must live on %s (no CPU fallback)' % (self.dev,))
        B, _, H, W = x.shape
        if x.dtype == torch.uint8:
            import ctypes
            mean = (ctypes.c_float * 3)(*[255.0 * v for v in self.model.input_mean])
            std = (ctypes.c_float * 3)(*[255.0 * v for v in self.model.input_std])
            xf = ops.new(B, 3, H, W)
            _lib.check(self.lib.effdet_normalize_u8(ops.st(), 0, x.contiguous().data_ptr(), mean, std, xf.data_ptr(), B, 3, H * W),
                       'effdet_normalize_u8')
            x = xf
        elif x.dtype != torch.float32:
            raise RuntimeError('training input must be float32 or uint8')
        x = x.contiguous()
        stem_c, stages = bb.arch
        Ho, Wo = _same_out(H, 2), _same_out(W, 2)
        col = ops.new(B, Ho, Wo, 32)
        _lib.check(self.lib.effdet_train_im2col_stem(ops.st(), x.data_ptr(), col.data_ptr(), B | ops.pad, H, W), 'effdet_train_im2col_stem')
        saved = dict(blocks=[])
        # stochastic depth (timm drop_path: x / keep_prob * floor(keep_prob + U[0,1)) per sample, then + shortcut), active while the
        # backbone MODULE is in training mode - pretrain.py:168-176 only puts its BatchNorm layers in eval mode
        drop_rates = bb.block_drop_rates() if (bb.training and bb.drop_path_rate > 0.0) else None
        fixed_masks = getattr(bb, 'drop_path_masks', None)
        flat_idx = 0
        drop_scale = None
        if drop_rates is not None and fixed_masks is None:
            # every residual block's per-image scale floor(keep + U) / keep from ONE draw (a handful of launches, not five per block)
            key = tuple(float(v) for v in drop_rates)
            if getattr(self, '_keeps', (None,))[0] != key:
                self._keeps = (key, torch.tensor([1.0 - v for v in key], dtype=torch.float32, device=self.dev).reshape(-1, 1))
            keeps = self._keeps[1]
            drop_scale = torch.floor(keeps + torch.rand(len(key), B, device=self.dev, dtype=torch.float32)) / keeps

        class _StemConv(object):                      # conv_stem as a 1x1 conv over the 32-wide patches
            pass
        sc = _StemConv()
        w = bb.conv_stem.weight.detach().permute(0, 2, 3, 1).reshape(stem_c, 27)
        sc.weight = torch.cat([w, w.new_zeros(stem_c, 5)], 1)
        (z0, cur), rec = self._pw_bneval_fwd(col, sc, bb.bn1, ('conv_stem.weight', 'bn1.weight', 'bn1.bias'), silu_out=True,
                                             table=False)       # its padded weight is a fresh tensor every step
        rec['wshape'] = (stem_c, 32)
        saved['stem'] = (rec, z0)
        feats = []
        for si, blocks in enumerate(stages):
            for bi, b in enumerate(blocks):
                m = bb.blocks[si][bi]
                p = 'blocks.%d.%d.' % (si, bi)
                r = dict(b=b, p=p, x=cur)
                if b['type'] == 'ir':
                    (z1, a1), r['pw'] = self._pw_bneval_fwd(cur, m.conv_pw, m.bn1, (p + 'conv_pw.weight', p + 'bn1.weight', p + 'bn1.bias'),
                                                            silu_out=True)
                    r['z1'] = z1
                    z2, a2, part, nblk, r['dw'] = self._dw_bneval_fwd(a1, m.conv_dw, m.bn2, b['k'], b['s'],
                                                                      (p + 'conv_dw.weight', p + 'bn2.weight', p + 'bn2.bias'))
                    proj, bnp, pn = m.conv_pwl, m.bn3, (p + 'conv_pwl.weight', p + 'bn3.weight', p + 'bn3.bias')
                else:
                    z2, a2, part, nblk, r['dw'] = self._dw_bneval_fwd(cur, m.conv_dw, m.bn1, b['k'], b['s'],
                                                                      (p + 'conv_dw.weight', p + 'bn1.weight', p + 'bn1.bias'))
                    proj, bnp, pn = m.conv_pw, m.bn2, (p + 'conv_pw.weight', p + 'bn2.weight', p + 'bn2.bias')
                r['z2'] = z2
                gate, r['se'] = self._se_fwd(a2, part, nblk, m.se, b['se'], p + 'se.')
                r['drop'] = None
                dropped = b['residual'] and drop_rates is not None and drop_rates[flat_idx] > 0.0
                # project conv: the SE gate multiplies its input while the GEMM loads it, the shortcut is added in its epilogue
                z3, r['proj'] = self._pw_bneval_fwd(a2, proj, bnp, pn, gate=gate, resid=cur if (b['residual'] and not dropped) else None)
                if dropped:
                    keep = 1.0 - drop_rates[flat_idx]
                    if drop_scale is not None:
                        scale = drop_scale[flat_idx]
                    elif flat_idx in fixed_masks:
                        scale = fixed_masks[flat_idx].to(device=self.dev, dtype=torch.float32).reshape(B) / keep
                    else:
                        scale = torch.floor(keep + torch.rand(B, device=self.dev, dtype=torch.float32)) / keep
                    # per-image scale as a [B, C] table for the element-wise kernels (a * v0[img, c])
                    r['drop'] = scale.reshape(B, 1).expand(B, b['cout']).contiguous()
                    z3 = ops.ew(13, z3, cur, v=(r['drop'], None, None, None), hw=z3.shape[1] * z3.shape[2])
                cur = z3
                flat_idx += 1
                saved['blocks'].append(r)
            if si in (2, 4, 6):
                feats.append(cur)
        return feats, saved

    def _bb_backward(self, dfeats, saved):
        """dfeats: d loss / d feature maps (NHWC, None allowed) -> {param name (relative to backbone): grad}"""
        ops = self.ops
        grads = {}
        stem_c, stages = self.model.backbone.arch
        flat = [(si, bi) for si, blocks in enumerate(stages) for bi in range(len(blocks))]
        feat_of_stage = {2: 0, 4: 1, 6: 2}
        dcur = None
        for idx in range(len(flat) - 1, -1, -1):
            si, bi = flat[idx]
            r = saved['blocks'][idx]
            b = r['b']
            if bi == len(stages[si]) - 1 and si in feat_of_stage:
                df = dfeats[feat_of_stage[si]]
                if df is not None:
                    dcur = df if dcur is None else ops.add(dcur, df)
            if dcur is None:
                raise RuntimeError('no gradient reached the last backbone stage')
            dz3 = dcur if r.get('drop') is None else ops.ew(4, dcur, v=(r['drop'], None, None, None), hw=dcur.shape[1] * dcur.shape[2])
            dag = self._pw_bneval_bwd(r['proj'], dz3, grads)
            dz2 = self._se_bwd(r['se'], dag, r['z2'], grads, r['p'] + 'se.')
            shortcut = dcur if b['residual'] else None
            if b['type'] == 'ir':
                dz1 = self._dw_bneval_bwd(r['dw'], dz2, grads, z_below=r['z1'])
                dcur = self._pw_bneval_bwd(r['pw'], dz1, grads, resid=shortcut)
            else:
                dx = self._dw_bneval_bwd(r['dw'], dz2, grads)
                dcur = ops.add(dx, shortcut) if shortcut is not None else dx
        rec, z0 = saved['stem']
        dz0 = ops.silu_bwd(z0, dcur)
        self._pw_bneval_bwd(rec, dz0, grads, need_dx=False)
        return grads

    # =============================================================================================
    # BiFPN + heads
    # =============================================================================================
    def _convbn_fwd(self, x, cba, prefix):
        """ConvBnAct2d without activation (efficientdet.py:42-57): 1x1 conv (+bias) -> BN"""
        c, rpw = self._pw_fwd(x, cba.conv, prefix + 'conv.')
        if cba.bn is None:
            return c, dict(pw=rpw, bn=None)
        y, rbn = self._bn_fwd(c, cba.bn, prefix + 'bn.')
        return y, dict(pw=rpw, bn=rbn)

    def _convbn_bwd(self, rec, dy, grads):
        dc = dy if rec['bn'] is None else self._bn_bwd(rec['bn'], dy, grads)
        return self._pw_bwd(rec['pw'], dc, grads)

    def _resample_fwd(self, x, rs, prefix, delta):
        """ResampleFeatureMap (efficientdet.py:140-177): conv(+BN) when channels differ, then max-pool (delta = -1: the input
        is one level finer) or nearest x2 upsample (delta = +1)."""
        rec = dict(conv=None, delta=delta)
        if hasattr(rs, 'conv'):
            x, rec['conv'] = self._convbn_fwd(x, rs.conv, prefix + 'conv.')
        if delta == -1:
            rec['pool_in'] = x
            x = self.ops.maxpool(x)
        elif delta == 1:
            x = self.ops.spatial(0, x)
        elif delta != 0:
            raise NotImplementedError('BiFPN edge spanning %d levels' % delta)
        return x, rec

    def _resample_bwd(self, rec, dy, grads):
        if rec['delta'] == -1:
            dy = self.ops.spatial(2, rec['pool_in'], dy)
        elif rec['delta'] == 1:
            dy = self.ops.spatial(1, dy)
        if rec['conv'] is not None:
            dy = self._convbn_bwd(rec['conv'], dy, grads)
        return dy

    def _fh_forward(self, feats, want_cls=True, want_box=True):
        """feats: backbone feature maps (NHWC).  -> (cls_all [B,N,C], box_all [B,N,4], saved)"""
        model, ops, F, L = self.model, self.ops, self.F, self.L
        fpn = model.fpn
        info = fpn.in_feature_info
        nbb = len(info)
        red0 = info[0]['reduction']
        saved = dict(extra=[], nodes=[], nbb=nbb)
        x = [dict(t=f, level=i, src=('feat', i)) for i, f in enumerate(feats)]
        chs = [i['num_chs'] for i in info]
        for level in range(nbb, L):
            y, rec = self._resample_fwd(x[-1]['t'], fpn.resample[str(level)], 'fpn.resample.%d.' % level, -1)
            saved['extra'].append((rec, len(x) - 1))
            x.append(dict(t=y, level=level))
            chs.append(F)
        nodes = fpn.fpn_config.nodes
        lvl_hw = [(t['t'].shape[1], t['t'].shape[2]) for t in x]              # level -> (H, W)
        # x grows by 8 nodes per cell and is cut back to the last L; `ids` tracks global tensor ids for the backward pass
        tensors = list(x)
        ids = list(range(len(x)))
        for ci in range(len(fpn.cell)):
            layer = fpn.cell[ci]
            for ni, node in enumerate(nodes):
                fn = layer.fnode[ni]
                lvl = int(round(math.log2(node['reduction'] / red0)))
                p = 'fpn.cell.%d.fnode.%d.' % (ci, ni)
                ins, recs, src_ids = [], [], []
                for off in node['inputs_offsets']:
                    src = tensors[ids[off]]
                    rs = fn.combine.resample[str(off)]
                    if abs(src['level'] - lvl) > 1:
                        raise NotImplementedError('BiFPN edge spanning %d levels' % (src['level'] - lvl))
                    t_in, rec = src['t'], dict(conv=None)
                    if hasattr(rs, 'conv'):                      # channel-matching 1x1 conv + BN first (conv_after_downsample=False)
                        t_in, rec['conv'] = self._convbn_fwd(t_in, rs.conv, '%scombine.resample.%d.conv.' % (p, off))
                    ins.append(t_in)                             # max-pool / nearest x2 happen inside the combine kernel
                    recs.append(rec)
                    src_ids.append(ids[off])
                method = node['weight_method']
                if method not in _FPN_METHODS:
                    raise ValueError('unknown weight_method %r' % (method,))
                mid = _FPN_METHODS[method]
                ewp = fn.combine.edge_weights.detach() if mid < 2 else None
                # fusion weights stay on the device (a float[4] = w0, w1, w2, den read by the kernels): no host read-back
                if mid < 2 and self._stage is not None:
                    wdev = self._tables[self._stage].edge_weights(p + 'combine.edge_weights', ewp, len(ins), mid,
                                                                  lambda: ops.fpn_weights(ewp, len(ins), mid))
                else:
                    wdev = ops.fpn_weights(ewp, len(ins), mid)
                H_, W_ = lvl_hw[lvl]
                fused, act = ops.fpn_combine(ins, wdev, mid, H_, W_)
                sc = fn.after_combine.conv
                d, rdw = self._dw_fwd(act, sc.conv_dw, p + 'after_combine.conv.conv_dw.')
                c, rpw = self._pw_fwd(d, sc.conv_pw, p + 'after_combine.conv.conv_pw.')
                y, rbn = self._bn_fwd(c, sc.bn, p + 'after_combine.conv.bn.')
                tensors.append(dict(t=y, level=lvl))
                ids.append(len(tensors) - 1)
                saved['nodes'].append(dict(p=p, ins=ins, recs=recs, src_ids=src_ids, method=mid, wdev=wdev, ewp=ewp, fused=fused,
                                           dw=rdw, pw=rpw, bn=rbn, out_id=len(tensors) - 1, n_in=len(ins)))
            ids = ids[-L:]
        pyr = [tensors[i] for i in ids]
        saved['pyr_ids'] = list(ids)
        saved['n_tensors'] = len(tensors)
        saved['levels'] = [(t['t'].shape[1], t['t'].shape[2]) for t in pyr]
        # ---- heads
        B = feats[0].shape[0]
        hw = saved['levels']
        P = sum(h * w for h, w in hw)
        offs, o = [], 0
        for h, w in hw:
            offs.append(o)
            o += h * w
        A = self.A
        outs = []
        saved['heads'] = []
        for head, name, K, want in ((model.class_net, 'class_net.', self.cfg.num_classes, want_cls), (model.box_net, 'box_net.', 4, want_box)):
            if not want:
                outs.append(None)
                saved['heads'].append(None)
                continue
            if not hasattr(head, 'conv_rep') or not hasattr(head, 'bn_rep'):
                raise NotImplementedError('the training path needs HeadNet heads (MetaHead gradients are not built)')
            NO = A * K
            out_t = ops.new(B, A * P, K)
            hrec = dict(name=name, NO=NO, levels=[None] * L, out=out_t)
            outs.append(out_t)
            saved['heads'].append(hrec)
        # The towers share their conv weights over the levels (only BatchNorm is per level): every layer of a head is ONE launch
        # over the packed pyramid [B * P, F] (csrc/train_levels.hip) instead of one per level.
        lv = _Levels(B, hw)
        saved['lv'] = lv
        packed = torch.cat([t['t'].reshape(-1, F) for t in pyr], 0)
        for hi, head in enumerate((model.class_net, model.box_net)):
            if saved['heads'][hi] is not None:
                self._head_fwd(lv, packed, head, saved['heads'][hi])
        saved['P'] = P
        return outs[0], outs[1], saved

    def _dw_taps(self, conv, key):
        C, k = conv.weight.shape[0], conv.weight.shape[-1]
        if k != 3 or tuple(conv.stride) != (1, 1):
            raise NotImplementedError('head depthwise convs other than 3x3 / stride 1')
        return self._transposed(key, conv.weight.detach().reshape(C, k * k))

    def _head_fwd(self, lv, t, head, hrec):
        """HeadNet.forward (efficientdet.py:438-452) over all levels at once: per repeat dw3x3 -> 1x1 conv (+bias) -> per-level
        BatchNorm -> SiLU; then the predict SeparableConv2d, written straight into the image-major [B, A*P, K] head tensor."""
        ops, name, NO, out_t = self.ops, hrec['name'], hrec['NO'], hrec['out']
        L = lv.L
        hrec['reps'] = []
        for r in range(len(head.conv_rep)):
            conv = head.conv_rep[r]
            taps = self._dw_taps(conv.conv_dw, '%sconv_rep.%d.conv_dw.weight^T' % (name, r))
            d = ops.lv_dw(lv, t, taps)
            N = conv.conv_pw.weight.shape[0]
            W = conv.conv_pw.weight.detach().reshape(N, -1).contiguous()
            bias = None if conv.conv_pw.bias is None else conv.conv_pw.bias.detach().contiguous()
            c = ops.gemm_nt(d, W, bias)
            bns = [head.bn_rep[r][l].bn for l in range(L)]
            sums = sq = None
            if any(bn.training for bn in bns):
                sums = ops.lv_col_reduce(lv, 0, c)
                sq = ops.lv_col_reduce(lv, 2, c, v=sums, vscale=lv.inv_m)
            st = ops.lv_bn_finalize(lv, sums, sq, bns, N)                  # mean, scale, shift, rstd  [4, L, N]
            y, tn = ops.lv_ew(lv, 3, c, v=(st[1], st[2], None, None), silu_out=True)
            import ctypes
            hrec['reps'].append(dict(x=t, taps=taps, d=d, W=W, Wt=self._transposed('%sconv_rep.%d.conv_pw.weight^T' % (name, r), W),
                                     c=c, y=y, st=st, has_bias=bias is not None, r=r,
                                     train=(ctypes.c_int * L)(*[int(bn.training) for bn in bns]),
                                     wshape=conv.conv_pw.weight.shape, dwshape=conv.conv_dw.weight.shape))
            t = tn
        taps = self._dw_taps(head.predict.conv_dw, name + 'predict.conv_dw.weight^T')
        d = ops.lv_dw(lv, t, taps)
        Wp = head.predict.conv_pw.weight.detach().reshape(NO, -1).contiguous()
        bias = None if head.predict.conv_pw.bias is None else head.predict.conv_pw.bias.detach().contiguous()
        pk = (lv.P * NO, NO)
        ops.gemm_nt_levels(lv, d, Wp, bias, out_packed=out_t, pk=pk)
        hrec['predict'] = dict(x=t, taps=taps, d=d, W=Wp, Wt=self._transposed(name + 'predict.conv_pw.weight^T', Wp),
                               has_bias=bias is not None, pk=pk,
                               wshape=head.predict.conv_pw.weight.shape, dwshape=head.predict.conv_dw.weight.shape)

    def _head_bwd(self, lv, hrec, g, grads):
        """-> d packed pyramid [B * P, F]; parameter gradients into `grads` (conv weights: one reduction over all levels)"""
        ops, name, NO = self.ops, hrec['name'], hrec['NO']
        L = lv.L

        def dw_grads(rec, dd, prefix):
            grads[prefix + 'conv_dw.weight'] = ops.lv_dw_bwd_dw(lv, dd, rec['x'], cmajor=True).reshape(rec['dwshape'])
            return ops.lv_dw(lv, dd, rec['taps'], flip=True)

        rec = hrec['predict']
        F = rec['W'].shape[1]
        dW, dsum = ops.gemm_tn_levels(lv, g, rec['d'], NO, F, y_packed=True, pk=rec['pk'])
        grads[name + 'predict.conv_pw.weight'] = dW.reshape(rec['wshape'])
        if rec['has_bias']:
            grads[name + 'predict.conv_pw.bias'] = dsum
        dd = ops.gemm_nt_levels(lv, g, rec['Wt'], a_packed=True, pk=rec['pk'])
        da = dw_grads(rec, dd, name + 'predict.')
        for rec in reversed(hrec['reps']):
            r, st = rec['r'], rec['st']
            N = rec['W'].shape[0]
            # SiLU backward (pre = y) folded into both passes of the BatchNorm backward
            sums = ops.lv_col_reduce(lv, 4, da, b=rec['c'], v=st[0], pre=rec['y'])
            pb = ops.lv_bn_bwd_prep(lv, sums, st[3], N)                    # d gamma, d beta, v1, v3  [4, L, N]
            for l in range(L):
                grads['%sbn_rep.%d.%d.bn.weight' % (name, r, l)] = pb[0, l]
                grads['%sbn_rep.%d.%d.bn.bias' % (name, r, l)] = pb[1, l]
            dc = ops.lv_ew(lv, 6, da, b=rec['c'], pre=rec['y'], v=(st[1], pb[2], st[0], pb[3]), train=rec['train'])
            dW, dsum = ops.gemm_tn(dc, rec['d'], N, rec['W'].shape[1])
            grads['%sconv_rep.%d.conv_pw.weight' % (name, r)] = dW.reshape(rec['wshape'])
            if rec['has_bias']:
                grads['%sconv_rep.%d.conv_pw.bias' % (name, r)] = dsum
            dd = ops.gemm_nt(dc, rec['Wt'])
            da = dw_grads(rec, dd, '%sconv_rep.%d.' % (name, r))
        return da

    def _fh_backward(self, g_cls, g_box, saved, need_dfeats=True):
        """-> (d feats list (NHWC), {param name: grad})"""
        ops, L = self.ops, self.L
        grads = {}
        P = saved['P']
        hw = saved['levels']
        dt = [None] * saved['n_tensors']                      # gradient per global tensor id

        def add_to(i, g):
            dt[i] = g if dt[i] is None else ops.add(dt[i], g)

        gs = [None if g is None else g.contiguous() for g in (g_cls, g_box)]
        lv = saved['lv']
        dpyr = None
        for hi in range(2):
            if saved['heads'][hi] is None or gs[hi] is None:
                continue
            da = self._head_bwd(lv, saved['heads'][hi], gs[hi], grads)
            dpyr = da if dpyr is None else ops.add(dpyr, da)
        if dpyr is not None:
            for l, d in enumerate(lv.split(dpyr)):
                add_to(saved['pyr_ids'][l], d)
        for nrec in reversed(saved['nodes']):
            dy = dt[nrec['out_id']]
            if dy is None:
                continue
            p = nrec['p']
            dc = self._bn_bwd(nrec['bn'], dy, grads)
            dd = self._pw_bwd(nrec['pw'], dc, grads)
            dact = self._dw_bwd(nrec['dw'], dd, grads)
            # SiLU backward, the edge-weight gradient and every input's gradient read dact / fused directly (csrc/train_fpn.hip)
            fused, wdev, ins = nrec['fused'], nrec['wdev'], nrec['ins']
            if nrec['method'] < 2:
                grads[p + 'combine.edge_weights'] = ops.fpn_wgrad(ins, wdev, nrec['method'], nrec['ewp'], dact, fused)
            for i in range(nrec['n_in']):
                sid, rec = nrec['src_ids'][i], nrec['recs'][i]
                if rec['conv'] is None:
                    dt[sid] = ops.fpn_input_bwd(i, ins[i], wdev, dact, fused, acc=dt[sid])
                else:
                    add_to(sid, self._convbn_bwd(rec['conv'], ops.fpn_input_bwd(i, ins[i], wdev, dact, fused), grads))
        nbb = saved['nbb']
        for k in range(len(saved['extra']) - 1, -1, -1):
            rec, src = saved['extra'][k]
            dy = dt[nbb + k]
            if dy is None:
                continue
            add_to(src, self._resample_bwd(rec, dy, grads))
        return dt[:nbb], grads


# =================================================================================================
# autograd plumbing
# =================================================================================================
def _param_list(module, prefix):
    return [(prefix + n, p) for n, p in module.named_parameters()]


def _param_grads(ctx, grads, first):
    """Gradients of the stage's parameters in input order.  With `eng.direct_grad` (set by PretrainStep) a parameter whose
    `.grad` already exists gets its gradient added right there by ONE multi-tensor launch for the whole stage and autograd
    receives None for it - instead of one AccumulateGrad addition per parameter (460 tiny launches per step for d0)."""
    out, dst, src = [], [], []
    for i, (n, p) in enumerate(zip(ctx.names, ctx.params)):
        g = grads.get(n) if ctx.needs_input_grad[first + i] else None
        if g is None:
            out.append(None)
            continue
        g = g.contiguous()
        if ctx.eng.direct_grad and p.grad is not None and p.grad.is_contiguous() and p.grad.shape == g.shape:
            dst.append(p.grad)
            src.append(g)
            out.append(None)
        else:
            # `g` may be (a view of) a persistent buffer of the stage tables, which the next backward overwrites: autograd's
            # AccumulateGrad keeps the tensor it is handed as `.grad`, so it must get its own copy (two backwards without
            # zero_grad, or two 'bb' nodes in one graph as in infer.py:345-351, would otherwise add a buffer to itself)
            out.append(g.clone())
    if dst:
        torch._foreach_add_(dst, src)
    return out


class BackboneFn(torch.autograd.Function):
    """x -> backbone features (NHWC tensors); gradients for the backbone parameters."""

    @staticmethod
    def forward(ctx, eng, names, x, *params):
        feats, saved = eng.bb_forward(x)
        ctx.eng, ctx.saved, ctx.names, ctx.params = eng, saved, names, params
        return tuple(feats)

    @staticmethod
    def backward(ctx, *dfeats):
        dfeats = [None if d is None else d.contiguous() for d in dfeats]
        grads = ctx.eng.bb_backward(dfeats, ctx.saved)
        ctx.saved = None
        return (None, None, None) + tuple(_param_grads(ctx, grads, 3))


class FpnHeadFn(torch.autograd.Function):
    """backbone features (NHWC) -> packed head outputs cls [B,N,C], box [B,N,4]; gradients for BiFPN / head parameters."""

    @staticmethod
    def forward(ctx, eng, names, n_feats, *tensors):
        feats = [t.contiguous() for t in tensors[:n_feats]]
        cls_all, box_all, saved = eng.fh_forward(feats)
        ctx.eng, ctx.saved, ctx.names, ctx.n_feats, ctx.params = eng, saved, names, n_feats, tensors[n_feats:]
        return cls_all, box_all

    @staticmethod
    def backward(ctx, g_cls, g_box):
        dfeats, grads = ctx.eng.fh_backward(g_cls, g_box, ctx.saved)
        ctx.saved = None
        out = [d if ctx.needs_input_grad[3 + i] else None for i, d in enumerate(dfeats)]
        return (None, None, None) + tuple(out) + tuple(_param_grads(ctx, grads, 3 + ctx.n_feats))


def run_backbone(eng, x):
    """-> NCHW-shaped views of the NHWC feature maps, attached to the autograd graph"""
    pl = _param_list(eng.model.backbone, '')
    names = [n for n, _ in pl]
    feats = BackboneFn.apply(eng, names, x, *[p for _, p in pl])
    return [f.permute(0, 3, 1, 2) for f in feats]


def run_fpn_heads(eng, feats_nchw):
    model = eng.model
    pl = _param_list(model.fpn, 'fpn.') + _param_list(model.class_net, 'class_net.') + _param_list(model.box_net, 'box_net.')
    names = [n for n, _ in pl]
    feats = [f.permute(0, 2, 3, 1) for f in feats_nchw]
    cls_all, box_all = FpnHeadFn.apply(eng, names, len(feats), *(feats + [p for _, p in pl]))
    hw = [(f.shape[2], f.shape[3]) for f in feats_nchw]
    while len(hw) < eng.L:
        hw.append((_same_out(hw[-1][0], 2), _same_out(hw[-1][1], 2)))
    A = eng.A

    def views(t, K):
        out, off = [], 0
        B = t.shape[0]
        for h, w in hw:
            v = t[:, off * A:(off + h * w) * A, :].reshape(B, h, w, A * K)
            out.append(v.permute(0, 3, 1, 2))
            off += h * w
        return out

    return views(cls_all, model.config.num_classes), views(box_all, 4)
