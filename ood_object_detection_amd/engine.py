"""Launch plan for the EfficientDet forward pass on libeffdet_hip.so.

`Engine(model, B, image_size)` reads the model's parameters once, folds every BatchNorm into a
per-channel (scale, shift) pair, repacks conv weights into the kernel layouts and records, for the
three stages backbone / BiFPN / heads, a flat list of C-ABI calls on preallocated NHWC buffers.
Running a stage replays that list on torch's current HIP stream (so a stage can be captured into a
hipGraph with torch.cuda.graph).  PyTorch is used for device memory and streams only.

Data layout in HBM
  activations        NHWC, dtype = the model's parameter dtype (float32 parity / bfloat16 throughput); a float32 model with
                     `model.compute_mode = 'accurate'` runs dtype 2 of the C ABI instead: two-term bf16 values (pairfmt.py: 16
                     significand bits, three matrix-core products per multiply) in float32-sized storage, float32 logits / boxes
  pyramid features   one packed tensor [B, P, F] (P = sum_l H_l*W_l), level l at pixel offset off_l
  class / box heads  [B, N, C] and [B, N, 4] with N = 9*P: exactly the concatenated layout that
                     `_post_process` builds with permute/reshape/cat (effdet/bench.py:36-42)
  OOD scores         energy, max_logit: [B, N] float32
"""
import ctypes
import math

import torch

from . import _lib
from . import pairfmt

_DT = {torch.float32: 0, torch.bfloat16: 1}
_PAIR = 2                    # dtype code of the two-term bf16 ("accurate") mode
_PAD_SYMMETRIC = 1 << 24     # EFFDET_PAD_SYMMETRIC: OR-ed into the dtype argument of the entry points that pad


def _same_out(n, s):
    return (n + s - 1) // s


def _arr(ctype, values):
    return (ctype * len(values))(*values)


class Engine(object):
    def __init__(self, model, B, image_size, ood_out=None):
        cfg = model.config
        self._ood_out = ood_out
        self.wtoken = model.weights_token()                  # fingerprint of the parameters the packed copies are made from
        p0 = model.backbone.conv_stem.weight
        if p0.device.type != 'cuda':
            raise RuntimeError('the EfficientDet HIP path needs the model on a GPU (cuda:N); there is no CPU fallback')
        if p0.dtype not in _DT:
            raise RuntimeError('supported parameter dtypes: float32, bfloat16 (got %s)' % p0.dtype)
        self.lib = _lib.load()
        self.device, self.dtype, self.dt = p0.device, p0.dtype, _DT[p0.dtype]
        self.mode = getattr(model, 'compute_mode', 'native') or 'native'
        if self.mode not in ('native', 'accurate'):
            raise ValueError("compute_mode must be 'native' or 'accurate' (got %r)" % (self.mode,))
        self.pair = self.mode == 'accurate'
        # padding convention (timm `padding=`): TF-"SAME" for the tf_ family, static symmetric for pad_type = '' (efficientdet_d0 / d1
        # on efficientnet_b0 / b1, the scripts' default models); the backbone follows its own name, BiFPN / heads config.pad_type
        self.bb_pad = _PAD_SYMMETRIC if getattr(model.backbone, 'pad_type', 'same') == '' else 0
        self.fpn_pad = _PAD_SYMMETRIC if cfg.pad_type == '' else 0
        if self.pair:
            if p0.dtype != torch.float32:
                raise RuntimeError("compute_mode='accurate' splits float32 master weights into two bf16 terms: it needs a float32 model")
            self.dt = _PAIR
        self.B, self.image_size = B, tuple(image_size)
        self.cfg = cfg
        self.F = cfg.fpn_channels
        self.C = cfg.num_classes
        self.A = model.num_anchors
        self.L = cfg.num_levels
        self._keep = []          # tensors / ctypes arrays referenced by the launch lists
        self.input_mean = tuple(getattr(model, 'input_mean', (0.485, 0.456, 0.406)))     # used for uint8 inputs only
        self.input_std = tuple(getattr(model, 'input_std', (0.229, 0.224, 0.225)))
        self.pyr_es = torch.empty(0, dtype=self.dtype).element_size()
        H, W = self.image_size
        if cfg.fpn_channels > 288:
            raise NotImplementedError('BiFPN / head width %d: the fused separable-conv kernel keeps a tile of all channels in LDS and is '
                                      'built for widths up to 288 (tf_efficientdet_d0 ... d5); d6 / d7 are not built' % cfg.fpn_channels)
        if H % (2 ** cfg.max_level) or W % (2 ** cfg.max_level):
            raise ValueError('image size must be divisible by 2**max_level (reference: effdet/anchors.py:229-230)')
        with torch.no_grad():
            self._build_backbone(model.backbone, H, W)
            self._build_fpn(model.fpn)
            self._build_heads(model)

    def matches(self, model):
        p0 = model.backbone.conv_stem.weight
        return p0.device == self.device and p0.dtype == self.dtype and model.config.num_classes == self.C and \
            (getattr(model, 'compute_mode', 'native') or 'native') == self.mode

    # ------------------------------------------------------------------------------------ utils
    def _new(self, *shape, dtype=None):
        t = torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)
        self._keep.append(t)
        return t

    def _f32(self, t):
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._keep.append(t)
        return t

    def _w(self, t):
        """weight matrix [N, K] of an MFMA GEMM in the engine's dtype (accurate mode: the two-term layout along K)"""
        if self.pair:
            t = pairfmt.encode(t.detach().to(device=self.device, dtype=torch.float32))
        else:
            t = t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        self._keep.append(t)
        return t

    def _act_in(self, t):
        """values -> the engine's activation storage (accurate mode: encode); t is NHWC-shaped"""
        return pairfmt.encode(t.float()) if self.pair else t

    def _act_out(self, t):
        """the engine's activation storage -> values (accurate mode: a decoded copy)"""
        return pairfmt.decode(t) if self.pair else t

    def _fold(self, bn, conv_bias=None):
        """BatchNorm (eval) -> scale, shift with an optional preceding conv bias folded in."""
        w, b = bn.weight.detach().float(), bn.bias.detach().float()
        m, v = bn.running_mean.detach().float(), bn.running_var.detach().float()
        scale = w / torch.sqrt(v + bn.eps)
        shift = b - m * scale
        if conv_bias is not None:
            shift = shift + conv_bias.detach().float() * scale
        return scale, shift

    @staticmethod
    def _dw_taps(w):          # [C,1,k,k] -> [k*k, C]
        k = w.shape[-1]
        return w.detach().float().permute(2, 3, 0, 1).reshape(k * k, w.shape[0])

    def _run(self, plan):
        st = torch.cuda.current_stream(self.device).cuda_stream
        for fn, args, what, _meta in plan:
            rc = fn(st, *args)
            if rc != 0:
                _lib.check(rc, what)

    def _gemm_meta(self, M, K, N, residual=False, gate=False):
        es = self.pyr_es
        b = (M * K + M * N + N * K) * es + (M * N * es if residual else 0) + (self.B * K * 4 if gate else 0)
        return dict(kind='pw_gemm', bytes=b, flops=2 * M * K * N)

    def profile(self, reps=5):
        """Per-launch device time of every recorded launch (HIP events on the launch stream).
        Returns [(what, kind, algorithmic_bytes, flops, ms)]; call after a forward so buffers are live."""
        st = torch.cuda.current_stream(self.device)
        out = []
        for plan in (self._bb_plan, self._fpn_plan, self._cls_plan, self._box_plan):
            for fn, args, what, meta in plan:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(st.cuda_stream, *args)
                e0.record(st)
                for _ in range(reps):
                    fn(st.cuda_stream, *args)
                e1.record(st)
                e1.synchronize()
                out.append((what, meta['kind'], meta['bytes'], meta['flops'], e0.elapsed_time(e1) / reps))
        return out

    # --------------------------------------------------------------------------------- backbone
    def _build_backbone(self, bb, H, W):
        lib, B, dt = self.lib, self.B, self.dt
        dtp = dt | self.bb_pad                   # dtype argument of the entry points that pad (stem, depthwise): + the padding convention
        stem_c, stages = bb.arch
        plan = []
        Hs, Ws = _same_out(H, 2), _same_out(W, 2)
        # geometry pass: buffer sizes
        io_max, mid_max, part_max, exp_max = B * Hs * Ws * stem_c, 0, 0, 0
        h, w = Hs, Ws
        for blocks in stages:
            for b in blocks:
                ho, wo = _same_out(h, b['s']), _same_out(w, b['s'])
                mid_max = max(mid_max, B * ho * wo * b['mid'])
                if b['type'] == 'ir':
                    nblk = lib.effdet_mbconv_tiles_per_image(dtp, h, w, b['cin'], b['mid'], b['k'], b['s'])
                    if nblk <= 0:        # no fused geometry (very wide fp32 inputs): expand GEMM + depthwise kernels
                        # (accurate mode: the two-term MBConv kernels cover inputs up to 192 channels - every block of d0; wider
                        #  blocks take this unfused path too, with the expanded tensor in HBM)
                        nblk = lib.effdet_dwconv_blocks_per_image(ho, wo, b['mid'])
                        exp_max = max(exp_max, B * h * w * b['mid'])
                else:
                    nblk = max(lib.effdet_dwconv_blocks_per_image(ho, wo, b['mid']), lib.effdet_stem_dw_parts(dtp, H, W, stem_c))
                if nblk <= 0:
                    raise NotImplementedError('block geometry (mid=%d) is outside the built range' % b['mid'])
                part_max = max(part_max, B * nblk * b['mid'])
                io_max = max(io_max, B * ho * wo * b['cout'])
                h, w = ho, wo
        ping = [self._new(io_max), self._new(io_max)]
        dbuf = self._new(max(mid_max, 1))
        # Stage 0 = one residual-free depthwise-separable block followed by a residual-free MBConv block: the first
        # block's project conv + BN (linear) are folded into the second block's expand weights and its SE gate is
        # applied while that kernel loads its input tile, so the narrow tensor in between never exists.
        b00, b10 = stages[0][0], (stages[1][0] if len(stages) > 1 else None)
        self._compose01 = bool(len(stages[0]) == 1 and b10 is not None and b00['type'] == 'ds' and not b00['residual']
                               and b10['type'] == 'ir' and not b10['residual'] and b00['mid'] % 8 == 0 and
                               lib.effdet_mbconv_gated_tiles_per_image(dtp, Hs, Ws, b00['mid'], b10['mid'], b10['k'], b10['s']) > 0)
        dbuf2 = None
        if self._compose01:
            h1, w1_ = _same_out(Hs, b10['s']), _same_out(Ws, b10['s'])
            dbuf2 = self._new(B * h1 * w1_ * b10['mid'])
            part_max = max(part_max, B * lib.effdet_mbconv_gated_tiles_per_image(dtp, Hs, Ws, b00['mid'], b10['mid'], b10['k'], b10['s']) * b10['mid'])
        ebuf = self._new(exp_max) if exp_max else None
        partial = self._new(part_max, dtype=torch.float32)
        composed = None          # (W_proj folded, t3) of block 0.0 while block 1.0 is being planned
        gate_max = B * max(b['mid'] for blocks in stages for b in blocks)
        gate = self._new(gate_max, dtype=torch.float32)

        self.x_shape = (B, 3, H, W)
        s, t = self._fold(bb.bn1)
        s, t = self._f32(s), self._f32(t)
        b00 = stages[0][0]
        self._fuse_stem = (b00['type'] == 'ds' and b00['k'] == 3 and b00['s'] == 1 and stem_c <= 64)
        if self._fuse_stem and self.pair and lib.effdet_stem_dw_parts(dtp, H, W, stem_c) <= 0:
            self._fuse_stem = False              # accurate mode outside the fused stem's form (32 channels, even left pad): stem conv + depthwise
        if self._fuse_stem:
            # conv_stem + bn1 + SiLU + blocks.0.0.conv_dw + bn1 + SiLU in one launch (stem map stays in LDS)
            wk = torch.zeros(stem_c, 32, dtype=torch.float32, device=self.device)
            wk[:, :27] = bb.conv_stem.weight.detach().float().permute(0, 2, 3, 1).reshape(stem_c, 27).to(self.device)
            wk = self._f32(wk) if self.pair else self._w(wk)          # (two-term mode: the kernel splits the float32 stem weights itself)
            m0 = bb.blocks[0][0]
            s2, t2 = self._fold(m0.bn1)
            taps0 = self._f32(self._dw_taps(m0.conv_dw.weight))
            s2, t2 = self._f32(s2), self._f32(t2)
            self._stem = (wk, s, t, taps0, s2, t2, dbuf, partial, H, W, stem_c)
            self._stem_meta = dict(kind='stem_dw', flops=2 * B * Hs * Ws * stem_c * (27 + 9),
                                   bytes=B * (3 * H * W + Hs * Ws * stem_c) * self.pyr_es)
        else:
            wt = self._f32(bb.conv_stem.weight.detach().float().permute(2, 3, 1, 0).reshape(27, stem_c))
            self._stem = (wt, s, t, ping[0], H, W, stem_c)
            self._stem_meta = dict(kind='stem', flops=2 * 27 * B * Hs * Ws * stem_c,
                                   bytes=B * (3 * H * W + Hs * Ws * stem_c) * self.pyr_es)
        cur = ping[0]
        h, w = Hs, Ws
        self.feats = []
        for si, blocks in enumerate(stages):
            for bi, b in enumerate(blocks):
                m = bb.blocks[si][bi]
                ho, wo = _same_out(h, b['s']), _same_out(w, b['s'])
                last_of_feature_stage = (bi == len(blocks) - 1) and (si in (2, 4, 6))
                if last_of_feature_stage:
                    out = self._new(B, ho, wo, b['cout'])
                    self.feats.append(out)
                else:
                    out = ping[1] if (cur is not None and cur.data_ptr() == ping[0].data_ptr()) else ping[0]
                what = 'backbone.blocks.%d.%d' % (si, bi)
                es = self.pyr_es
                if b['type'] == 'ir':
                    # fused expand 1x1 + BN + SiLU -> depthwise + BN + SiLU (+ SE pool partials); the expanded
                    # activation stays in LDS
                    s1, t1 = self._fold(m.bn1)
                    s2, t2 = self._fold(m.bn2)
                    taps = self._f32(self._dw_taps(m.conv_dw.weight))
                    s2, t2 = self._f32(s2), self._f32(t2)
                    if composed is not None:
                        # expand(project(x)) = W_e (S3 W_p (g*x) + t3) = (W_e S3 W_p)(g*x) + W_e t3: one conv over the
                        # previous block's gated depthwise output; the constant goes into the BN shift
                        wp, s3c, t3c, cmid = composed
                        we = m.conv_pw.weight.detach().reshape(b['mid'], b['cin']).to(device=self.device, dtype=torch.float32)
                        wcomb = we @ (s3c.view(-1, 1) * wp)                           # [mid][cmid]
                        t1 = t1.to(self.device) + s1.to(self.device) * (we @ t3c)
                        w1 = self._w(wcomb)
                        s1, t1 = self._f32(s1), self._f32(t1)
                        nblk = lib.effdet_mbconv_gated_tiles_per_image(dtp, h, w, cmid, b['mid'], b['k'], b['s'])
                        plan.append((lib.effdet_mbconv_expand_dw_gated,
                                     (dtp, dbuf.data_ptr(), gate.data_ptr(), dbuf2.data_ptr(), w1.data_ptr(), s1.data_ptr(), t1.data_ptr(),
                                      taps.data_ptr(), s2.data_ptr(), t2.data_ptr(), partial.data_ptr(),
                                      B, h, w, cmid, b['mid'], b['k'], b['s']), what + '.conv_pw(+blocks.0.0.conv_pw)+conv_dw',
                                     dict(kind='mbconv', bytes=B * (h * w * cmid + ho * wo * b['mid']) * es + b['mid'] * cmid * es,
                                          flops=2 * B * (h * w * cmid * b['mid'] + b['k'] * b['k'] * ho * wo * b['mid']))))
                        composed = None
                        mid_buf = dbuf2
                        pw_out, bn_out = m.conv_pwl, m.bn3
                    else:
                      w1 = self._w(m.conv_pw.weight.reshape(b['mid'], b['cin']))
                      s1, t1 = self._f32(s1), self._f32(t1)
                      nblk = lib.effdet_mbconv_tiles_per_image(dtp, h, w, b['cin'], b['mid'], b['k'], b['s'])
                      mid_buf = dbuf
                    if mid_buf is dbuf2:
                        pass
                    elif nblk > 0:
                        plan.append((lib.effdet_mbconv_expand_dw,
                                     (dtp, cur.data_ptr(), dbuf.data_ptr(), w1.data_ptr(), s1.data_ptr(), t1.data_ptr(),
                                      taps.data_ptr(), s2.data_ptr(), t2.data_ptr(), partial.data_ptr(),
                                      B, h, w, b['cin'], b['mid'], b['k'], b['s']), what + '.conv_pw+conv_dw',
                                     dict(kind='mbconv', bytes=B * (h * w * b['cin'] + ho * wo * b['mid']) * es + b['mid'] * b['cin'] * es,
                                          flops=2 * B * (h * w * b['cin'] * b['mid'] + b['k'] * b['k'] * ho * wo * b['mid']))))
                    else:
                        nblk = lib.effdet_dwconv_blocks_per_image(ho, wo, b['mid'])
                        plan.append((lib.effdet_pw_gemm_bn_act,
                                     (dt, cur.data_ptr(), B * h * w, b['cin'], w1.data_ptr(), b['mid'], s1.data_ptr(),
                                      t1.data_ptr(), 1, None, None, 0, ebuf.data_ptr(), 0, 0), what + '.conv_pw',
                                     self._gemm_meta(B * h * w, b['cin'], b['mid'])))
                        plan.append((lib.effdet_dwconv_bn_act,
                                     (dtp, ebuf.data_ptr(), dbuf.data_ptr(), taps.data_ptr(), s2.data_ptr(), t2.data_ptr(), 1,
                                      partial.data_ptr(), B, h, w, b['mid'], b['k'], b['s']), what + '.conv_dw',
                                     dict(kind='dwconv', bytes=B * (h * w + ho * wo) * b['mid'] * es,
                                          flops=2 * b['k'] * b['k'] * B * ho * wo * b['mid'])))
                    pw_out, bn_out = m.conv_pwl, m.bn3
                elif si == 0 and bi == 0 and self._fuse_stem:
                    nblk = lib.effdet_stem_dw_parts(dt, H, W, stem_c)     # launched by run_backbone (takes x)
                    pw_out, bn_out = m.conv_pw, m.bn2
                    mid_buf = dbuf
                else:
                    s2, t2 = self._fold(m.bn1)
                    taps = self._f32(self._dw_taps(m.conv_dw.weight))
                    s2, t2 = self._f32(s2), self._f32(t2)
                    nblk = lib.effdet_dwconv_blocks_per_image(ho, wo, b['mid'])
                    plan.append((lib.effdet_dwconv_bn_act,
                                 (dtp, cur.data_ptr(), dbuf.data_ptr(), taps.data_ptr(), s2.data_ptr(), t2.data_ptr(), 1,
                                  partial.data_ptr(), B, h, w, b['mid'], b['k'], b['s']), what + '.conv_dw',
                                 dict(kind='dwconv', bytes=B * (h * w + ho * wo) * b['mid'] * es,
                                      flops=2 * b['k'] * b['k'] * B * ho * wo * b['mid'])))
                    pw_out, bn_out = m.conv_pw, m.bn2
                    mid_buf = dbuf
                W1 = self._f32(m.se.conv_reduce.weight.reshape(b['se'], b['mid']))
                b1 = self._f32(m.se.conv_reduce.bias)
                W2 = self._f32(m.se.conv_expand.weight.reshape(b['mid'], b['se']).t())      # [R][C]: coalesced over channels
                b2 = self._f32(m.se.conv_expand.bias)
                plan.append((lib.effdet_se_gate,
                             (partial.data_ptr(), nblk, ho * wo, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                              gate.data_ptr(), B, b['mid'], b['se']), what + '.se',
                             dict(kind='se_gate', bytes=B * (nblk + 1) * b['mid'] * 4 + 8 * b['mid'] * b['se'],
                                  flops=4 * B * b['mid'] * b['se'])))
                s3, t3 = self._fold(bn_out)
                if si == 0 and bi == 0 and self._compose01 and self._fuse_stem:
                    # no project launch: hand (W_p, s3, t3) to the next block (see above)
                    composed = (pw_out.weight.detach().reshape(b['cout'], b['mid']).to(device=self.device, dtype=torch.float32),
                                s3.to(device=self.device, dtype=torch.float32), t3.to(device=self.device, dtype=torch.float32), b['mid'])
                    cur, h, w = None, ho, wo
                    continue
                w3 = self._w(pw_out.weight.reshape(b['cout'], b['mid']))
                s3, t3 = self._f32(s3), self._f32(t3)
                plan.append((lib.effdet_pw_gemm_bn_act,
                             (dt, mid_buf.data_ptr(), B * ho * wo, b['mid'], w3.data_ptr(), b['cout'], s3.data_ptr(),
                              t3.data_ptr(), 0, cur.data_ptr() if b['residual'] else None, gate.data_ptr(), ho * wo,
                              out.data_ptr(), 0, 0), what + '.conv_pwl',
                             self._gemm_meta(B * ho * wo, b['mid'], b['cout'], residual=b['residual'], gate=True)))
                cur, h, w = out, ho, wo
        self._bb_plan = plan
        self.feat_hw = [(f.shape[1], f.shape[2]) for f in self.feats]

    def run_backbone(self, x, ret=True):
        if tuple(x.shape) != self.x_shape:
            raise ValueError('engine was prepared for input %s, got %s' % (self.x_shape, tuple(x.shape)))
        if x.device != self.device or (x.dtype not in _DT and x.dtype != torch.uint8):
            raise RuntimeError('input must be a float32/bfloat16 (normalised) or uint8 (raw) tensor on %s' % (self.device,))
        if self.pair and x.dtype == torch.bfloat16:
            x = x.float()
        x = x.contiguous()
        self._stem_call(x)
        self._run(self._bb_plan)
        return [self._act_out(f).permute(0, 3, 1, 2) for f in self.feats] if ret else None

    def _stem_call(self, x):
        st = torch.cuda.current_stream(self.device).cuda_stream
        if x.dtype == torch.uint8:
            # raw images: the loader's (x - 255*mean) / (255*std) (effdet/data/loader.py:114-128) happens inside the
            # stem kernel's input load
            import ctypes
            mean = (ctypes.c_float * 3)(*[255.0 * v for v in self.input_mean])
            std = (ctypes.c_float * 3)(*[255.0 * v for v in self.input_std])
            if self._fuse_stem:
                wk, s, t, taps0, s2, t2, dbuf, partial, H, W, c = self._stem
                _lib.check(self.lib.effdet_stem_dw_fused_u8(st, self.dt | self.bb_pad, x.data_ptr(), mean, std, wk.data_ptr(), s.data_ptr(),
                                                            t.data_ptr(), taps0.data_ptr(), s2.data_ptr(), t2.data_ptr(),
                                                            dbuf.data_ptr(), partial.data_ptr(), self.B, H, W, c),
                           'backbone.conv_stem+blocks.0.0.conv_dw (uint8)')
            else:
                wt, s, t, out, H, W, c = self._stem
                _lib.check(self.lib.effdet_stem_conv_u8(st, self.dt | self.bb_pad, x.data_ptr(), mean, std, wt.data_ptr(), s.data_ptr(),
                                                        t.data_ptr(), out.data_ptr(), self.B, H, W, c), 'backbone.conv_stem (uint8)')
            return
        if self._fuse_stem:
            wk, s, t, taps0, s2, t2, dbuf, partial, H, W, c = self._stem
            _lib.check(self.lib.effdet_stem_dw_fused(st, _DT[x.dtype], self.dt | self.bb_pad, x.data_ptr(), wk.data_ptr(), s.data_ptr(),
                                                     t.data_ptr(), taps0.data_ptr(), s2.data_ptr(), t2.data_ptr(),
                                                     dbuf.data_ptr(), partial.data_ptr(), self.B, H, W, c),
                       'backbone.conv_stem+blocks.0.0.conv_dw')
        else:
            wt, s, t, out, H, W, c = self._stem
            _lib.check(self.lib.effdet_stem_conv(st, _DT[x.dtype], self.dt | self.bb_pad, x.data_ptr(), wt.data_ptr(), s.data_ptr(),
                                                 t.data_ptr(), out.data_ptr(), self.B, H, W, c), 'backbone.conv_stem')

    # ------------------------------------------------------------------------------------- BiFPN
    def _build_fpn(self, fpn):
        lib, B, dt, F, cfg = self.lib, self.B, self.dt, self.F, self.cfg
        plan = []
        lateral = []                            # 1x1 convs of backbone features (all depend on the backbone only): grouped below
        in_info = fpn.in_feature_info
        nbb = len(in_info)
        # level geometry
        hw = list(self.feat_hw)
        while len(hw) < self.L:
            hw.append((_same_out(hw[-1][0], 2), _same_out(hw[-1][1], 2)))
        self.level_hw = hw
        offs, P = [], 0
        for (h, w) in hw:
            offs.append(P)
            P += h * w
        self.level_off, self.P = offs, P
        self.pyr = self._new(B, P, F)

        def dense(level):
            h, w = hw[level]
            return self._new(B, h, w, F)

        # inputs of cell 0: backbone features (raw channels) + extra levels by conv+BN+maxpool / maxpool
        self.fpn_in = list(self.feats)          # raw backbone maps, [B,h,w,c]
        level_src = [None] * self.L             # (tensor, image_stride) of F-channel maps for levels >= nbb
        prev_c = in_info[-1]['num_chs']
        prev = self.feats[-1]
        for level in range(nbb, self.L):
            rs = fpn.resample[str(level)]
            ph, pw_ = hw[level - 1]
            src = prev
            if prev_c != F:
                conv = rs.conv
                tmp = dense(level - 1)
                wq = self._w(conv.conv.weight.reshape(F, prev_c))
                if conv.bn is not None:
                    s, t = self._fold(conv.bn, conv.conv.bias)
                    s, t = self._f32(s), self._f32(t)
                    sp = s.data_ptr()
                else:
                    t = self._f32(conv.conv.bias)
                    sp = None
                lateral.append((prev.data_ptr(), B * ph * pw_, prev_c, wq.data_ptr(), sp, t.data_ptr(), tmp.data_ptr(),
                                'fpn.resample.%d.conv' % level))
                src = tmp
            out = dense(level)
            plan.append((lib.effdet_maxpool_same, (dt | self.fpn_pad, src.data_ptr(), 0, out.data_ptr(), 0, B, ph, pw_, F),
                         'fpn.resample.%d.downsample' % level,
                         dict(kind='maxpool', bytes=B * (ph * pw_ + hw[level][0] * hw[level][1]) * F * self.pyr_es, flops=0)))
            level_src[level] = out
            prev, prev_c = out, F

        nodes = fpn.fpn_config.nodes
        node_red = [n['reduction'] for n in nodes]
        n_cells = len(fpn.cell)
        red0 = in_info[0]['reduction']

        def level_of(reduction):
            return int(round(math.log2(reduction / red0)))

        x = []
        for i in range(nbb):
            x.append(dict(raw=True, t=self.feats[i], chs=in_info[i]['num_chs'], level=i))
        for level in range(nbb, self.L):
            x.append(dict(raw=False, ptr=level_src[level].data_ptr(), stride=hw[level][0] * hw[level][1] * F, level=level))

        for ci in range(n_cells):
            layer = fpn.cell[ci]
            last_cell = ci == n_cells - 1
            for ni, node in enumerate(nodes):
                fn = layer.fnode[ni]
                lvl = level_of(node['reduction'])
                h, w = hw[lvl]
                ins = []
                for off in node['inputs_offsets']:
                    src = x[off]
                    if src.get('raw'):
                        # lateral 1x1 conv + BN of a backbone feature (own weights per use)
                        rs = fn.combine.resample[str(off)]
                        conv = rs.conv
                        lat = dense(src['level'])
                        sh, sw = hw[src['level']]
                        wq = self._w(conv.conv.weight.reshape(F, src['chs']))
                        if conv.bn is not None:
                            s, t = self._fold(conv.bn, conv.conv.bias)
                            s, t = self._f32(s), self._f32(t)
                            sp = s.data_ptr()
                        else:
                            t = self._f32(conv.conv.bias)
                            sp = None
                        lateral.append((src['t'].data_ptr(), B * sh * sw, src['chs'], wq.data_ptr(), sp, t.data_ptr(), lat.data_ptr(),
                                        'fpn.cell.%d.fnode.%d.combine.resample.%d.conv' % (ci, ni, off)))
                        src = dict(raw=False, ptr=lat.data_ptr(), stride=sh * sw * F, level=src['level'])
                    d = src['level'] - lvl
                    mode = 0 if d == 0 else (1 if d == 1 else (2 if d == -1 else None))
                    if mode is None:
                        raise NotImplementedError('BiFPN edge spanning %d levels' % d)
                    ins.append((src['ptr'], src['stride'], hw[src['level']], mode))
                # fusion weights, computed like FpnCombine.forward (efficientdet.py:232-244)
                method = node['weight_method']
                if method == 'fastattn':
                    ew = torch.relu(fn.combine.edge_weights.detach().float())
                    den = float((ew.sum() + 0.0001).item())
                    fuse_mode, fw = 1, [float(v) for v in ew.tolist()]
                elif method == 'attn':
                    ew = torch.softmax(fn.combine.edge_weights.detach().float(), dim=0)
                    fuse_mode, fw, den = 2, [float(v) for v in ew.tolist()], 1.0
                else:
                    fuse_mode, fw, den = 2, [1.0] * len(ins), 1.0
                sc = fn.after_combine.conv
                taps = self._f32(self._dw_taps(sc.conv_dw.weight))
                wq = self._w(sc.conv_pw.weight.reshape(F, F))
                s, t = self._fold(sc.bn, sc.conv_pw.bias)
                s, t = self._f32(s), self._f32(t)
                if last_cell and ni >= len(nodes) - self.L:
                    out_ptr = self.pyr.data_ptr() + self.level_off[lvl] * F * self.pyr.element_size()
                    out_stride = P * F
                else:
                    o = dense(lvl)
                    out_ptr, out_stride = o.data_ptr(), h * w * F
                plan.append(self._sepconv_call(
                    [(h, w)], [ins], fuse_mode, fw, den, 1, taps, wq, s, t, [0], 0, F, F,
                    [out_ptr], [out_stride], 'fpn.cell.%d.fnode.%d' % (ci, ni)))
                x.append(dict(raw=False, ptr=out_ptr, stride=out_stride, level=lvl))
            x = x[-self.L:]
        # the lateral convs read backbone features only, so they can all run first - in ONE launch when they share the output
        # tile (F <= 96), else one launch each
        head = []
        if lateral and F <= 96 and len(lateral) <= 8:
            n = len(lateral)
            c_a = _arr(ctypes.c_void_p, [l[0] for l in lateral]); c_m = _arr(ctypes.c_longlong, [l[1] for l in lateral])
            c_k = _arr(ctypes.c_int, [l[2] for l in lateral]); c_w = _arr(ctypes.c_void_p, [l[3] for l in lateral])
            c_n = _arr(ctypes.c_int, [F] * n); c_s = _arr(ctypes.c_void_p, [l[4] for l in lateral])
            c_t = _arr(ctypes.c_void_p, [l[5] for l in lateral]); c_c = _arr(ctypes.c_void_p, [l[6] for l in lateral])
            self._keep += [c_a, c_m, c_k, c_w, c_n, c_s, c_t, c_c]
            es = self.pyr_es
            meta = dict(kind='pw_gemm', bytes=sum((l[1] * l[2] + l[1] * F + F * l[2]) * es for l in lateral),
                        flops=sum(2 * l[1] * l[2] * F for l in lateral))
            head.append((lib.effdet_pw_gemm_group, (dt, n, c_a, c_m, c_k, c_w, c_n, c_s, c_t, 0, c_c),
                         'fpn lateral 1x1 convs (%s)' % ', '.join(l[7].replace('fpn.', '').replace('.combine.resample', '').replace('.conv', '') for l in lateral), meta))
        else:
            for l in lateral:
                head.append((lib.effdet_pw_gemm_bn_act, (dt, l[0], l[1], l[2], l[3], F, l[4], l[5], 0, None, None, 0, l[6], 0, 0),
                             l[7], self._gemm_meta(l[1], l[2], F)))
        self._fpn_plan = head + plan

    def _sepconv_call(self, level_hw, level_inputs, fuse_mode, fw, den, pre_act, taps, wq, scale, shift, affine_rows,
                      post_act, F, N, out_ptrs, out_strides, what, ood=None, out_f32=False):
        nl, n_in = len(level_hw), len(level_inputs[0])
        c_hw = _arr(ctypes.c_int, [v for hw in level_hw for v in hw])
        c_ptr = _arr(ctypes.c_void_p, [i[0] for lv in level_inputs for i in lv])
        c_str = _arr(ctypes.c_longlong, [i[1] for lv in level_inputs for i in lv])
        c_ihw = _arr(ctypes.c_int, [v for lv in level_inputs for i in lv for v in i[2]])
        c_mode = _arr(ctypes.c_int, [i[3] for lv in level_inputs for i in lv])
        c_fw = _arr(ctypes.c_float, list(fw) + [0.0] * (3 - len(fw)))
        c_aff = _arr(ctypes.c_int, affine_rows)
        c_out = _arr(ctypes.c_void_p, out_ptrs)
        c_ostr = _arr(ctypes.c_longlong, out_strides)
        if ood is None:
            ood_args = (0, self.A, None, None, 0, None)
        else:
            c_ooff = _arr(ctypes.c_longlong, ood['level_off'])
            self._keep.append(c_ooff)
            ood_args = (ood['classes'], self.A, ood['energy'].data_ptr(), ood['maxlogit'].data_ptr(), ood['stride'], c_ooff)
        self._keep += [c_hw, c_ptr, c_str, c_ihw, c_mode, c_fw, c_aff, c_out, c_ostr]
        args = (self.dt | (2 if (out_f32 and self.dt == 1) else 0) | (4 if (out_f32 and self.dt == _PAIR) else 0) | self.fpn_pad, self.B, nl, c_hw, n_in, c_ptr, c_str, c_ihw, c_mode, fuse_mode, c_fw, ctypes.c_float(den), pre_act,
                taps.data_ptr(), wq.data_ptr(), scale.data_ptr() if scale is not None else None, shift.data_ptr(),
                c_aff, post_act, F, N, c_out, c_ostr) + ood_args
        es = self.pyr_es
        in_px = sum(i[2][0] * i[2][1] for lv in level_inputs for i in lv)
        out_px = sum(h * w for h, w in level_hw)
        oes_ = 4 if out_f32 else es
        meta = dict(kind='sepconv', bytes=self.B * (in_px * F * es + out_px * N * oes_) + N * F * es + 9 * F * 4 +
                    (2 * self.B * out_px * self.A * 4 if ood is not None else 0),
                    flops=2 * self.B * out_px * (9 * F + F * N))
        return (self.lib.effdet_sepconv_fused, args, what, meta)

    def _load_feature_list(self, xs, dst_tensors):
        for src, dst in zip(xs, dst_tensors):
            if src.device != self.device:
                raise RuntimeError('feature maps must live on %s' % (self.device,))
            dst.copy_(self._act_in(src.permute(0, 2, 3, 1)))

    def pyramid_views(self):
        out = []
        pyr = self._act_out(self.pyr)
        for (h, w), off in zip(self.level_hw, self.level_off):
            out.append(pyr[:, off:off + h * w, :].unflatten(1, (h, w)).permute(0, 3, 1, 2))
        return out

    def run_fpn(self, feats, ret=True):
        if feats is not None:
            if len(feats) != len(self.feats):
                raise ValueError('expected %d backbone feature maps' % len(self.feats))
            self._load_feature_list(feats, self.feats)
        self._run(self._fpn_plan)
        return self.pyramid_views() if ret else None

    # ------------------------------------------------------------------------------------- heads
    def _build_heads(self, model):
        lib, B, dt, F, cfg = self.lib, self.B, self.dt, self.F, self.cfg
        A, C, L, P = self.A, self.C, self.L, self.P
        N = A * P
        self.N = N
        # (accurate mode: the logits leave the class head as plain float32, like the box regressions)
        self.cls_all = self._new(B, N, C)
        # box regressions are written as float32 straight from the accumulators, whatever the model dtype (1.2 MB / image): decode
        # (anchors.py:136 `.float()`) then sees unrounded values
        self.box_all = self._new(B, N, 4, dtype=torch.float32)
        if self._ood_out is not None:                      # caller-provided [B, N] float32 views (DetBenchPredict's split batches)
            self.ood_energy, self.ood_max_logit = self._ood_out
            if tuple(self.ood_energy.shape) != (B, N) or tuple(self.ood_max_logit.shape) != (B, N) or not self.ood_energy.is_contiguous() \
                    or not self.ood_max_logit.is_contiguous() or self.ood_energy.dtype != torch.float32:
                raise ValueError('ood_out must be two contiguous float32 [B, N] tensors')
        else:
            self.ood_energy = self._new(B, N, dtype=torch.float32)
            self.ood_max_logit = self._new(B, N, dtype=torch.float32)
        t1, t2 = self._new(B, P, F), self._new(B, P, F)
        es = self.pyr.element_size()

        def level_ptrs(t, width):
            return [t.data_ptr() + off * width * es for off in self.level_off]

        def plan_for(head, name, out_t, K, ood, out_f32=False):
            plan = []
            src = self.pyr
            bufs = [t1, t2]
            for r in range(cfg.box_class_repeats):
                conv = head.conv_rep[r]
                taps = self._f32(self._dw_taps(conv.conv_dw.weight))
                wq = self._w(conv.conv_pw.weight.reshape(F, F))
                ss, ts = [], []
                for l in range(L):
                    s, t = self._fold(head.bn_rep[r][l].bn, conv.conv_pw.bias)
                    ss.append(s)
                    ts.append(t)
                s, t = self._f32(torch.stack(ss)), self._f32(torch.stack(ts))
                dst = bufs[r % 2]
                ins = [[(p, P * F, hw, 0)] for p, hw in zip(level_ptrs(src, F), self.level_hw)]
                plan.append(self._sepconv_call(self.level_hw, ins, 0, [], 1.0, 0, taps, wq, s, t, list(range(L)), 1, F, F,
                                               level_ptrs(dst, F), [P * F] * L, '%s.conv_rep.%d' % (name, r)))
                src = dst
            conv = head.predict
            NO = A * K
            taps = self._f32(self._dw_taps(conv.conv_dw.weight))
            wq = self._w(conv.conv_pw.weight.reshape(NO, F))
            t = self._f32(conv.conv_pw.bias.detach().float().reshape(1, NO))
            ins = [[(p, P * F, hw, 0)] for p, hw in zip(level_ptrs(src, F), self.level_hw)]
            oes = out_t.element_size()
            outs = [out_t.data_ptr() + off * NO * oes for off in self.level_off]
            oodd = None
            if ood:
                oodd = dict(classes=K, energy=self.ood_energy, maxlogit=self.ood_max_logit, stride=N,
                            level_off=[off * A for off in self.level_off])
            plan.append(self._sepconv_call(self.level_hw, ins, 0, [], 1.0, 0, taps, wq, None, t, [0] * L, 0, F, NO,
                                           outs, [P * NO] * L, '%s.predict' % name, ood=oodd, out_f32=out_f32 or self.pair))
            return plan

        # infer.py:186-191 replaces `model.class_net` by a MetaHead (functional head, own launches in effdet/meta_head.py):
        # backbone / BiFPN / box head then still run from this plan, the class outputs come from the MetaHead's forward
        self._cls_plan = plan_for(model.class_net, 'class_net', self.cls_all, C, True) if hasattr(model.class_net, 'conv_rep') else None
        self._box_plan = plan_for(model.box_net, 'box_net', self.box_all, 4, False, out_f32=True)

    def head_views(self, t, K):
        out = []
        A = self.A
        for (h, w), off in zip(self.level_hw, self.level_off):
            v = t[:, off * A:(off + h * w) * A, :].reshape(self.B, h, w, A * K)
            out.append(v.permute(0, 3, 1, 2))
        return out

    def run_heads(self, activs, want_cls, want_box):
        if activs is not None:
            if len(activs) != self.L:
                raise ValueError('expected %d pyramid levels' % self.L)
            for src, (h, w), off in zip(activs, self.level_hw, self.level_off):
                self.pyr[:, off:off + h * w, :].copy_(self._act_in(src.permute(0, 2, 3, 1).reshape(self.B, h * w, self.F)))
        cls_o = box_o = None
        if want_cls:
            if self._cls_plan is None:
                raise RuntimeError('model.class_net is not a HeadNet (MetaHead?): its outputs come from the head\'s own forward')
            self._run(self._cls_plan)
            cls_o = self.head_views(self.cls_all, self.C)
        if want_box:
            self._run(self._box_plan)
            box_o = self.head_views(self.box_all, 4)
        return cls_o, box_o
