"""EfficientDet model surface (`EfficientDet`, `BiFpn`, `HeadNet`, ...) on the HIP engine.

Mirrors the class / attribute / state-dict layout of the reference's effdet/efficientdet.py
(ConvBnAct2d :42, SeparableConv2d :60, ResampleFeatureMap :140, FpnCombine :196, Fnode :248,
BiFpnLayer :261, BiFpn :303, HeadNet :368, _init_weight :472, EfficientDet :831) so that
checkpoints (`backbone.*`, `fpn.resample.{3,4}...`, `fpn.cell.{c}.fnode.{n}...`,
`{class,box}_net.{conv_rep,bn_rep,predict}...`) load with `strict=True` and scripts that poke at
`model.config`, `model.class_net`, `model.fpn` keep working.

The sub-modules here are parameter containers: the arithmetic of every mode of
`EfficientDet.forward` runs in hand-written HIP kernels through `engine.Engine`; there is no
PyTorch / CPU fallback, and calling a container's own `forward` raises.
"""
import logging
import math
from collections import OrderedDict

import weakref

import torch
import torch.nn as nn

from ..backbone import create_backbone
from .config import get_fpn_config, set_config_readonly, set_config_writeable


def _no_forward(self, *a, **k):
    raise RuntimeError('%s is a parameter container of the HIP engine; run it through '
                       'EfficientDet.forward(..., mode=...)' % type(self).__name__)


# Parameter / buffer / sub-module registrations on the modules of ONE model (weights_token, train_signature): every EfficientDet
# enters its modules here with its own counter, and torch's global registration hooks bump the counter of the model that owns the
# module being changed - a registration anywhere else in the process (a validation bench, an EMA copy, another model) leaves this
# model's packed engine, training tables and captured graph alone.
_MODULE_OWNER = weakref.WeakKeyDictionary()          # module -> that model's [count]


def _count_registration(module, *_args):
    c = _MODULE_OWNER.get(module)
    if c is not None:
        c[0] += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_count_registration)
torch.nn.modules.module.register_module_buffer_registration_hook(_count_registration)
torch.nn.modules.module.register_module_module_registration_hook(_count_registration)


class _Holder(nn.Module):
    forward = _no_forward


def _check(config):
    if config.pad_type not in ('same', ''):
        raise NotImplementedError("pad_type must be 'same' (TF-SAME, the tf_ model family) or '' (static symmetric); got %r" % (config.pad_type,))
    if (config.act_type or 'swish') not in ('swish', 'silu') or (getattr(config, 'head_act_type', None) or 'swish') not in ('swish', 'silu'):
        raise NotImplementedError('only the swish/SiLU activation is built')
    if config.norm_layer is not None:
        raise NotImplementedError('only BatchNorm2d norm layers are built')
    if not config.separable_conv or config.conv_bn_relu_pattern or config.conv_after_downsample:
        raise NotImplementedError('only separable_conv=True, conv_bn_relu_pattern=False, conv_after_downsample=False is built')
    if config.downsample_type != 'max' or config.upsample_type != 'nearest':
        raise NotImplementedError("only downsample 'max' / upsample 'nearest' is built")


def _bn(config, c):
    kw = dict(config.norm_kwargs or {})
    return nn.BatchNorm2d(c, **kw)


class ConvBnAct2d(_Holder):
    def __init__(self, in_channels, out_channels, kernel_size, bias=False, bn=None):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=bias)
        self.bn = bn


class SeparableConv2d(_Holder):
    def __init__(self, in_channels, out_channels, kernel_size=3, bias=False, bn=None):
        super().__init__()
        self.conv_dw = nn.Conv2d(in_channels, in_channels, kernel_size, groups=in_channels, bias=False)
        self.conv_pw = nn.Conv2d(in_channels, out_channels, 1, bias=bias)
        self.bn = bn


class ResampleFeatureMap(nn.Sequential):
    """1x1 conv(+BN) when channels differ, then max-pool (ratio > 1) or nearest upsample (< 1)."""
    forward = _no_forward

    def __init__(self, config, in_channels, out_channels, reduction_ratio=1.):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.reduction_ratio = reduction_ratio
        self.conv_after_downsample = False
        if in_channels != out_channels:
            apply_bn = config.apply_resample_bn
            self.add_module('conv', ConvBnAct2d(
                in_channels, out_channels, 1, bias=(not apply_bn) or config.redundant_bias,
                bn=_bn(config, out_channels) if apply_bn else None))
        if reduction_ratio > 1:
            self.add_module('downsample', _Holder())
        elif reduction_ratio < 1:
            self.add_module('upsample', _Holder())


class FpnCombine(_Holder):
    def __init__(self, config, feature_info, fpn_nodes, inputs_offsets, target_reduction, weight_method):
        super().__init__()
        self.inputs_offsets = tuple(inputs_offsets)
        self.weight_method = weight_method
        self.resample = nn.ModuleDict()
        for offset in self.inputs_offsets:
            in_channels = config.fpn_channels
            if offset < len(feature_info):
                in_channels = feature_info[offset]['num_chs']
                input_reduction = feature_info[offset]['reduction']
            else:
                input_reduction = fpn_nodes[offset - len(feature_info)]['reduction']
            self.resample[str(offset)] = ResampleFeatureMap(
                config, in_channels, config.fpn_channels, reduction_ratio=target_reduction / input_reduction)
        if weight_method in ('attn', 'fastattn'):
            self.edge_weights = nn.Parameter(torch.ones(len(self.inputs_offsets)), requires_grad=True)
        elif weight_method == 'sum':
            self.edge_weights = None
        else:
            raise ValueError('unknown weight_method {}'.format(weight_method))


class Fnode(_Holder):
    def __init__(self, combine, after_combine):
        super().__init__()
        self.combine = combine
        self.after_combine = after_combine


class BiFpnLayer(_Holder):
    def __init__(self, config, feature_info, fpn_config):
        super().__init__()
        self.num_levels = config.num_levels
        self.fnode = nn.ModuleList()
        self.feature_info = []
        for i, fnode_cfg in enumerate(fpn_config.nodes):
            logging.debug('fnode {} : {}'.format(i, fnode_cfg))
            combine = FpnCombine(config, feature_info, fpn_config.nodes, fnode_cfg['inputs_offsets'],
                                 fnode_cfg['reduction'], fnode_cfg['weight_method'])
            after_combine = nn.Sequential()
            after_combine.add_module('act', _Holder())
            after_combine.add_module('conv', SeparableConv2d(
                config.fpn_channels, config.fpn_channels, 3, bias=config.redundant_bias,
                bn=_bn(config, config.fpn_channels)))
            self.fnode.append(Fnode(combine, after_combine))
            self.feature_info.append(dict(num_chs=config.fpn_channels, reduction=fnode_cfg['reduction']))
        self.feature_info = self.feature_info[-self.num_levels::]


class SequentialList(nn.Sequential):
    forward = _no_forward


class BiFpn(_Holder):
    def __init__(self, config, feature_info):
        super().__init__()
        self.num_levels = config.num_levels
        fpn_config = config.fpn_config or get_fpn_config(
            config.fpn_name, min_level=config.min_level, max_level=config.max_level)
        self.fpn_config = fpn_config
        feature_info = [dict(f) for f in feature_info]
        self.in_feature_info = [dict(f) for f in feature_info]
        self.resample = nn.ModuleDict()
        for level in range(config.num_levels):
            if level < len(feature_info):
                in_chs = feature_info[level]['num_chs']
                reduction = feature_info[level]['reduction']
            else:
                self.resample[str(level)] = ResampleFeatureMap(config, in_chs, config.fpn_channels, reduction_ratio=2)
                in_chs = config.fpn_channels
                reduction = int(reduction * 2)
                feature_info.append(dict(num_chs=in_chs, reduction=reduction))
        self.level_feature_info = [dict(f) for f in feature_info]
        self.cell = SequentialList()
        for rep in range(config.fpn_cell_repeats):
            layer = BiFpnLayer(config, feature_info, fpn_config)
            self.cell.add_module(str(rep), layer)
            feature_info = layer.feature_info


class HeadNet(_Holder):
    def __init__(self, config, num_outputs):
        super().__init__()
        self.num_levels = config.num_levels
        self.bn_level_first = getattr(config, 'head_bn_level_first', False)
        if self.bn_level_first:
            raise NotImplementedError('head_bn_level_first=True (torchscript layout) is not built')
        F = config.fpn_channels
        self.conv_rep = nn.ModuleList([
            SeparableConv2d(F, F, 3, bias=config.redundant_bias) for _ in range(config.box_class_repeats)])
        self.bn_rep = nn.ModuleList()
        for _ in range(config.box_class_repeats):
            self.bn_rep.append(nn.ModuleList([
                nn.Sequential(OrderedDict([('bn', _bn(config, F))])) for _ in range(self.num_levels)]))
        self.act = _Holder()
        num_anchors = len(config.aspect_ratios) * config.num_scales
        self.predict = SeparableConv2d(F, num_outputs * num_anchors, 3, bias=True)


def _init_weight(m, n=''):
    """TF-style init of the reference (effdet/efficientdet.py:472-537)."""

    def _fan_in_out(w, groups=1):
        rf = w[0][0].numel() if w.dim() > 2 else 1
        return w.size(1) * rf, (w.size(0) * rf) // groups

    def _glorot_uniform(w, gain=1, groups=1):
        fan_in, fan_out = _fan_in_out(w, groups)
        gain /= max(1., (fan_in + fan_out) / 2.)
        limit = math.sqrt(3.0 * gain)
        w.data.uniform_(-limit, limit)

    def _variance_scaling(w, gain=1, groups=1):
        fan_in, _ = _fan_in_out(w, groups)
        gain /= max(1., fan_in)
        w.data.normal_(std=math.sqrt(gain))

    if isinstance(m, SeparableConv2d):
        if 'box_net' in n or 'class_net' in n:
            _variance_scaling(m.conv_dw.weight, groups=m.conv_dw.groups)
            _variance_scaling(m.conv_pw.weight)
            if m.conv_pw.bias is not None:
                if 'class_net.predict' in n:
                    m.conv_pw.bias.data.fill_(-math.log((1 - 0.01) / 0.01))
                else:
                    m.conv_pw.bias.data.zero_()
        else:
            _glorot_uniform(m.conv_dw.weight, groups=m.conv_dw.groups)
            _glorot_uniform(m.conv_pw.weight)
            if m.conv_pw.bias is not None:
                m.conv_pw.bias.data.zero_()
    elif isinstance(m, ConvBnAct2d):
        if 'box_net' in n or 'class_net' in n:
            m.conv.weight.data.normal_(std=.01)
            if m.conv.bias is not None:
                m.conv.bias.data.fill_(-math.log((1 - 0.01) / 0.01)) if 'class_net.predict' in n else m.conv.bias.data.zero_()
        else:
            _glorot_uniform(m.conv.weight)
            if m.conv.bias is not None:
                m.conv.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1.0)
        m.bias.data.zero_()


def _init_weight_alt(m, n=''):
    """Alternative init (effdet/efficientdet.py:540-555)."""
    if isinstance(m, nn.Conv2d):
        fan_out = (m.kernel_size[0] * m.kernel_size[1] * m.out_channels) // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            if 'class_net.predict' in n:
                m.bias.data.fill_(-math.log((1 - 0.01) / 0.01))
            else:
                m.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1.0)
        m.bias.data.zero_()


def get_feature_info(backbone):
    return backbone.feature_info.get_dicts(keys=['num_chs', 'reduction'])


_MODES = ('full_net', 'bb', 'fpn', 'only_fpn', 'fpn_and_head', 'head', 'not_cls', 'supp_bb')


class EfficientDet(nn.Module):
    """Same constructor / attributes / forward modes as the reference class (efficientdet.py:831-933).

    Extra (build-defined): after any forward that runs the class head,
    `self.ood_energy` and `self.ood_max_logit` hold the per-anchor scores [B, N] (fp32) computed in
    the class-head epilogue.
    """

    def __init__(self, config, pretrained_backbone=True, alternate_init=False):
        super().__init__()
        _check(config)
        self.config = config
        self.backbone = create_backbone(
            config.backbone_name, features_only=True, out_indices=(2, 3, 4),
            pretrained=pretrained_backbone, **(config.backbone_args or {}))
        feature_info = get_feature_info(self.backbone)
        self.fpn = BiFpn(self.config, feature_info)
        self.class_net = HeadNet(self.config, num_outputs=self.config.num_classes)
        self.box_net = HeadNet(self.config, num_outputs=4)
        self.num_anchors = len(config.aspect_ratios) * config.num_scales
        for n, m in self.named_modules():
            if 'backbone' not in n:
                (_init_weight_alt if alternate_init else _init_weight)(m, n)
        self._engine = None
        self._train_engine = None
        self.autograd = None        # None: differentiable forward iff self.training and grad mode and trainable params; True/False forces
        # 'native': the kernels compute in the parameters' dtype (float32 parity / bfloat16 throughput).  'accurate' (float32
        # parameters): two-term bf16 values everywhere - float32-grade results (north_star's 1e-3) at matrix-core speed.
        self.compute_mode = 'native'
        # [0]: bumped whenever parameters may have changed (load_state_dict, .to(), reset_head, invalidate(), PretrainStep);
        # [1]: cached (module ids, parameter + buffer tensors) behind weights_token().  Shared by shallow copies of the model.
        self._wver = [0, None]
        self._reg = [0]             # registrations on this model's modules (see _MODULE_OWNER); shared by shallow copies
        self._own_modules()
        self.ood_energy = None
        self.ood_max_logit = None
        # normalisation applied when a raw uint8 batch is passed to forward (effdet/data/loader.py:114-128)
        self.input_mean = (0.485, 0.456, 0.406)
        self.input_std = (0.229, 0.224, 0.225)
        self.supp_level_offset = 2  # FLAGS.supp_level_offset of the reference (infer.py:94, pretrain.py:63), read by mode='supp_cls'
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    # ---- engine management -----------------------------------------------------------------
    def _own_modules(self):
        for m in self.modules():
            _MODULE_OWNER[m] = self._reg

    def invalidate(self):
        """Drop packed weights; call after changing parameters in place (load_state_dict does it)."""
        self._engine = None
        self._wver[0] += 1

    def _apply(self, fn, *a, **k):
        self._engine = None
        self._train_engine = None
        self._wver[0] += 1
        return super()._apply(fn, *a, **k)

    def __getstate__(self):
        # the engine holds device buffers and ctypes arrays: never copied / pickled, rebuilt on demand
        d = self.__dict__.copy()
        d['_engine'] = None
        d['_train_engine'] = None
        return d

    def train(self, mode=True):
        # a train() <-> eval() transition usually brackets parameter updates: packed / BN-folded weights are rebuilt after it
        if bool(mode) != self.training:
            self._engine = None
            self._wver[0] += 1
        return super().train(mode)

    def weights_token(self):
        """Cheap fingerprint of the parameters the packed engine was built from: the explicit version, the identity of the
        sub-modules the scripts replace (infer.py:191 swaps class_net, reset_head swaps predict.conv_pw) and the sum of the
        autograd version counters of every parameter and buffer - any in-place update through the tensor itself
        (`optimizer.step()`, `p.add_()`, `p.copy_()`, BatchNorm running statistics) changes it.  Writes through `p.data`
        bypass version counters: call `invalidate()` after those."""
        pred = getattr(self.class_net, 'predict', None)
        mods = (id(self.backbone), id(self.fpn), id(self.class_net), id(self.box_net), id(getattr(pred, 'conv_pw', None)))
        c = self._wver[1]
        # the cached tensor list is rebuilt when a module of THIS model registered a parameter, buffer or sub-module since it was
        # made (`m.weight = nn.Parameter(...)`, `m.conv = nn.Conv2d(...)`: self._reg counts those through torch's global
        # registration hooks) - a replaced Parameter object would otherwise leave its predecessor in the list
        if c is None or c[0] != mods or c[2] != self._wver[0] or c[3] != self._reg[0] or _MODULE_OWNER.get(self) is not self._reg:   # (last: a deep copy)
            self._own_modules()                 # new sub-modules join the owner map
            c = self._wver[1] = (mods, list(self.parameters()) + list(self.buffers()), self._wver[0], self._reg[0])
        v = 0
        for t in c[1]:
            v += t._version
        return (self._wver[0], mods, v, c[3])

    def train_signature(self):
        """What the training engine's recorded tables depend on besides parameter VALUES: the identity of the sub-modules the
        scripts swap and the count of parameter / buffer / module registrations on this model's modules (a replaced Parameter
        object, a new head).  A change means: build a new TrainEngine (its first step records again)."""
        pred = getattr(self.class_net, 'predict', None)
        self.weights_token()                    # modules added since the last look join the owner map
        return (id(self.backbone), id(self.fpn), id(self.class_net), id(self.box_net), id(getattr(pred, 'conv_pw', None)), self._reg[0])

    def prepare(self, batch_size, image_size=None, ood_out=None):
        """Fold BN, repack weights to the kernel layouts and build the launch plan."""
        from ..engine import Engine
        image_size = tuple(image_size or self.config.image_size)
        self._engine = Engine(self, int(batch_size), image_size, ood_out=ood_out)
        return self._engine

    def engine_for(self, batch_size, image_size):
        e = self._engine
        if e is None or e.B != batch_size or e.image_size != tuple(image_size) or not e.matches(self) or e.wtoken != self.weights_token():
            e = self.prepare(batch_size, image_size)
        return e

    # ---- reference API -----------------------------------------------------------------------
    @torch.jit.ignore()
    def reset_head(self, num_classes=None, num_channels=None, aspect_ratios=None, num_scales=None, alternate_init=False):
        """effdet/efficientdet.py:854-886: a new class count only swaps class_net.predict.conv_pw."""
        reset_box_head = False
        set_config_writeable(self.config)
        if num_classes is not None:
            self.config.num_classes = num_classes
        if aspect_ratios is not None:
            reset_box_head = True
            self.config.aspect_ratios = aspect_ratios
        if num_scales is not None:
            reset_box_head = True
            self.config.num_scales = num_scales
        if num_classes is not None:
            like = self.class_net.predict.conv_pw.weight
            conv = nn.Conv2d(self.config.fpn_channels, num_classes * self.num_anchors, 1, bias=True)
            conv.bias.data.fill_(-math.log((1 - 0.01) / 0.01))
            self.class_net.predict.conv_pw = conv.to(device=like.device, dtype=like.dtype)
        if reset_box_head:
            like = self.box_net.predict.conv_pw.weight
            self.box_net = HeadNet(self.config, num_outputs=4)
            for n, m in self.box_net.named_modules(prefix='box_net'):
                (_init_weight_alt if alternate_init else _init_weight)(m, n)
            self.box_net.to(device=like.device, dtype=like.dtype)
        self.invalidate()

    @torch.jit.ignore()
    def toggle_head_bn_level_first(self):
        raise NotImplementedError('torchscript BN layout is not part of the HIP path')

    def forward(self, x, fast_weights=None, ret_activs=False, mode='full_net'):
        if mode in ('supp_cls', 'qry_cls'):
            from .meta_head import MetaHead
            if not isinstance(self.class_net, MetaHead):
                raise RuntimeError("mode %r needs `model.class_net = MetaHead(...)` as in infer.py:191" % mode)
            if mode == 'supp_cls':        # efficientdet.py:896-897: level_offset = FLAGS.supp_level_offset (default 2 in both
                # scripts, infer.py:94 / pretrain.py:63 - the support pass skips the two finest levels and its outputs line up
                # with the 3-level `proj_anchors` of dataloader.py:66); here: `model.supp_level_offset` or the config field
                off = getattr(self.config, 'supp_level_offset', None)
                return self.class_net(x, fast_weights=fast_weights, ret_activs=True,
                                      level_offset=self.supp_level_offset if off is None else off, heads='both')
            return self.class_net(x, fast_weights=fast_weights, ret_activs=ret_activs, heads='None')
        if mode not in _MODES:
            raise ValueError('unknown mode %r' % (mode,))
        if fast_weights is not None or ret_activs:
            raise NotImplementedError('fast_weights / ret_activs belong to the MetaHead path')
        if mode in _TRAIN_MODES and self.wants_autograd():
            return _run_train(self, x, mode)
        return _run(self, x, mode)

    def wants_autograd(self):
        """True when forward must be differentiable (pretrain.py:226-236): module in training mode, grad mode on and
        trainable parameters - or `self.autograd` forced.  The differentiable path is float32, keeps activations, uses
        batch statistics in every BatchNorm whose `.training` is set, and is several times slower than inference."""
        if self.autograd is not None:
            return bool(self.autograd)
        return self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())


_TRAIN_MODES = ('full_net', 'bb', 'fpn_and_head')


def _run_train(model, x, mode):
    """Differentiable forward on the training engine (train_engine.py); outputs carry autograd history."""
    from ..train_engine import TrainEngine, run_backbone, run_fpn_heads
    eng = model._train_engine
    if eng is None or eng.dev != model.backbone.conv_stem.weight.device or eng.signature != model.train_signature():
        eng = model._train_engine = TrainEngine(model)
    model.ood_energy = model.ood_max_logit = None          # the OOD epilogue belongs to the inference kernels
    if mode == 'bb':
        return run_backbone(eng, x)
    if mode == 'fpn_and_head':
        return run_fpn_heads(eng, list(x))
    return run_fpn_heads(eng, run_backbone(eng, x))


def _run(model, x, mode):
    if mode in ('full_net', 'bb', 'fpn', 'supp_bb'):
        if not (torch.is_tensor(x) and x.dim() == 4 and x.shape[1] == 3):
            raise ValueError('expected an image batch [B,3,H,W]')
        B, size = x.shape[0], (x.shape[2], x.shape[3])
    else:
        B = x[0].shape[0]
        first = model.fpn.in_feature_info[0] if mode in ('only_fpn', 'fpn_and_head', 'not_cls') else model.fpn.level_feature_info[0]
        red = first['reduction']
        size = (x[0].shape[2] * red, x[0].shape[3] * red)
    eng = model.engine_for(B, size)
    if mode == 'bb':
        return eng.run_backbone(x)

    def heads(activs_in, want_cls, want_box):
        """class / box outputs; with a MetaHead as class_net (infer.py:191) the class outputs are the MetaHead's forward on
        the pyramid, exactly as the reference's `self.class_net(x)` call (efficientdet.py:905-933)"""
        if want_cls and eng._cls_plan is None:
            _, box_o = eng.run_heads(activs_in, False, want_box) if want_box or activs_in is not None else (None, None)
            pyr = [v.contiguous() for v in eng.pyramid_views()] if activs_in is None else list(activs_in)
            model.ood_energy = model.ood_max_logit = None
            return model.class_net(pyr), box_o
        cls_o, box_o = eng.run_heads(activs_in, want_cls, want_box)
        if want_cls:
            model.ood_energy, model.ood_max_logit = eng.ood_energy, eng.ood_max_logit
        return cls_o, box_o

    if mode in ('full_net', 'fpn', 'supp_bb'):
        feats = eng.run_backbone(x, ret=mode == 'fpn')
        activs = eng.run_fpn(None, ret=mode != 'full_net' or eng._cls_plan is None)
        if mode == 'fpn':
            return feats, activs
        if mode == 'supp_bb':
            return activs
        return heads(None, True, True)
    if mode == 'only_fpn':
        return eng.run_fpn(x)
    if mode in ('fpn_and_head', 'not_cls'):
        activs = eng.run_fpn(x, ret=mode == 'not_cls' or eng._cls_plan is None)
        if mode == 'not_cls':
            return activs, heads(None, False, True)[1]
        return heads(None, True, True)
    # 'head'
    return heads(x, True, True)

from .meta_head import MetaHead  # noqa: E402,F401  (reference: effdet/efficientdet.py:569)
from .aux_nets import AnchorNet, ProjectionNet  # noqa: E402,F401  (reference: effdet/efficientdet.py:697,765)
