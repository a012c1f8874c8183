"""EfficientNet feature extractor: parameter container + architecture table.

Stands in for `timm.create_model(backbone_name, features_only=True, out_indices=(2, 3, 4))`
(reference call site: effdet/efficientdet.py:837-840).  timm is a third-party dependency that is not
vendored in the reference; its EfficientNet definition (B0 stage table, compound scaling with
`round_channels` to multiples of 8 and `ceil` depth scaling, SE width = 1/4 of the block's input
channels, TF-"SAME" padding and BN eps 1e-3 for the `tf_` variants) is restated here.

The modules below only HOLD parameters under timm's state-dict names
(`conv_stem.weight`, `bn1.*`, `blocks.{stage}.{block}.{conv_pw,bn1,conv_dw,bn2,se.conv_reduce,
se.conv_expand,conv_pwl,bn3}`); the arithmetic is done by the HIP engine (engine.py).
"""
import math

import torch
import torch.nn as nn

# (type, repeats, kernel, stride, expand, out_channels) for B0; se_ratio 0.25 everywhere
B0_STAGES = [
    ('ds', 1, 3, 1, 1, 16),
    ('ir', 2, 3, 2, 6, 24),
    ('ir', 2, 5, 2, 6, 40),
    ('ir', 3, 3, 2, 6, 80),
    ('ir', 3, 5, 1, 6, 112),
    ('ir', 4, 5, 2, 6, 192),
    ('ir', 1, 3, 1, 6, 320),
]
SCALING = {  # name -> (channel multiplier, depth multiplier)
    'tf_efficientnet_b0': (1.0, 1.0), 'tf_efficientnet_b1': (1.0, 1.1), 'tf_efficientnet_b2': (1.1, 1.2),
    'tf_efficientnet_b3': (1.2, 1.4), 'tf_efficientnet_b4': (1.4, 1.8), 'tf_efficientnet_b5': (1.6, 2.2),
}
# timm's non-`tf_` variants (efficientnet_b0 ... b5: the backbones of efficientdet_d0 ... d5, the models pretrain.py:81-112 and
# infer.py:119-149 build by default) share the architecture; they differ in the padding convention (pad_type '' = static symmetric
# padding instead of TF-"SAME") and in BatchNorm's eps / momentum (PyTorch defaults 1e-5 / 0.1 instead of TensorFlow's 1e-3 / 0.01)
SCALING.update({k[3:]: v for k, v in list(SCALING.items())})
FEATURE_STAGES = (2, 4, 6)      # feature_info indices (2, 3, 4) -> strides 8, 16, 32
BN_EPS_TF = 1e-3


def round_channels(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def efficientnet_arch(name):
    """-> (stem_chs, [[block dict]]) ; block: type,k,s,cin,mid,cout,se,residual."""
    if name not in SCALING:
        raise KeyError('unknown backbone %r (known: %s)' % (name, ', '.join(sorted(SCALING))))
    cm, dm = SCALING[name]
    stem = round_channels(32 * cm)
    cin = stem
    stages = []
    for (btype, rep, k, s, e, c) in B0_STAGES:
        cout = round_channels(c * cm)
        blocks = []
        for b in range(int(math.ceil(rep * dm))):
            stride = s if b == 0 else 1
            blocks.append(dict(type=btype, k=k, s=stride, cin=cin, mid=cin * e, cout=cout,
                               se=max(1, int(cin * 0.25 + 0.5)), residual=(stride == 1 and cin == cout)))
            cin = cout
        stages.append(blocks)
    return stem, stages


_BN_KW = dict(eps=BN_EPS_TF, momentum=0.01)          # of the network being constructed (EfficientNetFeatures.__init__ sets it)


def _bn(c):
    return nn.BatchNorm2d(c, **_BN_KW)


class _SE(nn.Module):
    def __init__(self, c, r):
        super().__init__()
        self.conv_reduce = nn.Conv2d(c, r, 1, bias=True)
        self.conv_expand = nn.Conv2d(r, c, 1, bias=True)


class _DsBlock(nn.Module):
    def __init__(self, b):
        super().__init__()
        self.conv_dw = nn.Conv2d(b['cin'], b['cin'], b['k'], groups=b['cin'], bias=False)
        self.bn1 = _bn(b['cin'])
        self.se = _SE(b['cin'], b['se'])
        self.conv_pw = nn.Conv2d(b['cin'], b['cout'], 1, bias=False)
        self.bn2 = _bn(b['cout'])


class _IrBlock(nn.Module):
    def __init__(self, b):
        super().__init__()
        self.conv_pw = nn.Conv2d(b['cin'], b['mid'], 1, bias=False)
        self.bn1 = _bn(b['mid'])
        self.conv_dw = nn.Conv2d(b['mid'], b['mid'], b['k'], groups=b['mid'], bias=False)
        self.bn2 = _bn(b['mid'])
        self.se = _SE(b['mid'], b['se'])
        self.conv_pwl = nn.Conv2d(b['mid'], b['cout'], 1, bias=False)
        self.bn3 = _bn(b['cout'])


class _FeatureInfo(object):
    def __init__(self, info):
        self.info = info

    def get_dicts(self, keys=None):
        return [{k: d[k] for k in (keys or d.keys())} for d in self.info]


class EfficientNetFeatures(nn.Module):
    """Parameter container; `forward` is provided by the owning EfficientDet's engine."""

    def __init__(self, name, drop_path_rate=0.0, **unused_backbone_args):
        super().__init__()
        self.name = name
        stem, stages = efficientnet_arch(name)
        self.arch = (stem, stages)
        self.pad_type = 'same' if name.startswith('tf_') else ''
        _BN_KW.clear()
        _BN_KW.update(dict(eps=BN_EPS_TF, momentum=0.01) if name.startswith('tf_') else dict(eps=1e-5, momentum=0.1))
        # Stochastic depth (timm's `drop_path_rate` in config.backbone_args, 0.2 for every tf_efficientdet config and the value
        # pretrain.py:49,94 passes): block i of n drops its residual branch per SAMPLE with probability rate * i / n while the
        # backbone module is in training mode; applied by the training engine (train_engine.py), never at inference.
        self.drop_path_rate = float(drop_path_rate)
        self.drop_path_masks = None     # tests: explicit keep masks {block index: float tensor [B] of 0 / 1} instead of random ones
        self.conv_stem = nn.Conv2d(3, stem, 3, stride=2, bias=False)
        self.bn1 = _bn(stem)
        self.blocks = nn.Sequential(*[
            nn.Sequential(*[(_DsBlock(b) if b['type'] == 'ds' else _IrBlock(b)) for b in blocks])
            for blocks in stages])
        info, red = [], 2
        for si, blocks in enumerate(stages):
            for b in blocks:
                red *= b['s']
            if si in FEATURE_STAGES:
                info.append(dict(num_chs=blocks[-1]['cout'], reduction=red))
        self.feature_info = _FeatureInfo(info)
        self._init_weights()

    def _init_weights(self):
        # timm's efficientnet init: conv ~ N(0, sqrt(2/fan_out)), BN (1, 0), zero biases
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = (m.kernel_size[0] * m.kernel_size[1] * m.out_channels) // m.groups
                m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1.0)
                m.bias.data.zero_()

    def block_drop_rates(self):
        """per block (flat order) drop probability: drop_path_rate * block_index / block_count (timm's EfficientNetBuilder);
        only blocks with a residual connection use it"""
        n = sum(len(b) for b in self.arch[1])
        rates, i = [], 0
        for blocks in self.arch[1]:
            for b in blocks:
                rates.append(self.drop_path_rate * i / n if b['residual'] else 0.0)
                i += 1
        return rates

    def forward(self, x):
        raise RuntimeError('EfficientNetFeatures has no standalone forward: call EfficientDet(x, mode="bb")')


def create_backbone(name, features_only=True, out_indices=(2, 3, 4), pretrained=False, **kwargs):
    if not features_only or tuple(out_indices) != (2, 3, 4):
        raise ValueError('only features_only=True, out_indices=(2, 3, 4) is on the hot path')
    if pretrained:
        # timm would download ImageNet weights here.  Nothing can be fetched in this environment, and the scripts that construct
        # EfficientDet(h) with the default pretrained_backbone=True load a complete checkpoint right afterwards (pretrain.py:137-141,
        # infer.py:174-185, strict load): say so loudly and go on with timm's random initialisation.
        import warnings
        warnings.warn('pretrained backbone weights for %r cannot be fetched offline: the backbone is randomly initialised - load a '
                      'checkpoint with load_state_dict (reference: effdet/helpers.py:14-22)' % (name,), RuntimeWarning, stacklevel=3)
    return EfficientNetFeatures(name, **kwargs)
